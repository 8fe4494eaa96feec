// job_dev.h — device-side view of one job (plain pointers, passed to kernels by value) and the
// HIP-event kernel timer used by bench.py's roofline leg.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "pileup_core.h"

#define HERRO_ROWS 31
#define HERRO_MAX_WINDOW 8192
#define HERRO_TILE 1024       // rows of the final matrix per k_tokens workgroup (a "tile" of the job's tile list)

namespace herro {

// A/B switches of measurements (the environment variables the documents' same-box comparisons were made with) exist only in libraries built
// with -DHERRO_PROF_BUILD (tools/prof.sh, `build_hip(out=..., defines=("HERRO_PROF_BUILD",))`); a release library takes the default and
// carries no untested product path behind an environment variable.  Options a user may set stay plain getenv calls:
// HERRO_HOST_THREADS, HERRO_FEATURIZE_PLANES, HERRO_ZERO_COPY, HERRO_FASTX_*, HERRO_TRACE, HERRO_HOST_PROFILE.
#ifdef HERRO_PROF_BUILD
inline int ab_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
inline int ab_env(const char*, int dflt) { return dflt; }
#endif

// A window's column table (one entry per row of the final matrix; entry 0 = the target): what the token and quality
// kernels need to know about a selected overlap, written by k_layout.
struct __attribute__((aligned(16))) CTab {
  int32_t off;         // window position where the overlap starts
  uint32_t t_total;    // target bases it covers (0: padding column)
  uint32_t tokc;       // token offset of a base (0 forward, 5 reverse) | gap token << 8
  uint32_t ow;         // overlap index (0xffffffff: none)
  uint32_t n_ev;       // insertion events
  uint32_t ev_off;     // first of them in JobDev::iev
  int32_t sbase, sdir; // stored index of alignment-orientation query base q = sbase + sdir * q (features.rs:97-108,128-153)
  uint64_t qual_off;   // first quality byte of the read
  uint64_t q_woff;     // first bit-plane word of the read
};

struct PlaneRec { uint32_t m, lo, hi; };   // (12 bytes, 4-byte aligned: one global_load_dwordx3)

struct JobDev {
  // ---- read store (context-owned; HBM-resident for the life of the context)
  const uint64_t* read_words;     // 2-bit packed bases, every read starts on a u64 boundary (+1 pad word)
  const uint32_t* read_p0;        // the same bases as bit planes: bit i of word k = low code bit of base 32k+i
  const uint32_t* read_p1;        // ... high code bit (A0 C1 G2 T3); indexed like read_words, +2 pad words
  const uint64_t* read_word_off;  // [n_reads+1]
  const uint8_t* read_qual;       // phred+33 bytes
  const uint64_t* read_qual_off;  // [n_reads+1]
  uint64_t read_qual_bytes;       // size of read_qual
  uint64_t read_n_words;          // u64 words in read_words (planes hold as many u32 words + 2)
  const double* ln_table;         // ln(k+1) computed on the host with glibc (bit-faithful to Rust std)
  uint32_t ln_table_n;
  // ---- descriptors (uploaded by herro_job_create)
  uint32_t n_ow, n_win, n_cls, n_tiles, window_size;
  uint32_t dbg_flags;  // bit 0 (HERRO_DEBUG_CDIR_OVERFLOW=1 at job creation): k_cols marks every directory record as not fitting, so that k_rfq's counting path runs (tests)
  uint32_t nw;        // plane words per column: ceil(window_size / 32)
  uint32_t max_cols;  // 1 + max overlaps per window (sizes the bit-sliced counters)
  const uint32_t* ops;
  const OwDesc* ow;
  const WinDesc* win;
  const uint32_t* tile_win;  // [n_tiles] window of each tile of HERRO_TILE rows (tiles cover the upper bound lub of every window)
  const uint32_t* tile_r0;   // [n_tiles] first row of the tile
  // ---- scratch / results
  PlaneRec* cw;          // [ow][nw] per word of 32 window positions of a kept overlap, ONE 12-byte record: {M plane word (a query base is aligned here), low code bit, high
                         // code bit (complemented for reverse-strand queries)} — what k_win and k_rows stream (every column of every window), and
  uint32_t* cwd;         // [ow][nw] the word's directory: alignment-orientation query index of the first base at or behind position 32 * word | insertion events of the
                         // overlap in front of that position << 20 (0xffffffff: does not fit).  Only k_rfq reads it; inside the record (16 bytes, rounds 5-6) it was a quarter
                         // of the bytes k_win / k_rows move
  uint4* iev;            // per overlap (at scr_off): insertion events {pos | trimmed len << 16, query index, first 16 bases, untrimmed len}
  uint32_t* ins_cnt;     // [ow] number of insertion events
  uint4* ocol;           // [ow] {window position where the overlap starts, target bases covered, kept, ratio class}
  uint8_t* ow_keep;      // long-indel filter verdict
  float* ow_acc;         // accuracy
  uint32_t* ow_ttotal;   // target bases consumed by the slice
  uint32_t* slot_ow;     // [win.ow_begin + slot] -> overlap index, slots ordered by accuracy rank
  uint32_t* sel_ow;      // [win * 32 + c], c in [1,31): overlap feeding final row c (0xffffffff: padding)
  CTab* ctab;            // [win * 32 + c]
  uint2* chdr2;          // [tile] {position of the tile's first row, 1 if that row is the position's base row}
  uint32_t* tile_nsup;   // [tile] informative rows found in the tile
  uint4* tev;            // per window (at ctab[0].ev_off): per tile, the inserted-base runs reaching into it {pos | len << 16, query index, first 16 bases, column | hidden rows << 8}
  uint2* tile_ev;        // [tile] {first slot of the tile's runs relative to the window's base, count}
  uint32_t* win_nkept;
  uint32_t* win_Lf;      // rows of the final matrix (L')
  uint32_t* win_nsup;
  uint32_t* row_of_pos2; // [win.pos_off + p], p in [0, win_len]: final row of each target position
  uint32_t* rowmap2;     // [win.row_off + row] = pos | ins << 16
  uint32_t* sup_row;     // [win.row_off + k]   final row of informative position k
  uint32_t* sup_pi;      // [win.row_off + k]   pos | ins << 16
  uint32_t* sup_nr;      // [win.row_off + k]   lean path: rows (1 + max insertion, <= 51) of positions pos - 2, pos - 1, pos, pos + 1, 6 bits each (0: outside the window) —
                         // with sup_row / sup_pi everything k_rfq needs to name the cells of the five rows around the informative row, without the row map
  uint8_t* fin_b;        // final token planes   [win.fin_off + c*lub + row], c in [0,31)
  uint8_t* fin_q;        // qualities: only receptive-field cells after infer, complete after launch_full_quals
  uint32_t* nd;          // [2*cls]: matches, mismatches (features.rs:461-500)
  uint32_t* rank_qid;    // [win.ow_begin + rank] ranked query ids (features.rs:569)
  uint8_t* cons_seq;     // [win.row_off ..] corrected bases of the window (ASCII), cons_len[w] of them
  uint8_t* cons_tmp;     // planes path (k_tokens): [win.row_off + row] per-row call before '*' removal.  Lean path (k_rows): [win.row_off + i], i = index of an
                         // INSERTION row among the window's insertion rows (row - position - 1): its majority vote | informative << 7
  // lean path, receptive fields gathered by k_rows itself (null rf: not this pass): the job's record buffer and its room in informative rows, the
  // receptive field's half width (2), the allocation counter (one atomicAdd per window) and the first record slot every window got (0xffffffff: none)
  uint8_t* rf;
  uint64_t rf_cap;
  uint32_t rf_half;
  uint32_t* rf_alloc;
  uint32_t* win_rfbase;  // [win]
  uint32_t* vpl;         // lean path: [win][4][nw] majority vote of every target position's base row as three bit planes (code 0..4 = A C G T *) + the positions with insertion rows behind them
  uint32_t* cons_len;    // [win]
  unsigned long long* prof;  // HERRO_PROF_BUILD libraries run with HERRO_PROF=1: [kernel * 16 + phase][32 shards] shader cycles, [.. + 15] workgroups (null otherwise)
};

// Phase timer for kernel development.  Compiled in only when the library is built with -DHERRO_PROF_BUILD
// (HERRO_PROF_BUILD=1 python -m herro_amd.build); such a build, run with HERRO_PROF=1, prints shader cycles per phase of the
// pileup kernels when a context is destroyed.  Release kernels carry none of it (the marks expand to nothing).
// Thread 0 of one workgroup in 32 adds the cycles since the previous mark, sharded 32 ways: an atomic per mark from every
// workgroup on one address slowed the kernels 5x and drowned the signal.
#ifdef HERRO_PROF_BUILD
#define PROF_ON(J) ((J).prof && threadIdx.x == 0 && (blockIdx.x & 31u) == 0)
#define PROF_BEGIN(J) unsigned long long _pt = PROF_ON(J) ? __builtin_readcyclecounter() : 0ull
#define PROF_MARK(J, kern, phase)                                                              \
  do {                                                                                        \
    if (PROF_ON(J)) {                                                                         \
      const unsigned long long _t = __builtin_readcyclecounter();                             \
      const unsigned _sh = (blockIdx.x >> 5) & 31u;                                           \
      atomicAdd(&(J).prof[((kern) * 16 + (phase)) * 32 + _sh], _t - _pt);                     \
      if ((phase) == 0) atomicAdd(&(J).prof[((kern) * 16 + 15) * 32 + _sh], 1ull);            \
      _pt = __builtin_readcyclecounter();                                                     \
    }                                                                                         \
  } while (0)
#else
#define PROF_BEGIN(J) do { } while (0)
#define PROF_MARK(J, kern, phase) do { } while (0)
#endif

// Accumulates GPU time per kernel group with HIP events recorded on the launch stream.
struct KernelTimer {
  bool on = false;
  struct Rec { std::string name; hipEvent_t a, b; };
  std::vector<Rec> pending;
  std::map<std::string, std::pair<double, uint64_t>> acc;  // name -> (ms, calls)
  std::vector<std::string> order;
  // An event that cannot be created or recorded drops that one measurement; timing never fails the pipeline.
  void begin(const char* name, hipStream_t st) {
    Rec r;
    r.name = name;
    r.a = r.b = nullptr;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess || hipEventRecord(r.a, st) != hipSuccess) {
      if (r.a) (void)hipEventDestroy(r.a);
      if (r.b) (void)hipEventDestroy(r.b);
      r.a = r.b = nullptr;
    }
    pending.push_back(r);
  }
  void end(hipStream_t st) {
    Rec& r = pending.back();
    if (r.b && hipEventRecord(r.b, st) != hipSuccess) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); r.a = r.b = nullptr; }
  }
  void collect() {  // caller has synchronised the stream
    for (auto& r : pending) {
      if (!r.a) continue;
      float ms = 0;
      if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
        if (!acc.count(r.name)) order.push_back(r.name);
        auto& e = acc[r.name];
        e.first += ms;
        e.second += 1;
      }
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    pending.clear();
  }
  void reset() {
    collect();
    acc.clear();
    order.clear();
  }
};

#define KT_BEGIN(tm, name, st) do { if ((tm) && (tm)->on) (tm)->begin(name, st); } while (0)
#define KT_END(tm, st) do { if ((tm) && (tm)->on) (tm)->end(st); } while (0)

// lean (the default of herro_job_featurize): no token planes — informative rows, votes and (launch_rf_quals) the receptive fields of the
// informative rows are derived from the column bit planes in position space; launch_full_tokens builds the planes for whoever asks.
// !lean: the planes path of rounds 3-4 (k_tokens derives everything while it writes the planes).
void launch_featurize(const JobDev& J, hipStream_t st, KernelTimer* tm, bool lean);
void launch_full_tokens(const JobDev& J, hipStream_t st, KernelTimer* tm);   // token planes + row map behind a lean featurize (informative rows / votes are left alone)
// the model's receptive fields (rows within `half` of an informative row)
// rf != null: compact, one 16-byte record per (informative row, column) at [(sup_off[w] + k) * 31 + column]: bytes 0..7 the TOKENS of rows
// sup_row[k] - half .. + 7 - half of that column, bytes 8..15 their qualities (needs 2 * half + 1 <= 8); else the qualities go into the
// quality planes (which needs the row map, i.e. the planes path or launch_full_tokens)
// cap: informative rows rf has room for (a window whose slots would lie beyond it is left out: launches in front of the host's count)
// lean: the job was featurized on the lean path (sup_nr is valid: the rows around an informative row are named without row_of_pos2)
// left_only: behind a fused gather (k_rows) — only the windows it reserved slots for and did not fill (more informative rows than it stages), at those slots
void launch_rf_quals(const JobDev& J, uint32_t half, const uint64_t* sup_off, uint8_t* rf, uint64_t cap, bool lean, hipStream_t st, KernelTimer* tm, bool left_only = false);
void launch_supoff(const JobDev& J, uint64_t* sup_off, hipStream_t st);   // sup_off[0 .. n_win]: prefix of win_nsup
// the complete quality planes (featurize itself only writes tokens)
void launch_full_quals(const JobDev& J, hipStream_t st);
void launch_consensus(const JobDev& J, const uint64_t* sup_off, const float* base_logits, bool lean, hipStream_t st, KernelTimer* tm);

}  // namespace herro
