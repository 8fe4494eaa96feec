// job_dev.h — device-side view of one job (plain pointers, passed to kernels by value) and the
// HIP-event kernel timer used by bench.py's roofline leg.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "pileup_core.h"

#define HERRO_ROWS 31
#define HERRO_MAX_WINDOW 8192
#define HERRO_TILE 256        // pileup rows per workgroup in the tile kernels

namespace herro {

// Per overlap-window column header, read through scalar loads by the tile kernels.
struct __attribute__((aligned(16))) ColHdr {
  int32_t off;        // window-relative position where the overlap starts
  uint32_t t_total;   // target bases the slice consumes
  uint32_t strand, cls;
  int32_t sbase, sdir;  // stored index of alignment-orientation query base q = sbase + sdir*q
  uint32_t md_off;    // first entry of the overlap's M/D op table
  uint32_t n_md;      // entries in it
  uint64_t q_woff;    // first 2-bit word of the query read
  uint64_t qual_off;  // first quality byte of the query read
};

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
  uint32_t n_ow, n_win, n_cls, n_tiles, window_size, n_bw;
  uint32_t max_cols;  // 1 + max overlaps per window (sizes the bit-sliced counters)
  uint32_t dbg;       // HERRO_DBG phase mask (profiling experiments only; 0 in production)
  const uint32_t* ops;
  const OwDesc* ow;
  const WinDesc* win;
  const uint32_t* tile_win;  // [n_tiles] window of each row tile
  const uint32_t* tile_r0;   // [n_tiles] first row of the tile
  // ---- scratch / results
  uint32_t* op_t;        // per (overlap, op): target bases consumed before the op
  uint32_t* op_q;        // ... query bases consumed before the op
  uint32_t* ins_ev;      // per overlap (at scr_off): insertion events, window-relative pos | len << 16
  uint32_t* ins_cnt;     // [ow] number of insertion events
  uint4* md;             // per overlap (at scr_off): M/D ops {t_beg, q_beg, len | isM<<31, following ins len}
  uint2* bm;             // [ow * n_bw + i] {bitmap of M/D op starts for positions 32i.., ops before 32i}
  ColHdr* chdr;          // [ow]
  struct TPlan* tplan;   // [tile * 32 + c] staging plan of final-row tile x selected column (64 B)
  struct TileHdr* thdr;  // [tile] window / row range / target words of the tile (64 B)
  uint8_t* ow_keep;      // long-indel filter verdict
  float* ow_acc;         // accuracy
  uint32_t* ow_ttotal;   // target bases consumed by the slice
  uint32_t* slot_ow;     // [win.ow_begin + slot] -> overlap index, slots ordered by accuracy rank
  uint32_t* sel_ow;      // [win * 32 + c], c in [1,31): overlap feeding final row c (0xffffffff: padding)
  uint32_t* win_nkept;
  uint32_t* win_Lf;      // rows of the final matrix (L')
  uint32_t* win_nsup;
  uint32_t* row_of_pos2; // [win.pos_off + p], p in [0, win_len]: final row of each target position
  uint32_t* rowmap2;     // [win.row_off + row] = pos | ins << 16
  uint8_t* sup_flag;     // [win.row_off + final row] informative?
  uint32_t* sup_row;     // [win.row_off + k]   final row of informative position k
  uint32_t* sup_pi;      // [win.row_off + k]   pos | ins << 16
  uint8_t* fin_b;        // final token planes   [win.fin_off + c*lub + row], c in [0,31)
  uint8_t* fin_q;        // qualities: only receptive-field cells after infer, complete after launch_full_quals
  uint32_t* nd;          // [2*cls]: matches, mismatches (features.rs:461-500)
  uint32_t* rank_qid;    // [win.ow_begin + rank] ranked query ids (features.rs:569)
  uint8_t* cons_seq;     // [win.row_off ..] corrected bases of the window (ASCII), cons_len[w] of them
  uint8_t* cons_tmp;     // [win.row_off + row] per-row call before '*' removal
  uint32_t* cons_len;    // [win]
  // ---- bit-plane featurizer (pileup.hip)
  uint32_t nw;           // plane words per column: ceil(window_size / 32)
  uint32_t* cpl;         // [ow][3][nw] column planes over the window's positions: query base present / low / high code bit (kept overlaps)
  uint4* iev;            // per overlap (at scr_off): insertion events {pos | trimmed len << 16, query index, first 16 bases, untrimmed len}
};

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

void launch_featurize(const JobDev& J, hipStream_t st, KernelTimer* tm);
void launch_featurize_old(const JobDev& J, hipStream_t st, KernelTimer* tm);
void launch_rf_quals_old(const JobDev& J, uint32_t half, hipStream_t st, KernelTimer* tm);
void launch_full_quals_old(const JobDev& J, hipStream_t st);
// qualities inside the model's receptive fields (rows within `half` of an informative row)
void launch_rf_quals(const JobDev& J, uint32_t half, hipStream_t st, KernelTimer* tm);
// the complete quality planes (featurize itself only writes tokens)
void launch_full_quals(const JobDev& J, hipStream_t st);
void launch_consensus(const JobDev& J, const uint64_t* sup_off, const float* base_logits, hipStream_t st, KernelTimer* tm);

}  // namespace herro
