// build_dev.h — herro_job_create's windowing and descriptor build ON THE DEVICE (round 6; build_dev.hip).
//
// Until round 5 the GPU scanned the CIGAR text (cigar_dev.hip) and then WAITED for the host: 8 MB of cut records came down, the host cut the windows
// from them (window_cuts, windowing.hpp), bucketed the overlaps by window, validated, merged the targets and sent the descriptors up — 0.7 + 0.4 ms of host
// work per 4096 windows on every feeder thread, the part of the end-to-end leg that does not scale with the GPU (features.rs:337-361 does the same work
// inside the feature thread, next to the data).  Here the same records are turned into the same descriptors by five small kernels behind the scan; the host
// reads back 64 bytes of totals (to size the job's arena) and the window descriptors (it needs them to hand windows out), nothing else.
//
// What the host still does, because it needs no CIGAR: the per-alignment rules of parse_paf (self overlaps, a second alignment of a (query, target) pair —
// overlaps.rs:175-185), the ratio classes by read name (features.rs:494), the coordinate checks.  Anything unusual the device meets — a text the scan
// kernel does not read, an input the reference panics on — only raises a flag: the job is then built by the host path as before, which also words the error.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cigar_dev.h"
#include "job_dev.h"
#include "pileup_core.h"

namespace herro {

struct AlnMeta {          // one alignment of the job, from the host's pre-pass (32 B)
  uint32_t qid;
  uint32_t qstart, qend;  // the alignment's query range (PAF columns 3, 4)
  uint32_t tstart, tend;
  uint32_t flags;         // bit 0: reverse strand; bit 1: left out (parse_paf would have dropped it): contributes no window
  uint32_t tgt;           // target index within the job
  uint32_t cls;           // job-level ratio class (same target + same query NAME)
};

struct TgtMeta {          // one target read of the job (32 B)
  uint32_t rid, tlen, n_windows, win0;   // win0: its first window in the job
  uint32_t aln0, n_aln;                  // its alignments in the job's arrays
  uint32_t pad0, pad1;
};

struct HowRec {           // one alignment restricted to one window, as extract_windows emits it (windowing.rs:7-16) + the slice's untrimmed totals (64 B)
  uint32_t win;           // window index within the target read
  uint32_t tstart, qstart, qend;
  uint32_t op_lo, op_hi, start_off, end_off;
  uint32_t st, sq, si;    // target / query / insertion bases of the untrimmed slice
  uint32_t op_first, op_last;
  uint32_t scr_local;     // ops of the alignment's earlier slices (its share of the event scratch starts there)
  uint32_t pad0, pad1;
};

struct AlnHead {          // per alignment (16 B)
  uint32_t n_how;         // windows it contributes
  uint32_t w_first;       // ... consecutive, from this window of the target on
  uint32_t op_sum;        // ops of all its slices
  uint32_t scr_base;      // job-level exclusive prefix of op_sum (k_scan_alns)
};

struct WinAcc {           // per window, first pass (32 B)
  uint32_t ow_cnt, lub;
  uint64_t ev;            // sum over its overlaps of op_cnt + 2: room of its insertion-event arrays
  uint64_t rd_bytes, op_bytes;   // algorithmic bytes (bench.py's roofline accounting)
};

struct BuildTotals {      // (64 B) what the host reads back
  uint32_t err;           // != 0: something the host path has to look at (it rebuilds the job and words the error)
  uint32_t n_ow, max_cols, n_tiles;
  uint64_t scr_ops, fin_bytes, row_elems, rd_bytes, op_bytes;
  uint64_t pad;
};

enum : uint32_t {         // BuildTotals::err bits (diagnostic only: the host path decides what the input is)
  BLD_SCAN_FLAG = 1,      // the scan kernel flagged the text (malformed, cut overflow)
  BLD_WINDOWING = 2,      // window_cuts would have failed
  BLD_VALIDATE = 4,       // an overlap failed a check of build_target
  BLD_SIZE = 8,           // a 32-bit offset would overflow
};

struct BuildDev {         // everything the kernels need; arrays in the job's scan arena unless noted
  const CigIn* in;
  const CigOut* out;
  CigCut* cuts;           // sorted in place by op index
  const AlnMeta* am;
  const TgtMeta* tm;
  const uint32_t* win_tgt;   // [n_win] target of every window
  AlnHead* head;
  HowRec* how;            // [cut slots]: an alignment's records at its cut_off
  WinAcc* wacc;
  BuildTotals* tot;
  uint32_t n_aln, n_tgt, n_win, W;
  const uint64_t* read_word_off;   // context read store
  const uint64_t* read_qual_off;
  // second phase (the job's arena exists)
  OwDesc* ow;
  WinDesc* win;
  uint32_t* tile_win;
  uint32_t* tile_r0;
  const uint32_t* ow_begin;  // [n_win + 1] (k_scan_wins)
  const uint64_t* ev_off;    // [n_win]
  const uint64_t* tile_off;  // [n_win]
  const uint64_t* row_off;   // [n_win]
};

// phase 1, behind launch_cigar_scan on the same stream: windows of every alignment, per-window counts, totals (the caller copies *tot down and decides)
void launch_build_phase1(const BuildDev& B, uint32_t* ow_begin, uint64_t* ev_off, uint64_t* tile_off, uint64_t* row_off, hipStream_t st);
// phase 2: overlap / window descriptors and the tile list into the job's arena
void launch_build_phase2(const BuildDev& B, hipStream_t st);

}  // namespace herro
