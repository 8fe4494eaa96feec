// cigar_dev.h — device side of herro_job_create's text work: CIGAR text -> binary ops + cut records, one workgroup per
// alignment (cigar_dev.hip).  The host keeps the windowing itself (window_cuts, windowing.hpp), which reads nothing but
// these records.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace herro {

struct CigIn {            // one alignment (32 B)
  uint64_t txt_off;       // 16-byte aligned offset in the text blob of the 16 bytes that hold its first byte (staged texts start there and are zero padded to 16)
  uint32_t len;           // text bytes
  uint32_t tstart;        // target position of the first op
  uint32_t op_off;        // first slot of its ops in the job's op array (room for len / 2 + 1)
  uint32_t cut_off;       // first slot of its cut records
  uint32_t cut_cap;       // room for that many
  uint32_t skip;          // bytes in front of its text inside those 16 (0..15; texts copied up from where the caller has them keep their host alignment)
};

enum : uint32_t {
  CIG_MALFORMED = 1,      // anything CigarIter would panic on, or a length above 30 bits: the host re-reads the text for the message
  CIG_INS_PAIR = 2,       // two insertion ops in a row (the host needs their positions: it scans this alignment itself)
  CIG_CUT_OVERFLOW = 4,   // more window boundaries than the alignment's coordinates allow for
};

struct CigOut {           // (32 B)
  uint32_t n_ops, t_end, q_end, ins_end;   // totals after the last op (t_end absolute)
  uint32_t n_cuts, flags;
  uint32_t op0, opn;      // first / last op
};

struct CigCut {           // == herro::Cut + padding (32 B); in discovery order, the host sorts by k
  uint32_t k, t, q, ins, o0, o1, o2, pad;
};

void launch_cigar_scan(const uint8_t* d_txt, const CigIn* d_in, CigOut* d_out, CigCut* d_cuts, uint32_t* d_ops,
                       uint32_t n_aln, uint32_t W, hipStream_t st);

}  // namespace herro
