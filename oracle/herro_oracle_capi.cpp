// ORACLE — TEST INFRASTRUCTURE ONLY (see herro_oracle.hpp header).
// Flat C API over the restatement so tests/ (ctypes) can drive it.  Never shipped, never
// linked into herro_amd/.
#include <cstdio>
#include <memory>

#include "herro_oracle.hpp"

using namespace oracle;

namespace {
thread_local std::string g_err;
struct Store {
  std::vector<HAECRecord> reads;
};
struct Result {
  std::vector<WindowFeatures> wins;
};
}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// ---- codec (haec_io.rs) -----------------------------------------------------------------
// encode ASCII -> words; returns number of words written (<= cap) or -1 on panic.
long orc_encode(const uint8_t* seq, uint64_t n, uint64_t* words, uint64_t cap) {
  try {
    HAECSeq s = encode(seq, n);
    if (s.data.size() > cap) { g_err = "cap too small"; return -1; }
    std::copy(s.data.begin(), s.data.end(), words);
    return (long)s.data.size();
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int orc_decode(const uint64_t* words, uint64_t n_words, uint64_t length, uint64_t start,
               uint64_t end, int reversed, uint8_t* out) {
  try {
    HAECSeq s;
    s.data.assign(words, words + n_words);
    s.length = length;
    decode(s, start, end, reversed != 0, out);
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// ---- read store ---------------------------------------------------------------------------
void* orc_store_new(uint32_t n_reads, const uint8_t* seq_cat, const uint8_t* qual_cat,
                    const uint64_t* seq_off, const uint8_t* id_cat, const uint64_t* id_off) {
  try {
    auto st = std::make_unique<Store>();
    st->reads.resize(n_reads);
    for (uint32_t i = 0; i < n_reads; i++) {
      auto& r = st->reads[i];
      const uint64_t a = seq_off[i], b = seq_off[i + 1];
      r.seq = encode(seq_cat + a, b - a);
      r.qual.assign(qual_cat + a, qual_cat + b);
      // haec_io.rs:52-54: id = up to first ' ' or '\t'; rest = description
      std::string full((const char*)id_cat + id_off[i], id_off[i + 1] - id_off[i]);
      size_t sp = full.find_first_of(" \t");
      if (sp == std::string::npos) r.id = full;
      else { r.id = full.substr(0, sp); r.description = full.substr(sp + 1); }
    }
    return st.release();
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void orc_store_free(void* s) { delete (Store*)s; }

struct orc_aln {
  uint32_t qid, qlen, qstart, qend, strand, tid, tlen, tstart, tend;
  uint32_t cigar_len;
  uint64_t cigar_off;
};

static std::vector<Alignment> make_alns(uint32_t n, const orc_aln* a, const uint8_t* blob) {
  std::vector<Alignment> v(n);
  for (uint32_t i = 0; i < n; i++) {
    v[i].overlap = Overlap{a[i].qid, a[i].qlen, a[i].qstart, a[i].qend,
                           a[i].strand ? Strand::Reverse : Strand::Forward,
                           a[i].tid, a[i].tlen, a[i].tstart, a[i].tend};
    v[i].cigar.assign((const char*)blob + a[i].cigar_off, a[i].cigar_len);
  }
  return v;
}

// ---- windowing only (windowing.rs) --------------------------------------------------------
// Writes up to cap rows of 8 u64: window, tstart, qstart, qend, cs_idx, cs_off, ce_idx, ce_off.
long orc_extract_windows(const orc_aln* a, const uint8_t* blob, uint32_t n_windows,
                         uint32_t window_size, int is_target, uint64_t* out, uint64_t cap) {
  try {
    auto alns = make_alns(1, a, blob);
    Windows w(n_windows);
    extract_windows(w, &alns[0].overlap, alns[0].cigar, 0, 0, is_target != 0, window_size);
    uint64_t k = 0;
    for (uint32_t i = 0; i < n_windows; i++)
      for (auto& ow : w[i]) {
        if (k >= cap) { g_err = "cap too small"; return -1; }
        uint64_t* o = out + 8 * k++;
        o[0] = i; o[1] = ow.tstart; o[2] = ow.qstart; o[3] = ow.qend;
        o[4] = ow.cigar_start_idx; o[5] = ow.cigar_start_offset;
        o[6] = ow.cigar_end_idx; o[7] = ow.cigar_end_offset;
      }
    return (long)k;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// ---- extract_features (features.rs:326) ---------------------------------------------------
void* orc_extract_features(void* store, uint32_t rid, uint32_t n_aln, const orc_aln* a,
                           const uint8_t* blob, uint32_t window_size) {
  try {
    auto* st = (Store*)store;
    auto alns = make_alns(n_aln, a, blob);
    auto res = std::make_unique<Result>();
    res->wins = extract_features(rid, st->reads, alns, window_size);
    return res.release();
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void orc_result_free(void* r) { delete (Result*)r; }
uint32_t orc_result_n_windows(void* r) { return (uint32_t)((Result*)r)->wins.size(); }

// dims: [L', n_supported, n_qids, n_alns, p1_n, p1_L, p1_n_supported, win_len(max_ins size)]
void orc_window_dims(void* r, uint32_t w, uint64_t* dims) {
  const auto& f = ((Result*)r)->wins[w];
  dims[0] = f.bases.rows; dims[1] = f.supported.size(); dims[2] = f.qids.size();
  dims[3] = f.n_alns(); dims[4] = f.p1_qids.size(); dims[5] = f.p1_L;
  dims[6] = f.p1_supported.size(); dims[7] = f.max_ins.size();
}
void orc_window_copy(void* r, uint32_t w, uint8_t* bases, uint8_t* quals, uint16_t* sup_pos,
                     uint8_t* sup_ins, uint32_t* qids) {
  const auto& f = ((Result*)r)->wins[w];
  std::memcpy(bases, f.bases.v.data(), f.bases.v.size());
  std::memcpy(quals, f.quals.v.data(), f.quals.v.size());
  for (size_t k = 0; k < f.supported.size(); k++) { sup_pos[k] = f.supported[k].pos; sup_ins[k] = f.supported[k].ins; }
  std::copy(f.qids.begin(), f.qids.end(), qids);
}
void orc_window_copy_p1(void* r, uint32_t w, uint32_t* p1_qids, float* p1_acc, double* scores,
                        uint16_t* max_ins, uint16_t* p1_sup_pos, uint8_t* p1_sup_ins) {
  const auto& f = ((Result*)r)->wins[w];
  std::copy(f.p1_qids.begin(), f.p1_qids.end(), p1_qids);
  std::copy(f.p1_acc.begin(), f.p1_acc.end(), p1_acc);
  std::copy(f.scores.begin(), f.scores.end(), scores);
  std::copy(f.max_ins.begin(), f.max_ins.end(), max_ins);
  for (size_t k = 0; k < f.p1_supported.size(); k++) { p1_sup_pos[k] = f.p1_supported[k].pos; p1_sup_ins[k] = f.p1_supported[k].ins; }
}

// ---- prepare_examples / collate (inference.rs:73-145,214-253) -----------------------------
// Runs prepare_examples over the result's windows with the given batch size and returns, for
// batch `bi`, the padded tensors.  Two-call protocol: dims first, then copy.
// dims: [n_batches, B, L, R, N(sum lens)]
int orc_collate_dims(void* r, uint32_t batch_size, uint32_t bi, uint64_t* dims) {
  try {
    auto d = prepare_examples(((Result*)r)->wins, batch_size);
    dims[0] = d.batches.size();
    if (bi < d.batches.size()) {
      const auto& b = d.batches[bi];
      dims[1] = b.B; dims[2] = b.L; dims[3] = b.R;
      uint64_t n = 0; for (auto l : b.lens) n += (uint64_t)l;
      dims[4] = n;
    }
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int orc_collate_copy(void* r, uint32_t batch_size, uint32_t bi, uint32_t* wids, uint8_t* bases,
                     uint8_t* quals, int32_t* lens, int32_t* indices_flat) {
  try {
    auto d = prepare_examples(((Result*)r)->wins, batch_size);
    const auto& b = d.batches.at(bi);
    std::copy(b.wids.begin(), b.wids.end(), wids);
    std::memcpy(bases, b.bases.data(), b.bases.size());
    std::memcpy(quals, b.quals.data(), b.quals.size());
    std::copy(b.lens.begin(), b.lens.end(), lens);
    size_t k = 0;
    for (auto& v : b.indices) for (auto x : v) indices_flat[k++] = x;
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
float orc_normalise_qual(uint8_t q) { return normalise_qual(q); }

// ---- consensus (consensus.rs:86-227) + FASTA (lib.rs:282-317) -----------------------------
// logits_flat: for each window in order, for each supported position, 5 floats (bases_logits);
// info logits are unused by the reference's consensus (consensus.rs:135 binds `_`).
// Returns FASTA text length (0 => consensus returned None) or -1 on panic.
long orc_consensus_fasta(void* store, void* r, const float* logits_flat, char* out, uint64_t cap) {
  try {
    auto* st = (Store*)store;
    auto d = prepare_examples(((Result*)r)->wins, 1u << 30);
    size_t k = 0;
    for (auto& cw : d.consensus_data) {
      if (cw.supported.empty()) continue;
      cw.has_logits = true;
      for (size_t s = 0; s < cw.supported.size(); s++, k++) {
        cw.info_logits.push_back(0.f);
        std::array<float, 5> b;
        for (int c = 0; c < 5; c++) b[c] = logits_flat[k * 5 + c];
        cw.bases_logits.push_back(b);
      }
    }
    std::stable_sort(d.consensus_data.begin(), d.consensus_data.end(),
                     [](const ConsensusWindow& a, const ConsensusWindow& b) { return a.wid < b.wid; });
    std::vector<std::vector<uint8_t>> seqs;
    if (d.consensus_data.empty() || !consensus(d.consensus_data, seqs)) return 0;
    std::string fa = write_read(st->reads[d.consensus_data[0].rid], seqs);
    if (fa.size() > cap) { g_err = "cap too small"; return -1; }
    std::memcpy(out, fa.data(), fa.size());
    return (long)fa.size();
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

}  // extern "C"
