// Synthetic overlap-batch generator (SURVEY.md §8 d).  Host-only C++ (g++), no HIP, no oracle.
//
// Produces what `herro inference --read-alns` would hold in RAM after parse_reads + parse_paf
// (reference lib.rs:133, overlaps.rs:117-202): a read set (ASCII bases + phred+33 quals) and,
// per target read, a list of PAF-style alignments with true-edit-script CIGARs (M/I/D only).
//
// Model: per read group a random diploid template (hap B = hap A + SNPs 2e-3 + 1-bp indels
// 2e-4); a target read of exactly `target_len` bases sampled from one haplotype with ONT-like
// noise (sub 0.6 %, ins 0.4 %, del 0.6 %, geometric indel lengths p=0.8 capped at 10);
// `n_overlaps` query reads, alternating haplotypes, spanning the target plus flanks, random
// strand, same noise.  CIGARs come from composing each read's edit script against the shared
// hap-A coordinate system, so they are exact.  PRNG: splitmix64-seeded xoshiro256**; normal
// variates by Irwin-Hall(12) so the stream is bit-reproducible on any libm.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Rng {
  uint64_t s[4];
  static uint64_t splitmix(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  explicit Rng(uint64_t seed) {
    for (auto& v : s) v = splitmix(seed);
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
  bool bern(double p) { return uniform() < p; }
  // uniform() < p without the conversion: uniform() is k * 2^-53 exactly (k = next() >> 11 < 2^53) and p * 2^53 is exact in a
  // double, so k * 2^-53 < p  <=>  k < p * 2^53  <=>  k < ceil(p * 2^53) for the integer k.  Same draws, same outcomes.
  static uint64_t threshold(double p) { return p > 0 ? (p >= 1 ? (1ull << 53) : (uint64_t)std::ceil(p * 9007199254740992.0)) : 0; }
  bool bern_t(uint64_t thr) { return (next() >> 11) < thr; }
  double normal() {
    double a = 0;
    for (int i = 0; i < 12; i++) a += uniform();
    return a - 6.0;
  }
  uint32_t geom(double p, uint32_t cap) {  // >=1
    uint32_t k = 1;
    while (k < cap && !bern(p)) k++;
    return k;
  }
};

const char ACGT[4] = {'A', 'C', 'G', 'T'};
const uint8_t DEL = 4;

// A sequence expressed in hap-A column coordinates: per column an optional base + insertions.  The insertions of all
// columns sit in one buffer (columns are always produced left to right): column i owns idata[ioff[i] .. ioff[i + 1]).
struct Cols {
  int64_t a0 = 0;                 // first column
  std::vector<uint8_t> base;      // 0..3 or DEL
  std::vector<uint32_t> ioff;     // base.size() + 1
  std::vector<uint8_t> idata;     // inserted bases 0..3
  uint32_t ins_n(size_t i) const { return ioff[i + 1] - ioff[i]; }
  const uint8_t* ins(size_t i) const { return idata.data() + ioff[i]; }
};

struct Params {
  uint64_t seed;
  uint32_t n_targets, target_len, n_overlaps, flank_min, flank_max;
  double p_sub, p_ins, p_del, p_long_indel, p_partial, p_n_base;
  uint32_t min_partial_len;
  double p_snp;  // SNP rate between the two haplotypes (SURVEY §8 d: 2e-3); raised by tests that need many informative rows
};

struct Out {
  std::vector<uint8_t> seq, qual;
  std::vector<uint64_t> off;  // n_reads+1
  // alignments
  std::vector<uint32_t> aln;  // 10 u32 per alignment: qid,qlen,qstart,qend,strand,tid,tlen,tstart,tend,cigar_len
  std::vector<uint64_t> cig_off;
  std::vector<uint8_t> cig;
  std::vector<uint64_t> tgt_aln_off;  // n_targets+1
  std::vector<uint32_t> tgt_rid;
};

// Apply sequencing noise to a haplotype given in columns; output also in columns.
Cols noisy_read(Rng& g, const Cols& hap, int64_t a_lo, int64_t a_hi, const Params& P) {
  Cols r;
  r.a0 = a_lo;
  const size_t n = (size_t)(a_hi - a_lo);
  r.base.assign(n, DEL);
  r.ioff.resize(n + 1);
  r.idata.reserve(n / 32 + 64);
  const uint64_t t_del = Rng::threshold(P.p_del), t_sub = Rng::threshold(P.p_sub), t_ins = Rng::threshold(P.p_ins),
                 t_long = Rng::threshold(P.p_long_indel);
  uint32_t del_left = 0;
  auto emit = [&](uint8_t hb, uint8_t* dst_base) {
    // one haplotype base -> maybe deleted / substituted, maybe followed by insertion
    if (del_left > 0) {
      del_left--;
    } else if (g.bern_t(t_del)) {
      del_left = (g.bern_t(t_long) ? 55 + g.below(20) : g.geom(0.8, 10)) - 1;
    } else {
      uint8_t b = hb;
      if (g.bern_t(t_sub)) b = (uint8_t)((b + 1 + g.below(3)) & 3);
      if (dst_base) *dst_base = b; else r.idata.push_back(b);
    }
    if (g.bern_t(t_ins)) {
      const uint32_t l = g.bern_t(t_long) ? 55 + g.below(20) : g.geom(0.8, 10);
      for (uint32_t k = 0; k < l; k++) r.idata.push_back((uint8_t)g.below(4));
    }
  };
  const size_t h0 = (size_t)(a_lo - hap.a0);
  for (size_t i = 0; i < n; i++) {
    const size_t h = h0 + i;
    r.ioff[i] = (uint32_t)r.idata.size();
    if (hap.base[h] != DEL) emit(hap.base[h], &r.base[i]);
    for (uint32_t k = hap.ioff[h]; k < hap.ioff[h + 1]; k++) emit(hap.idata[k], nullptr);
  }
  r.ioff[n] = (uint32_t)r.idata.size();
  return r;
}

void flatten(const Cols& c, std::vector<uint8_t>& out) {
  out.reserve(out.size() + c.base.size() + c.idata.size());
  for (size_t i = 0; i < c.base.size(); i++) {
    if (c.base[i] != DEL) out.push_back(c.base[i]);
    out.insert(out.end(), c.ins(i), c.ins(i) + c.ins_n(i));
  }
}

struct Cig {
  std::vector<std::pair<char, uint32_t>> ops;
  void add(char op, uint32_t n) {
    if (!n) return;
    if (!ops.empty() && ops.back().first == op) ops.back().second += n;
    else ops.push_back({op, n});
  }
};

void gen_quals(Rng& g, size_t n, std::vector<uint8_t>& out) {
  const size_t at = out.size();
  out.resize(at + n);
  uint8_t* d = out.data() + at;
  for (size_t i = 0; i < n; i++) {
    double q = 22.0 + 8.0 * g.normal();
    long v = (long)(q + (q >= 0 ? 0.5 : -0.5));
    v = std::max(2l, std::min(50l, v));
    d[i] = (uint8_t)(33 + v);
  }
}

void generate(const Params& P, Out& o) {
  Rng g(P.seed);
  o.off.push_back(0);
  o.tgt_aln_off.push_back(0);
  const int64_t margin = (int64_t)P.flank_max + 64;
  for (uint32_t t = 0; t < P.n_targets; t++) {
    // template long enough for the target (which may lose ~1% to deletions) plus flanks
    const int64_t TL = (int64_t)((double)P.target_len * 1.05) + 2 * margin + 256;
    Cols hap[2];
    for (int h = 0; h < 2; h++) {
      hap[h].a0 = 0;
      hap[h].base.resize((size_t)TL);
      hap[h].ioff.assign((size_t)TL + 1, 0);   // hap A carries no insertions
    }
    for (int64_t a = 0; a < TL; a++) {
      const uint8_t b = (uint8_t)g.below(4);
      hap[0].base[(size_t)a] = b;
      uint8_t bb = b;
      if (g.bern(P.p_snp)) bb = (uint8_t)((b + 1 + g.below(3)) & 3);
      if (g.bern(1e-4)) bb = DEL;
      hap[1].base[(size_t)a] = bb;
      hap[1].ioff[(size_t)a] = (uint32_t)hap[1].idata.size();
      if (g.bern(1e-4)) hap[1].idata.push_back((uint8_t)g.below(4));
    }
    hap[1].ioff[(size_t)TL] = (uint32_t)hap[1].idata.size();

    // --- target read: exactly target_len bases starting at column `margin`
    const int th = (int)g.below(2);
    Cols tc = noisy_read(g, hap[th], margin, TL - margin, P);
    {  // truncate to target_len bases
      size_t cnt = 0, i = 0;
      for (; i < tc.base.size() && cnt < P.target_len; i++) {
        if (tc.base[i] != DEL) cnt++;
        uint32_t ni = tc.ins_n(i);
        if (cnt + ni > P.target_len) {   // the last column keeps only the insertions that still fit (its successors are dropped below)
          ni = (uint32_t)(P.target_len - cnt);
          tc.ioff[i + 1] = tc.ioff[i] + ni;
        }
        cnt += ni;
      }
      // drop trailing columns that contributed nothing
      while (i > 0 && tc.base[i - 1] == DEL && tc.ins_n(i - 1) == 0) i--;
      tc.base.resize(i);
      tc.ioff.resize(i + 1);
      tc.idata.resize(tc.ioff[i]);
    }
    std::vector<uint8_t> tseq;
    flatten(tc, tseq);
    const uint32_t tlen = (uint32_t)tseq.size();
    const int64_t ta0 = tc.a0, ta1 = tc.a0 + (int64_t)tc.base.size();
    const uint32_t tid = (uint32_t)(o.off.size() - 1);
    // target prefix sums: bases before column i
    std::vector<uint32_t> tpre(tc.base.size() + 1, 0);
    for (size_t i = 0; i < tc.base.size(); i++)
      tpre[i + 1] = tpre[i] + (tc.base[i] != DEL) + tc.ins_n(i);
    {
      const uint64_t t_n = Rng::threshold(P.p_n_base);
      const size_t at = o.seq.size();
      o.seq.resize(at + tseq.size());
      for (size_t i = 0; i < tseq.size(); i++) o.seq[at + i] = (uint8_t)(g.bern_t(t_n) ? 'N' : ACGT[tseq[i]]);
    }
    gen_quals(g, tseq.size(), o.qual);
    o.off.push_back(o.seq.size());
    o.tgt_rid.push_back(tid);

    for (uint32_t k = 0; k < P.n_overlaps; k++) {
      const int qh = (int)(k & 1);
      int64_t lo = ta0 - (int64_t)(P.flank_min + g.below(P.flank_max - P.flank_min + 1));
      int64_t hi = ta1 + (int64_t)(P.flank_min + g.below(P.flank_max - P.flank_min + 1));
      if (g.bern(P.p_partial)) {  // overlap covering only part of the target
        const int64_t span = ta1 - ta0;
        const int64_t len = std::min<int64_t>(
            span, (int64_t)P.min_partial_len +
                      (int64_t)g.below((uint32_t)std::max<int64_t>(1, span - (int64_t)P.min_partial_len)));
        const int64_t st = ta0 + (int64_t)g.below((uint32_t)(span - len + 1));
        const uint32_t mode = g.below(3);
        if (mode == 0) { lo = st; hi = st + len; }   // internal
        else if (mode == 1) { hi = st + len; }       // keeps the left flank
        else { lo = st; }                            // keeps the right flank
      }
      lo = std::max<int64_t>(lo, 0);
      hi = std::min<int64_t>(hi, TL);
      Cols qc = noisy_read(g, hap[qh], lo, hi, P);
      std::vector<uint8_t> qseq;
      flatten(qc, qseq);
      // --- alignment over the shared columns
      const int64_t c0 = std::max(ta0, lo), c1 = std::min(ta1, hi);
      Cig cg;
      uint32_t q_before = 0;
      for (int64_t a = lo; a < c0; a++) {
        const size_t i = (size_t)(a - lo);
        q_before += (qc.base[i] != DEL) + qc.ins_n(i);
      }
      uint32_t tstart = tpre[(size_t)(c0 - ta0)], tend = tstart, qs = q_before, qe = q_before;
      cg.ops.reserve(512);
      for (int64_t a = c0; a < c1; a++) {
        const size_t ti = (size_t)(a - ta0), qi = (size_t)(a - lo);
        const bool tb = tc.base[ti] != DEL, qb = qc.base[qi] != DEL;
        if (tb && qb) cg.add('M', 1);
        else if (tb) cg.add('D', 1);
        else if (qb) cg.add('I', 1);
        const uint32_t x = tc.ins_n(ti), y = qc.ins_n(qi);
        if (x | y) {
          cg.add('M', std::min(x, y));
          if (x > y) cg.add('D', x - y);
          if (y > x) cg.add('I', y - x);
        }
        tend += tb + x;
        qe += qb + y;
      }
      // strip leading / trailing non-M ops (PAF coordinates shrink accordingly)
      while (!cg.ops.empty() && cg.ops.front().first != 'M') {
        if (cg.ops.front().first == 'I') qs += cg.ops.front().second; else tstart += cg.ops.front().second;
        cg.ops.erase(cg.ops.begin());
      }
      while (!cg.ops.empty() && cg.ops.back().first != 'M') {
        if (cg.ops.back().first == 'I') qe -= cg.ops.back().second; else tend -= cg.ops.back().second;
        cg.ops.pop_back();
      }
      const uint32_t qid = (uint32_t)(o.off.size() - 1);
      const uint32_t qlen = (uint32_t)qseq.size();
      const bool rev = g.bern(0.5);
      std::vector<uint8_t> qq;
      gen_quals(g, qseq.size(), qq);
      {
        const size_t at = o.seq.size(), n = qseq.size();
        o.seq.resize(at + n);
        o.qual.resize(at + n);
        uint8_t* ds = o.seq.data() + at;
        uint8_t* dq = o.qual.data() + at;
        if (rev) {
          for (size_t i = 0; i < n; i++) { ds[i] = (uint8_t)ACGT[3 - qseq[n - 1 - i]]; dq[i] = qq[n - 1 - i]; }
        } else {
          for (size_t i = 0; i < n; i++) ds[i] = (uint8_t)ACGT[qseq[i]];
          memcpy(dq, qq.data(), n);
        }
      }
      o.off.push_back(o.seq.size());
      if (cg.ops.empty()) continue;
      std::string cs;
      for (auto& op : cg.ops) { cs += std::to_string(op.second); cs += op.first; }
      const uint32_t pq0 = rev ? qlen - qe : qs, pq1 = rev ? qlen - qs : qe;
      const uint32_t rec[10] = {qid, qlen, pq0, pq1, rev ? 1u : 0u, tid, tlen, tstart, tend, (uint32_t)cs.size()};
      o.aln.insert(o.aln.end(), rec, rec + 10);
      o.cig_off.push_back(o.cig.size());
      o.cig.insert(o.cig.end(), cs.begin(), cs.end());
    }
    o.tgt_aln_off.push_back(o.cig_off.size());
  }
}

}  // namespace

extern "C" {

struct herro_synth_params {
  uint64_t seed;
  uint32_t n_targets, target_len, n_overlaps, flank_min, flank_max, min_partial_len;
  double p_sub, p_ins, p_del, p_long_indel, p_partial, p_n_base;
  double p_snp;  // 0: the default 2e-3
};

void* herro_synth_generate(const herro_synth_params* p) {
  Params P{p->seed, p->n_targets, p->target_len, p->n_overlaps, p->flank_min, p->flank_max,
           p->p_sub, p->p_ins, p->p_del, p->p_long_indel, p->p_partial, p->p_n_base,
           p->min_partial_len, p->p_snp > 0 ? p->p_snp : 2e-3};
  if (P.flank_max < P.flank_min) P.flank_max = P.flank_min;
  Out* o = new Out();
  generate(P, *o);
  return o;
}
void herro_synth_free(void* h) { delete (Out*)h; }
// sizes: n_reads, total_bases, n_alignments, cigar_bytes, n_targets
void herro_synth_sizes(void* h, uint64_t* s) {
  Out* o = (Out*)h;
  s[0] = o->off.size() - 1; s[1] = o->seq.size(); s[2] = o->cig_off.size(); s[3] = o->cig.size();
  s[4] = o->tgt_rid.size();
}
// One part of a batch generated in chunks, written straight into its place in the merged arrays (herro_amd/synth.py
// generate_parallel): read ids, base offsets, alignment and CIGAR offsets rebased by what the parts in front of it hold.
// seq / qual / cig / aln / cig_off / tgt_rid point at the part's first element; off and tgt_aln_off at the element BEFORE it
// (the merged arrays carry one leading 0, written by the caller).
void herro_synth_copy_into(void* h, uint8_t* seq, uint8_t* qual, uint64_t* off, uint32_t* aln, uint64_t* cig_off, uint8_t* cig,
                           uint64_t* tgt_aln_off, uint32_t* tgt_rid, uint64_t rbase, uint64_t bbase, uint64_t abase, uint64_t cbase) {
  Out* o = (Out*)h;
  memcpy(seq, o->seq.data(), o->seq.size());
  memcpy(qual, o->qual.data(), o->qual.size());
  for (size_t i = 1; i < o->off.size(); i++) off[i] = o->off[i] + bbase;
  const size_t na = o->cig_off.size();
  for (size_t a = 0; a < na; a++) {
    const uint32_t* r = &o->aln[a * 10];
    uint32_t* d = aln + a * 10;
    memcpy(d, r, 40);
    d[0] = r[0] + (uint32_t)rbase;   // qid
    d[5] = r[5] + (uint32_t)rbase;   // tid
    cig_off[a] = o->cig_off[a] + cbase;
  }
  memcpy(cig, o->cig.data(), o->cig.size());
  for (size_t t = 1; t < o->tgt_aln_off.size(); t++) tgt_aln_off[t] = o->tgt_aln_off[t] + abase;
  for (size_t t = 0; t < o->tgt_rid.size(); t++) tgt_rid[t] = o->tgt_rid[t] + (uint32_t)rbase;
}
void herro_synth_copy(void* h, uint8_t* seq, uint8_t* qual, uint64_t* off, uint32_t* aln,
                      uint64_t* cig_off, uint8_t* cig, uint64_t* tgt_aln_off, uint32_t* tgt_rid) {
  Out* o = (Out*)h;
  memcpy(seq, o->seq.data(), o->seq.size());
  memcpy(qual, o->qual.data(), o->qual.size());
  memcpy(off, o->off.data(), o->off.size() * 8);
  memcpy(aln, o->aln.data(), o->aln.size() * 4);
  memcpy(cig_off, o->cig_off.data(), o->cig_off.size() * 8);
  memcpy(cig, o->cig.data(), o->cig.size());
  memcpy(tgt_aln_off, o->tgt_aln_off.data(), o->tgt_aln_off.size() * 8);
  memcpy(tgt_rid, o->tgt_rid.data(), o->tgt_rid.size() * 4);
}
}
