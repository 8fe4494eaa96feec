// pool.cpp — one host process driving several contexts (GPUs) from ONE queue, behind the C ABI.
//
// The reference feeds its per-device replicas from one MPMC channel: `t` feature threads per device pull reads, one inference
// thread per device pulls their batches, whoever is free takes the next item (lib.rs:154-200); results go to one writer
// (lib.rs:267-291).  herro_amd/shard.py covers the several-PROCESSES layout (one rank per GPU, static ownership); this file is the
// in-process layout a Rust host would bind instead of spawning those threads itself: N contexts (any mix of devices; several on
// one device share its read store, herro_share_reads), one worker thread each, and a shared counter over groups of target reads —
// a worker that finishes early simply takes more groups (reads differ in length and depth: static partitions do not balance).
// Every worker keeps two jobs in flight: herro_job_create of group k+1 (host work + CIGAR scan) runs while the GPU works on k.
// Only calls of include/herro_amd.h are used: this is a client of the C ABI, not a second implementation.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/herro_amd.h"

// The per-job calls the workers make, as a table: the header's functions by default; herro_debug_pool_fake swaps in a device-free
// stand-in (below) so that the queue, the two-jobs-in-flight pipeline, the merge and the error paths can be driven on a machine
// without a GPU and with groups of very different cost (tests/test_pool_host.py).
struct PoolOps {
  herro_job* (*job_create)(herro_ctx*, uint32_t, const uint32_t*, const uint64_t*, const herro_alignment*, uint32_t);
  int (*job_create_status)(const herro_ctx*);
  int (*job_skipped)(const herro_job*, uint32_t*, uint32_t*);
  int (*job_featurize)(herro_job*);
  int (*job_infer)(herro_job*, uint32_t, int);
  int (*job_consensus)(herro_job*);
  int (*job_consensus_fetch)(herro_job*, uint64_t*);
  int64_t (*job_fasta)(herro_job*, const char* const*, const char* const*, char*, uint64_t, uint64_t*);
  void (*job_free)(herro_job*);
  const char* (*last_error)(const herro_ctx*);
};
static const PoolOps REAL_OPS = {herro_job_create, herro_job_create_status, herro_job_skipped, herro_job_featurize, herro_job_infer, herro_job_consensus,
                                 herro_job_consensus_fetch, herro_job_fasta, herro_job_free, herro_last_error};

struct herro_pool {
  std::vector<herro_ctx*> ctx;
  std::vector<int> dev;
  PoolOps ops = REAL_OPS;
  bool fake = false;
  std::string err;
  std::mutex err_mu;
  // result of the last herro_pool_correct
  std::vector<char> text;
  std::vector<uint64_t> rec_end;
  std::vector<uint32_t> groups_by_ctx;   // how many groups each context took (the dynamic hand-out, observable)
  std::atomic<uint64_t> skipped_alns{0}, skipped_targets{0};   // herro_job_skipped, summed over the groups of the last call
};

// ---- device-free stand-in for the per-job calls (herro_debug_pool_fake) ---------------------------------------------------------
// A fake context is a small record (worker index, microseconds per alignment); a fake job remembers its targets.  "Work" is a sleep
// proportional to the group's alignments — in herro_job_featurize's stand-in for the asynchronous part a real context would overlap,
// in herro_job_infer's for the part the worker waits for — so groups cost what their size says and contexts differ in speed.  The
// FASTA of target t is ">id\n" + (rid % 7 + 1) bases.  A target with rid == 0xfffffffe makes herro_job_create fail with
// HERRO_E_UNSUPPORTED, rid == 0xfffffffd makes the job's herro_job_infer fail with HERRO_E_STATE; a target whose first alignment has
// qid == tid counts as one skipped alignment.
namespace {
struct FakeCtx { uint32_t index; uint32_t us_per_aln; std::string err; int create_code = HERRO_OK; };
struct FakeJob { FakeCtx* c; std::vector<uint32_t> rids; uint64_t n_alns = 0; uint32_t skipped = 0; bool bad_infer = false; };
FakeCtx* fc(const herro_ctx* c) { return reinterpret_cast<FakeCtx*>(const_cast<herro_ctx*>(c)); }
FakeJob* fj(const herro_job* j) { return reinterpret_cast<FakeJob*>(const_cast<herro_job*>(j)); }
herro_job* fake_create(herro_ctx* c, uint32_t n, const uint32_t* rids, const uint64_t* off, const herro_alignment* alns, uint32_t) {
  FakeCtx* f = fc(c);
  f->create_code = HERRO_OK;
  auto j = new FakeJob{f, std::vector<uint32_t>(rids, rids + n)};
  j->n_alns = off[n] - off[0];
  for (uint32_t t = 0; t < n; t++) {
    if (rids[t] == 0xfffffffeu) { f->err = "fake: unsupported target [code -4]"; f->create_code = HERRO_E_UNSUPPORTED; delete j; return nullptr; }
    if (rids[t] == 0xfffffffdu) j->bad_infer = true;
    if (off[t + 1] > off[t] && alns && alns[off[t]].qid == alns[off[t]].tid) j->skipped++;
  }
  return reinterpret_cast<herro_job*>(j);
}
int fake_create_status(const herro_ctx* c) { return fc(c)->create_code; }
int fake_skipped(const herro_job* j, uint32_t* a, uint32_t* t) { if (a) *a = fj(j)->skipped; if (t) *t = 0; return HERRO_OK; }
int fake_featurize(herro_job*) { return HERRO_OK; }
int fake_infer(herro_job* j, uint32_t, int) {
  FakeJob* f = fj(j);
  std::this_thread::sleep_for(std::chrono::microseconds((uint64_t)f->c->us_per_aln * std::max<uint64_t>(f->n_alns, 1)));
  if (f->bad_infer) { f->c->err = "fake: infer failed"; return HERRO_E_STATE; }
  return HERRO_OK;
}
int fake_consensus(herro_job*) { return HERRO_OK; }
int fake_fetch(herro_job*, uint64_t*) { return HERRO_OK; }
int64_t fake_fasta(herro_job* j, const char* const* ids, const char* const*, char* out, uint64_t cap, uint64_t* ends) {
  FakeJob* f = fj(j);
  std::string s;
  for (size_t t = 0; t < f->rids.size(); t++) {
    s += ">"; s += ids[t]; s += "\n"; s.append(f->rids[t] % 7 + 1, "ACGT"[f->rids[t] & 3u]); s += "\n";
    if (ends) ends[t] = s.size();
  }
  if (out) { if (s.size() > cap) return HERRO_E_INVALID; memcpy(out, s.data(), s.size()); }
  return (int64_t)s.size();
}
void fake_free(herro_job* j) { delete fj(j); }
const char* fake_last_error(const herro_ctx* c) { return fc(c)->err.c_str(); }
const PoolOps FAKE_OPS = {fake_create, fake_create_status, fake_skipped, fake_featurize, fake_infer, fake_consensus, fake_fetch, fake_fasta, fake_free, fake_last_error};
}  // namespace

namespace {
void set_err(herro_pool* p, const std::string& m) {
  std::lock_guard<std::mutex> lk(p->err_mu);
  if (p->err.empty()) p->err = m;
}
}  // namespace

extern "C" {

herro_pool* herro_pool_create(const int* device_ids, uint32_t n_ctx) {
  if (!device_ids || n_ctx == 0) return nullptr;
  auto p = new herro_pool();
  for (uint32_t i = 0; i < n_ctx; i++) {
    herro_ctx* c = herro_create(device_ids[i]);
    if (!c) {
      for (herro_ctx* o : p->ctx) herro_destroy(o);
      delete p;
      return nullptr;   // herro_last_error(NULL) has the reason
    }
    p->ctx.push_back(c);
    p->dev.push_back(device_ids[i]);
  }
  p->groups_by_ctx.assign(n_ctx, 0);
  return p;
}

// device-free pool for tests: n_ctx stand-in contexts, context i "works" us_per_aln[i] microseconds per alignment of a group
herro_pool* herro_debug_pool_fake(uint32_t n_ctx, const uint32_t* us_per_aln) {
  if (n_ctx == 0 || !us_per_aln) return nullptr;
  auto p = new herro_pool();
  p->fake = true;
  p->ops = FAKE_OPS;
  for (uint32_t i = 0; i < n_ctx; i++) {
    p->ctx.push_back(reinterpret_cast<herro_ctx*>(new FakeCtx{i, us_per_aln[i]}));
    p->dev.push_back((int)i);
  }
  p->groups_by_ctx.assign(n_ctx, 0);
  return p;
}

void herro_pool_destroy(herro_pool* p) {
  if (!p) return;
  for (herro_ctx* c : p->ctx) { if (p->fake) delete fc(c); else herro_destroy(c); }
  delete p;
}

const char* herro_pool_last_error(const herro_pool* p) { return p ? p->err.c_str() : "null pool"; }
uint32_t herro_pool_size(const herro_pool* p) { return p ? (uint32_t)p->ctx.size() : 0; }
herro_ctx* herro_pool_ctx(herro_pool* p, uint32_t i) { return p && !p->fake && i < p->ctx.size() ? p->ctx[i] : nullptr; }
int herro_pool_skipped(const herro_pool* p, uint64_t* n_alignments, uint64_t* n_targets) {
  if (!p) return HERRO_E_INVALID;
  if (n_alignments) *n_alignments = p->skipped_alns.load();
  if (n_targets) *n_targets = p->skipped_targets.load();
  return HERRO_OK;
}

int herro_pool_set_reads(herro_pool* p, uint32_t n_reads, const uint8_t* seq, const uint8_t* qual, const uint64_t* off, const uint32_t* name_class) {
  if (!p || p->fake) return HERRO_E_INVALID;
  { std::lock_guard<std::mutex> lk(p->err_mu); p->err.clear(); }
  for (size_t i = 0; i < p->ctx.size(); i++) {
    int first = -1;   // the first context of the same device holds the store; the others adopt it
    for (size_t j = 0; j < i; j++) if (p->dev[j] == p->dev[i]) { first = (int)j; break; }
    const int rc = first < 0 ? herro_set_reads(p->ctx[i], n_reads, seq, qual, off, name_class) : herro_share_reads(p->ctx[i], p->ctx[first]);
    if (rc != HERRO_OK) { set_err(p, herro_last_error(p->ctx[i])); return rc; }
  }
  return HERRO_OK;
}

int herro_pool_load_model(herro_pool* p, const char* path) {
  if (!p || p->fake) return HERRO_E_INVALID;
  { std::lock_guard<std::mutex> lk(p->err_mu); p->err.clear(); }
  for (herro_ctx* c : p->ctx) {
    const int rc = herro_load_model(c, path);
    if (rc != HERRO_OK) { set_err(p, herro_last_error(c)); return rc; }
  }
  return HERRO_OK;
}

int64_t herro_pool_correct(herro_pool* p, uint32_t n_targets, const uint32_t* rids, const uint64_t* aln_off, const herro_alignment* alns,
                           uint32_t window_size, uint32_t batch_size, int batch_mode, uint32_t group_targets, const char* const* ids,
                           const char* const* descs) {
  if (!p || (n_targets && (!rids || !aln_off || !ids)) || batch_size == 0) return HERRO_E_INVALID;
  { std::lock_guard<std::mutex> lk(p->err_mu); p->err.clear(); }   // (no worker is running here; the lock keeps the rule simple: err is only touched under err_mu)
  p->skipped_alns = 0; p->skipped_targets = 0;
  const PoolOps& op = p->ops;
  p->text.clear();
  p->rec_end.assign(n_targets, 0);
  std::fill(p->groups_by_ctx.begin(), p->groups_by_ctx.end(), 0u);
  if (n_targets == 0) return 0;
  group_targets = std::max(1u, group_targets);
  const uint32_t n_groups = (n_targets + group_targets - 1) / group_targets;
  struct Part { std::vector<char> text; std::vector<uint64_t> ends; };
  std::vector<Part> parts(n_groups);
  std::atomic<uint32_t> next{0};
  std::atomic<int> failed{HERRO_OK};
  auto worker = [&](uint32_t k) {
    herro_ctx* c = p->ctx[k];
    struct Flight { herro_job* job = nullptr; uint32_t g = 0; };
    Flight prev;
    auto fail = [&](int rc) { int z = HERRO_OK; failed.compare_exchange_strong(z, rc); set_err(p, op.last_error(c)); };
    auto finish = [&](Flight& f) {
      if (!f.job) return;
      const uint32_t t0 = f.g * group_targets, t1 = std::min(n_targets, t0 + group_targets);
      int rc = op.job_infer(f.job, batch_size, batch_mode);
      if (rc == HERRO_OK) rc = op.job_consensus(f.job);
      if (rc == HERRO_OK) rc = op.job_consensus_fetch(f.job, nullptr);
      if (rc == HERRO_OK) {
        Part& pt = parts[f.g];
        pt.ends.assign(t1 - t0, 0);
        const int64_t need = op.job_fasta(f.job, ids + t0, descs ? descs + t0 : nullptr, nullptr, 0, pt.ends.data());
        if (need < 0) rc = (int)need;
        else {
          pt.text.resize((size_t)need);
          const int64_t got = op.job_fasta(f.job, ids + t0, descs ? descs + t0 : nullptr, pt.text.data(), (uint64_t)need, nullptr);
          if (got != need) rc = got < 0 ? (int)got : HERRO_E_STATE;
        }
      }
      if (rc != HERRO_OK) fail(rc);
      op.job_free(f.job);
      f.job = nullptr;
    };
    while (failed.load() == HERRO_OK) {
      const uint32_t g = next.fetch_add(1);           // the shared queue: whoever is free takes the next group
      if (g >= n_groups) break;
      p->groups_by_ctx[k]++;
      const uint32_t t0 = g * group_targets, t1 = std::min(n_targets, t0 + group_targets);
      std::vector<uint64_t> off(aln_off + t0, aln_off + t1 + 1);
      const uint64_t a0 = off[0];
      for (uint64_t& o : off) o -= a0;
      Flight cur;
      cur.g = g;
      cur.job = op.job_create(c, t1 - t0, rids + t0, off.data(), alns + a0, window_size);
      if (!cur.job) {   // the context's own verdict (UNSUPPORTED, INVALID, NO_DEVICE, the reference's panic), not a blanket code
        const int code = op.job_create_status(c);
        fail(code != HERRO_OK ? code : HERRO_E_STATE);
        break;
      }
      uint32_t sk_a = 0, sk_t = 0;
      if (op.job_skipped(cur.job, &sk_a, &sk_t) == HERRO_OK) {   // summed for herro_pool_skipped: the caller decides (parse_paf drops the same alignments; shard.py treats one as an error)
        p->skipped_alns += sk_a;
        p->skipped_targets += sk_t;
      }
      const int rc = op.job_featurize(cur.job);
      if (rc != HERRO_OK) { fail(rc); op.job_free(cur.job); break; }
      finish(prev);                                    // the GPU works on `cur` while the previous group is finished ... and the next one created
      prev = cur;
    }
    finish(prev);
  };
  std::vector<std::thread> th;
  for (uint32_t k = 0; k < p->ctx.size(); k++) th.emplace_back(worker, k);
  for (auto& t : th) t.join();
  if (failed.load() != HERRO_OK) return failed.load();
  // target order = group order
  uint64_t total = 0;
  for (const Part& pt : parts) total += pt.text.size();
  p->text.resize(total);
  uint64_t at = 0;
  for (uint32_t g = 0; g < n_groups; g++) {
    const Part& pt = parts[g];
    if (!pt.text.empty()) memcpy(p->text.data() + at, pt.text.data(), pt.text.size());
    for (size_t i = 0; i < pt.ends.size(); i++) p->rec_end[(size_t)g * group_targets + i] = at + pt.ends[i];
    at += pt.text.size();
  }
  return (int64_t)total;
}

const char* herro_pool_result(const herro_pool* p, const uint64_t** rec_end) {
  if (!p) return nullptr;
  if (rec_end) *rec_end = p->rec_end.data();
  return p->text.data();
}

uint32_t herro_pool_groups_taken(const herro_pool* p, uint32_t i) { return p && i < p->groups_by_ctx.size() ? p->groups_by_ctx[i] : 0; }

}  // extern "C"
