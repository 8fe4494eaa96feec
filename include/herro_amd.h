/* herro_amd.h — C ABI of the MI355X-native HERRO hot path (libherro_amd.so).
 *
 * Scope: the per-window pileup feature generation (reference src/features.rs) and the
 * correction-model forward (reference src/inference.rs) of lbcb-sci/herro v0.1.1, as
 * hand-written HIP kernels for gfx950.  The reference has no FFI of its own; the entry
 * points below are what a cgo/Rust-FFI binding for this path would bind — each one cites
 * the reference interface it replaces.  Plain pointers and sizes only; no torch types.
 *
 * Conventions: every function returning int returns HERRO_OK (0) or a negative error code;
 * the text of the last error is available from herro_last_error().  Where the reference
 * would panic (Cargo.toml:18 panic = "abort") this library returns HERRO_E_REFERENCE_PANIC.
 * Nothing here ever falls back to a CPU implementation: without a HIP device every
 * device-touching call fails with HERRO_E_NO_DEVICE.
 */
#ifndef HERRO_AMD_H
#define HERRO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HERRO_OK 0
#define HERRO_E_INVALID (-1)          /* bad argument */
#define HERRO_E_NO_DEVICE (-2)        /* no HIP device / HIP runtime error */
#define HERRO_E_REFERENCE_PANIC (-3)  /* input on which the reference itself panics */
#define HERRO_E_UNSUPPORTED (-4)      /* legal for the reference, not supported here */
#define HERRO_E_NO_MODEL (-5)
#define HERRO_E_STATE (-6)            /* call order violated */

#define HERRO_ROWS 31      /* target + TOP_K_SORT(30) overlaps — features.rs:22 */
#define HERRO_N_CLASSES 5  /* A C G T * — consensus.rs:142-149 */

typedef struct herro_ctx herro_ctx;
typedef struct herro_job herro_job;

/* One PAF record + its cg:Z: CIGAR — reference overlaps.rs:45-55 (Overlap) and :92-95
 * (Alignment).  strand: 0 = '+', 1 = '-'.  cigar: ASCII "\d+[MID]" (aligners.rs:252-293). */
typedef struct {
  uint32_t qid, qlen, qstart, qend;
  uint32_t strand;
  uint32_t tid, tlen, tstart, tend;
  uint32_t cigar_len;
  const uint8_t* cigar;
} herro_alignment;

/* Per-window result header — the fields of WindowExample / ConsensusWindow that the path
 * hands on (inference.rs:270-278, consensus.rs:22-33). */
typedef struct {
  uint32_t rid;          /* target read */
  uint32_t wid;          /* window index within the read */
  uint32_t n_total_wins; /* windows of that read */
  uint32_t length;       /* L' — rows after all-gap-row removal (features.rs:531-556) */
  uint32_t n_alns;       /* min(#ranked overlaps, 30) (features.rs:877) */
  uint32_t n_overlaps;   /* #ranked overlaps ("ids", features.rs:569) */
  uint32_t n_supported;  /* informative positions (features.rs:558) */
  uint32_t win_len;      /* target bases in the window */
} herro_window_info;

/* ---- library / context ------------------------------------------------------------------ */
const char* herro_version(void);
/* One context per GPU (the reference runs one worker set per `-d` device, lib.rs:154-200). */
herro_ctx* herro_create(int device_id);
void herro_destroy(herro_ctx* ctx);
const char* herro_last_error(const herro_ctx* ctx); /* ctx may be NULL (creation errors) */
/* Run all subsequent work of this context on an externally owned hipStream_t (e.g. the
 * current torch stream).  NULL restores the context's own stream. */
int herro_set_stream(herro_ctx* ctx, void* hip_stream);
int herro_synchronize(herro_ctx* ctx);

/* ---- 2-bit codec — haec_io.rs:121-173 (host utility, no device) --------------------------- */
/* words must hold (n+31)/32 u64.  Returns word count, or HERRO_E_REFERENCE_PANIC for a
 * byte >= 128 (the reference indexes a 128-entry table). */
int64_t herro_encode_2bit(const uint8_t* seq, uint64_t n, uint64_t* words);
int herro_decode_2bit(const uint64_t* words, uint64_t length, uint64_t start, uint64_t end,
                      int reverse_complement, uint8_t* out);

/* ---- device-resident read store — replaces `reads: &[HAECRecord]` (lib.rs:133, haec_io.rs:19-24).
 * Read i: bases seq[off[i]..off[i+1]) (ASCII; packed to 2 bit on the host exactly as
 * haec_io.rs:121-136 incl. the non-ACGT quirk), quals the same slice of `qual` (phred+33 bytes).
 * name_class (nullable): name_class[i] == name_class[j] iff reads i and j have the same id
 * string (the reference keys the haplotype ratios by read *name*, features.rs:494); NULL means
 * all names are distinct. */
int herro_set_reads(herro_ctx* ctx, uint32_t n_reads, const uint8_t* seq, const uint8_t* qual,
                    const uint64_t* off, const uint32_t* name_class);
/* Same, from already packed words (HAECSeq.data, haec_io.rs:78-81): word_off[n_reads+1]. */
int herro_set_reads_packed(herro_ctx* ctx, uint32_t n_reads, const uint64_t* words,
                           const uint64_t* word_off, const uint8_t* qual, const uint64_t* qual_off,
                           const uint32_t* name_class);
/* One read store per DEVICE: `ctx` adopts the store `from` holds (same device) instead of packing and uploading its own copy —
 * the reference's `reads: &[HAECRecord]` is likewise shared read-only by all feature threads (lib.rs:159-187).  The device memory
 * is reference-counted and freed with its last holder; a later herro_set_reads on either context gives that context a store of
 * its own again.  Jobs of `ctx` built on its previous store refuse to run, as after herro_set_reads. */
int herro_share_reads(herro_ctx* ctx, const herro_ctx* from);

/* ---- model ----------------------------------------------------------------------------------
 * Flat little-endian weight file written by tools/export_weights.py (stands in for
 * tch::CModule::load_on_device, inference.rs:185). */
int herro_load_model(herro_ctx* ctx, const char* path);
/* Operand format of the model GEMMs (accumulation is f32 in every mode; the contract is |logit error| <= 1e-3):
 *   0  f32 MFMA (exact f32)                                   2  f32 VALU (debug reference)
 *   1  bf16 hi/lo split of both operands, 3 MFMAs per product (~1e-5), transformer stack fused into one kernel
 *   3  the same arithmetic, layer by layer (what windows with > 64 informative rows run in mode 1, and windows with > 512 in modes 4, 5)
 *   4  f16: conv2 / FC / attention / QKV on single f16 operands, proj / FF1 / FF2 of every encoder layer on activation
 *      hi + lo (2 MFMAs), heads on three terms — 5.9e-4 max on 36 k rows; the DEFAULT when the model has the tuned shapes.
 *      Windows of 65 .. 512 informative rows stay on the fused stack (sibling tiles of 64 rows that exchange their K / V).
 *   5  f16, single terms everywhere but the heads (8.3e-4 end to end on random-init weights)
 *   6  as 4, with the activation remainder of proj / FF1 / FF2 as OCP e4m3 against an e4m3 copy of the weight on the K = 128 scaled MFMA
 *      (v_mfma_scale_f32_16x16x128_f8f6f4; 6.1e-4 end to end; 26 % less matrix-pipe time in those GEMMs and no faster on an MI355X — DESIGN.md §5 —
 *      so never chosen by the library; calibrated on demand)
 *   7  as 4 with FF1 / FF2 on single terms (proj keeps its two): 13 MFMA call-terms per encoder layer instead of 21
 *   8  as 4 with proj on a single term (FF1 / FF2 keep two): 20 call-terms
 * herro_load_model picks the mode itself unless this was called before: 1 when the model's shapes have no f16 kernels or
 * a weight lies outside the f16 range; otherwise it runs a calibration batch of 256 pileup-shaped rows in mode 0 (f32 MFMA) and in
 * the f16 tiers 5, 7, 8, 4 — cheapest first — and keeps the FIRST whose logits are finite and within 5e-4 (half the 1e-3 contract)
 * of mode 0; none: mode 1.  The margin of the f16 formats was measured on random-init weights; a trained model decides for itself.
 * herro_set_precision(4 .. 8) is held to the same bound — the mode is calibrated here if the load did not measure it — and refused
 * above it (HERRO_E_UNSUPPORTED; HERRO_FORCE_PRECISION=1 in A/B builds overrides); a mode set BEFORE herro_load_model is calibrated by
 * the load and replaced by mode 1 when it fails.  herro_model_describe reports the outcome, herro_precision the mode in force,
 * herro_calibration_error what a mode measured (-1: not measured). */
int herro_set_precision(herro_ctx* ctx, int mode);
int herro_precision(const herro_ctx* ctx);
/* Measurement aid: the shader clock (MHz) the device runs at right behind the work queued on the context's stream — one wave reads the
 * shader-clock and the constant-rate counters around ~40 us of dependent adds (synchronises the stream).  bench.py's `sustained` leg samples it. */
int herro_clock_probe(herro_ctx* ctx, double* shader_mhz);
float herro_calibration_error(const herro_ctx* ctx, int mode);

/* Text description of the loaded model: hyper-parameters, receptive field of an informative row (rows the conv stack
 * evaluates per token), GEMM FLOP per token and per 4096-bp window at 15 informative rows, the precision mode in force and
 * the calibration result of herro_load_model.  Returns the length of the text (truncated to cap - 1 bytes + NUL). */
int64_t herro_model_describe(const herro_ctx* ctx, char* out, uint64_t cap);

/* ---- job = a set of target reads with their alignments -------------------------------------
 * herro_job_create replaces the front half of `extract_features` (features.rs:326-361): `extract_windows`
 * (windowing.rs:44-273) for every alignment.  The CIGAR text is staged in pinned memory and decoded ON THE DEVICE (one
 * kernel, a workgroup per alignment: binary ops into the job's op array + the ops that reach a window boundary back to the
 * host); the host cuts the windows from those records, validates, and uploads the descriptors (one pinned block, one
 * asynchronous copy).  The call waits for that one short kernel on a separate high-priority stream of the context, not
 * for the context's execution stream.  Job memory is recycled through per-context arenas; the host work runs on a
 * per-context thread pool of HERRO_HOST_THREADS, default min(hardware threads / 4, 64, the cgroup CPU quota).
 * HERRO_HOST_SCAN=1 decodes the text on the host instead (same results; the device-free hook below always does).
 * Threading: a context's execution calls (featurize / infer / consensus / accessors) belong to one thread; herro_job_create
 * (host work + one asynchronous upload) may run on a SECOND thread of the same context at the same time, so that the next
 * job is built while the current one executes (bench.py end_to_end does this).
 * Alignments of target t are
 * alns[aln_off[t] .. aln_off[t+1]); every alignment must have tid == rids[t] (overlaps.rs:189-192).
 * window_size: the `-w` flag (main.rs:69-74); 16 <= window_size <= 8192 here. */
herro_job* herro_job_create(herro_ctx* ctx, uint32_t n_targets, const uint32_t* rids,
                            const uint64_t* aln_off, const herro_alignment* alns,
                            uint32_t window_size);
void herro_job_free(herro_job* job);
/* Zero-copy job creation.  herro_job_create has to get the CIGAR text of a job's alignments to the GPU; by default it gathers the texts
 * into a pinned staging block first.  A caller whose alignments' `cigar` pointers lie in ONE buffer — the PAF text behind
 * herro_paf_parse_view, a blob of CIGARs — registers that buffer once (it is pinned for the GPU, hipHostRegister): jobs whose texts lie
 * densely inside a registered range are then copied up straight from it, one copy, no pass over the bytes on the host.  Process-wide
 * and counted: several contexts may register the same range; unregister before the memory is freed (herro_host_unregister waits for the copies of
 * EVERY device and accepts ctx == NULL: the range outlives no particular context).  A range that cannot be pinned (the memlock limit) returns
 * HERRO_E_UNSUPPORTED and changes nothing: its jobs are staged.  (The reference has no counterpart:
 * its feature threads walk the CIGARs where the parser left them, features.rs:337-361.) */
int herro_host_register(herro_ctx* ctx, const void* p, uint64_t bytes);
int herro_host_unregister(herro_ctx* ctx, const void* p);
uint64_t herro_debug_zero_copy_jobs(void); /* test hook: jobs of this process created through the zero-copy path so far */
/* The HERRO_E_* code behind the last herro_job_create of this context that returned NULL (HERRO_OK after a successful one):
 * UNSUPPORTED / INVALID / STATE / NO_DEVICE / REFERENCE_PANIC — herro_last_error has the text. */
int herro_job_create_status(const herro_ctx* ctx);
uint32_t herro_job_n_windows(const herro_job* job);
/* Alignments herro_job_create left out instead of failing the call: exactly what parse_paf itself drops before extract_features
 * sees it — self overlaps and a second alignment of a (query, target) pair (overlaps.rs:175-185).  Everything else the reference
 * processes is processed (consecutive insertion ops, an alignment starting inside window 0 with an insertion, any number of
 * overlaps per window); *n_targets is always 0 since round 3 (no target is dropped any more) and kept for the binding.
 * herro_last_error() describes the first alignment left out.  Inputs on which the reference panics make herro_job_create
 * return NULL with the reference's message.  Callers that feed alignments straight from parse_paf / herro_paf_parse never see a
 * non-zero count; herro_amd/shard.py treats one as an error. */
int herro_job_skipped(const herro_job* job, uint32_t* n_alignments, uint32_t* n_targets);

/* GPU feature generation for all windows of the job — features.rs:364-580 (filter, accuracy
 * ranking, max-insertion map, pileup scatter, informative positions, haplotype-ratio re-ranking,
 * top-30 selection, all-gap-row removal, token encoding inference.rs:222).  Asynchronous on the
 * context stream; the accessors below synchronise. */
int herro_job_featurize(herro_job* job);

/* Model forward over the job's windows with >=1 informative position, batched in submission
 * order in chunks of batch_size (prepare_examples/collate/inference, inference.rs:73-175,214-253).
 * batch_mode 0: batches never span reads (the reference's grouping, features.rs:884-893);
 * batch_mode 1: windows of different reads share batches (BASELINE batch=64/128 configs). */
int herro_job_infer(herro_job* job, uint32_t batch_size, int batch_mode);

/* Results (synchronise first).  Window order = (target order, wid). */
int herro_job_window_info(herro_job* job, uint32_t w, herro_window_info* info);
/* bases/quals: [length, 31] row-major as the reference emits them (features.rs:547-556);
 * bases are ASCII when encoded == 0, BASES_MAP tokens (inference.rs:23-31) when encoded != 0.
 * sup_pos/sup_ins: SupportedPos (features.rs:896-900); qids: ranked overlap read ids.
 * Any output pointer may be NULL. */
int herro_job_window_copy(herro_job* job, uint32_t w, int encoded, uint8_t* bases, uint8_t* quals,
                          uint16_t* sup_pos, uint8_t* sup_ins, uint32_t* qids);
/* Logits of window w after herro_job_infer: info[n_supported], bases[n_supported*5]. */
int herro_job_window_logits(herro_job* job, uint32_t w, float* info_logits, float* bases_logits);

/* Consensus on the device (consensus.rs:86-227): per window, the corrected bases (informative rows:
 * argmax of the base logits; other rows: majority vote with target tie-break; '*' dropped) are left in
 * HBM; herro_job_consensus_fasta then only concatenates windows.  Optional: without it the FASTA call
 * decodes on the host from the planes.  Requires herro_job_infer if any window has informative rows. */
int herro_job_consensus(herro_job* job);

/* After herro_job_consensus: copy the corrected bases of every window to the host (the only result traffic of the
 * corrected-reads path, ~4 KB per window) and return how many there are; herro_job_consensus_fasta then only assembles
 * text.  Optional (the FASTA call fetches on demand). */
int herro_job_consensus_fetch(herro_job* job, uint64_t* n_bases);

/* Consensus + FASTA text for target t (consensus.rs:86-227, lib.rs:282-317); id/desc are the
 * read's id and optional description (NULL: none).  Returns bytes written (0: read not
 * emitted), or a negative error.  Host-side integer decode of device results. */
int64_t herro_job_consensus_fasta(herro_job* job, uint32_t t, const char* id, const char* desc,
                                  char* out, uint64_t cap);

/* The FASTA records of ALL targets of the job in one call, target order (the consensus worker's output for a batch of
 * reads, consensus.rs:229-263 + lib.rs:282-317): ids[t] / descs[t] (descs or descs[t] NULL: none) as above.  Requires
 * herro_job_consensus.  out == NULL: returns the number of bytes the records take; otherwise writes them (at most cap
 * bytes) and returns the bytes written, or a negative error.  rec_end, if not NULL, receives for every target the end
 * offset of its records in out ([n_targets]; equal ends: the read was not emitted). */
int64_t herro_job_fasta(herro_job* job, const char* const* ids, const char* const* descs, char* out, uint64_t cap,
                        uint64_t* rec_end);

/* ---- stand-alone model entry (parity checks) — mirrors `inference` (inference.rs:147-175):
 * bases u8 tokens [B,L,31], quals raw u8 [B,L,31] (normalised on device, :153), lens [B],
 * indices flat [sum(lens)].  Outputs info_logits [N], bases_logits [N,5] (host). */
int herro_model_forward(herro_ctx* ctx, uint32_t B, uint32_t L, const uint8_t* bases,
                        const uint8_t* quals, const int32_t* lens, const int32_t* indices,
                        float* info_logits, float* bases_logits);

/* ---- measurement hooks (bench.py) -------------------------------------------------------- */
/* Per-kernel GPU time accumulated with HIP events on the context stream since the last reset.
 * names: '\n'-separated kernel-group names; ms/calls: arrays of n entries. */
int herro_timing_enable(herro_ctx* ctx, int on);
int herro_timing_reset(herro_ctx* ctx);
int herro_timing_get(herro_ctx* ctx, char* names, uint64_t names_cap, double* ms, uint64_t* calls,
                     uint32_t* n);
/* Algorithmic bytes of the job's featurisation (SURVEY.md §8 d formula), measured on the data:
 * out[0]=2-bit+qual bytes read, out[1]=cigar-op bytes, out[2]=model-ready bytes written,
 * out[3]=sum L', out[4]=sum n_supported, out[5]=windows with n_supported>0. */
int herro_job_stats(herro_job* job, uint64_t* out);

/* ---- host-only test hook: the library's windowing (windowing.rs:44-273 restated on binary ops)
 * for one alignment; needs no device.  out rows of 8 u64: window, tstart, qstart, qend, op_lo,
 * op_hi (op-index slice), cigar_start_offset, cigar_end_offset.  Returns row count or <0. */
int64_t herro_debug_extract_windows(const herro_alignment* a, uint32_t n_windows, uint32_t window_size,
                                    uint64_t* out, uint64_t cap, char* err, uint64_t err_cap);

/* Host-only test hooks for the host half of herro_job_create (CIGAR parsing, windowing, validation, descriptor
 * layout): herro_debug_host_ctx makes a context without a device from the read lengths (name_class may be NULL);
 * herro_job_create on it returns a job that holds the descriptors only (free it with herro_job_free, the context with
 * herro_destroy; no other call is valid on them); herro_debug_job_array exposes them: which = 0 binary ops (u32),
 * 1 overlap-window descriptors, 2 window descriptors (layouts: csrc/pileup_core.h), 3 / 4 tile -> window / first row,
 * 5 first window of every target.  Returns the element count (<0: error). */
herro_ctx* herro_debug_host_ctx(uint32_t n_reads, const uint32_t* read_len, const uint32_t* name_class);
int64_t herro_debug_job_array(herro_job* job, int which, const void** ptr, uint32_t* elem_bytes);

/* Test hooks for the two feature-generation paths.  herro_job_featurize runs the LEAN path by default (round 5): informative rows,
 * decoder votes and the model's receptive fields are derived from the column bit planes in position space and the [31][L'] token
 * planes (features.rs:547-556) are only built when somebody asks for them (herro_job_window_copy, the features writer, a model
 * whose receptive field exceeds 8 rows).  herro_debug_set_featurize_planes(ctx, 1) selects the planes path of rounds 3-4 for the
 * jobs featurized from then on (environment: HERRO_FEATURIZE_PLANES=1) — two independent derivations of the same results, compared
 * by tests/test_gpu_lean.py.  herro_debug_job_rf copies the 16-byte receptive-field records of window w (after herro_job_infer:
 * n_supported x 31 records, bytes 0..7 tokens / 8..15 qualities of rows sup_row - half .. ) and returns their number (<0: error). */
int herro_debug_set_featurize_planes(herro_ctx* ctx, int on);
/* Test hooks for the two ways a job's windows and descriptors are built (herro_job_create): on the device behind the CIGAR scan (csrc/build_dev.hip, the
 * default since round 6) or by the host from the scan's cut records (rounds 3-5; still what runs when the device meets anything it does not settle itself —
 * a text the scan kernel flags, an input the reference panics on — and what words the error).  herro_debug_set_host_build(ctx, 1) selects the host build for
 * the jobs created from then on; herro_debug_job_dev_built says which one built a job.  tests/test_gpu_build_dev.py compares their descriptor arrays. */
int herro_debug_set_host_build(herro_ctx* ctx, int on);
int herro_debug_job_dev_built(const herro_job* job);
/* Host-only test hooks for the rules k_rows applies in position space (csrc/pileup_core.h): n count vectors counts[i][5] (A C G T * on a base row, the
 * target's base included; split[i][5] of them, or NULL, go through a second counter set that is merged in) with target[i] in 0..3 -> sup[i] (informative:
 * two symbols reach 3, features.rs:558,712) and vote[i] (consensus.rs:178-200; meaningful where sup[i] == 0); herro_debug_vote5: the vote on exact counts. */
int herro_debug_base_row_votes(const uint8_t* counts, const uint8_t* split, const uint8_t* target, uint32_t n, uint8_t* sup, uint8_t* vote);
uint32_t herro_debug_vote5(const uint32_t* c5, uint32_t tb);
int64_t herro_debug_job_rf(herro_job* job, uint32_t w, uint8_t* out, uint64_t cap);
int herro_debug_job_rf_fused(const herro_job* job);   /* 1: the last herro_job_infer read records k_rows gathered itself; 0: k_rfq's */
int herro_debug_job_rf_left(const herro_job* job);    /* ... of which this many windows (above the 256 informative rows k_rows stages) were filled by k_rfq behind it, the others staying fused */
uint32_t herro_debug_e4m3(float x);                   /* host f32 -> OCP e4m3 (round to nearest even, saturating) as used for the precision-6 weight copies */
int herro_debug_sib_fault(herro_ctx* ctx);            /* raises the context's sibling-tile error word as a tile that timed out would: the next fetch repeats its job's model pass without sibling tiles */
int herro_debug_force_precision(herro_ctx* ctx, int on); /* tests: herro_set_precision / herro_load_model skip the calibration gate (to MEASURE a mode the model's calibration refuses) */
int herro_debug_sib_retries(const herro_ctx* ctx);    /* model passes repeated that way on this context */

/* Host-only test hook for the token-tile plan of the fused transformer stack (herro_job_infer): n windows of cnt[i]
 * informative rows (1..64) -> order[k] = the window that is k-th in the launch's token stream; returns the number of
 * 64-token tiles of whole windows at the head of the stream.  packed bit 0: the best-fit-decreasing order herro_job_infer
 * uses (0: the given order); bits 1-2: 0 = 64-token tiles only; 1 = the f16 stack's default: when the last round of n_cu
 * tiles (one per compute unit) would be at most half full, its windows go into 32-token tiles instead (half the cost each);
 * 2 = every window of <= 32 rows in 32-token tiles, a window of 33..64 rows opens a 64-token tile and takes the best-fitting
 * small windows along.  *n_half (may be null) receives the number of 32-token tiles; tile_tok (capacity n + 1, may be null)
 * the first token of every tile, the 64-token ones first, then the end of the stream. */
int64_t herro_debug_tile_plan(const uint32_t* cnt, uint32_t n, int packed, uint32_t n_cu, uint32_t* order, uint32_t* n_half, uint32_t* tile_tok);
/* The same with windows above 64 informative rows admitted (cnt[i] in 1..512).  The f16 stack keeps such a window fused: on
 * ceil(rows / 64) consecutive 64-token tiles at the head of the stream — sibling tiles, which exchange the K / V fragments of
 * their heads layer by layer (inference.rs:134-141 takes any `lens`; windows above 512 rows run layer by layer).  The window's
 * last tile is filled up with the best-fitting small windows (packed plans only).  n_tiles[3] = sibling tiles, 64-token tiles,
 * 32-token tiles; tile_tok (capacity sum(ceil(cnt / 64)) + 2, may be null) the first token of every tile in stream order + end;
 * grp (capacity = the sibling tiles, may be null): first tile of the window's group | tiles in it << 20 | tokens of the window
 * in the group's last tile << 24.  The sibling tiles count as busy compute units when the last round is sized. */
int herro_debug_tile_plan_sib(const uint32_t* cnt, uint32_t n, int packed, uint32_t n_cu, uint32_t* order, uint32_t* n_tiles,
                              uint32_t* tile_tok, uint32_t* grp);

/* ---- PAF / .oec.zst ingest on the host (needs no device) ---------------------------------------
 * herro_paf_parse replaces `parse_paf` (overlaps.rs:117-202): one overlap per line, tab separated
 *   qname qlen qstart qend strand tname tlen tstart tend ... cg:Z:<CIGAR>   (CIGAR in the LAST column),
 * lines with unknown names, self overlaps and every (query, target) pair after its first occurrence are
 * dropped; the rest is grouped by target.  Quirks kept: the last byte of every line is dropped whether or
 * not it is a newline; numbers wrap in u32; malformed numbers / strand / CIGAR column are the reference's
 * panics (NULL is returned, `err` carries the message and the line).  `names` + `name_off[n_reads+1]` give
 * the read ids (index = read id of the read store; a repeated name maps to its last index, like the
 * reference's HashMap).  `core` (optional, one byte per read) replaces the `core` name set: targets with a 0
 * are skipped.  Targets come out in order of first appearance (the reference iterates a HashMap).
 * herro_oec_read replaces `read_batches` for one file (overlaps.rs:292-323): zstd stream holding
 * "<n_targets>\n", n_targets id lines (ignored, as in the reference), then PAF lines.
 * The result feeds herro_job_create directly: rids = herro_paf_target_ids, aln_off, alns.  The CIGAR
 * pointers stay valid until herro_paf_free. */
typedef struct herro_paf herro_paf;
/* The name -> read id index of a read set, built once (the reference's `name_to_id`, lib.rs:136-140) and shared by any number
 * of parses, also concurrent ones; herro_paf_parse / herro_oec_read build a temporary one per call (fine for one batch,
 * wasteful for a thousand batch files over millions of reads). */
typedef struct herro_name_index herro_name_index;
herro_name_index* herro_name_index_create(uint32_t n_reads, const char* names, const uint64_t* name_off);
void herro_name_index_free(herro_name_index* index);
herro_paf* herro_paf_parse_indexed(const char* text, uint64_t len, const herro_name_index* index, const uint8_t* core,
                                   int n_threads, char* err, uint64_t err_cap);
herro_paf* herro_oec_read_indexed(const char* path, const herro_name_index* index, const uint8_t* core, int n_threads,
                                  char* err, uint64_t err_cap);
/* herro_paf_parse_indexed without the copy: the result (its CIGAR pointers) refers to `text` itself — a buffer or file mapping
 * the caller keeps alive and unchanged until herro_paf_free. */
herro_paf* herro_paf_parse_view(const char* text, uint64_t len, const herro_name_index* index, const uint8_t* core,
                                int n_threads, char* err, uint64_t err_cap);
herro_paf* herro_paf_parse(const char* text, uint64_t len, uint32_t n_reads, const char* names,
                           const uint64_t* name_off, const uint8_t* core, int n_threads, char* err,
                           uint64_t err_cap);
herro_paf* herro_oec_read(const char* path, uint32_t n_reads, const char* names, const uint64_t* name_off,
                          const uint8_t* core, int n_threads, char* err, uint64_t err_cap);
uint32_t herro_paf_n_targets(const herro_paf* p);
const uint32_t* herro_paf_target_ids(const herro_paf* p);
const uint64_t* herro_paf_aln_off(const herro_paf* p);        /* [n_targets + 1] */
const herro_alignment* herro_paf_alignments(const herro_paf* p);
void herro_paf_free(herro_paf* p);

/* ---- reads and the `herro features` sink -----------------------------------------------------
 * herro_fastx_read = get_reads (haec_io.rs:37-75) over needletail's parse_fastx_file: FASTA or FASTQ by the first byte,
 * gzip by magic, multi-line records, '\r' dropped; records shorter than min_length dropped; header split at the first blank
 * or tab into id / description (NULL: none); qualities mandatory (a FASTA record fails with "Qualities should be
 * present."); keep_ids (NULL: keep all) = the union of the reference's core and neighbour sets, which it applies only
 * when both are given (haec_io.rs:63-69).  Returns NULL and a message in err on the inputs the reference panics on.
 * Sequence and quality bytes of read i are [off[i], off[i+1]) of herro_reads_seq / herro_reads_qual — the arguments
 * of herro_set_reads.  Pointers stay valid until herro_reads_free.  A plain regular file is read by byte ranges on
 * HERRO_FASTX_THREADS threads (default min(hardware threads, 16); ~9 GB/s on 8 cores, 1.4 GB/s on one), gzip and pipes as one
 * stream; the result is the same either way. */
typedef struct herro_reads herro_reads;
herro_reads* herro_fastx_read(const char* path, uint32_t min_length, const char* const* keep_ids, uint64_t n_keep, char* err,
                              uint64_t err_cap);
uint32_t herro_reads_count(const herro_reads* r);
const uint8_t* herro_reads_seq(const herro_reads* r);
const uint8_t* herro_reads_qual(const herro_reads* r);
const uint64_t* herro_reads_off(const herro_reads* r);            /* [n + 1] */
const char* const* herro_reads_ids(const herro_reads* r);         /* [n] */
const char* const* herro_reads_descs(const herro_reads* r);       /* [n], NULL entries: no description */
void herro_reads_free(herro_reads* r);

/* One window of `herro features` (output_features, features.rs:724-764): <dir>/<wid>.ids.txt (one id per line),
 * <wid>.features.npy = u8 [2, length, 31] (ASCII bases [length][31], then qualities), <wid>.supported.npy = records
 * {pos: <u2, ins: u1}.  NPY format 1.0, C order, header as numpy writes it (dict padded with spaces to a multiple of 64
 * bytes). */
int herro_write_window_features(const char* dir, uint32_t wid, const char* const* ids, uint32_t n_ids, const uint8_t* bases,
                                const uint8_t* quals, uint32_t length, const uint16_t* sup_pos, const uint8_t* sup_ins,
                                uint32_t n_sup);
/* `herro features` for a featurized job (FeatsGenOutput, features.rs:783-839): every window of every target into
 * <base_dir>/<read id of the target>/, read_names[rid] for all reads of the store.  Returns the number of windows
 * written or a negative error. */
int64_t herro_job_write_features(herro_job* job, const char* base_dir, const char* const* read_names);

/* ---- several contexts (GPUs) of ONE process fed from one queue (csrc/pool.cpp) --------------------------------------------------
 * Replaces the thread layout of lib.rs:154-200 (per device: `t` feature threads + one inference thread pulling from one MPMC
 * channel; one writer, lib.rs:267-291): herro_pool_create makes one context per entry of device_ids (several entries may name the
 * same device: its contexts share one read store) with a worker thread each; herro_pool_correct cuts the targets into groups of
 * `group_targets` reads and the workers PULL groups from one shared counter — whoever is free takes the next one — keeping two
 * jobs in flight per context (the next group is created while the GPU works on the current one).  The FASTA records of all
 * targets, in target order, stay with the pool until the next call: herro_pool_result returns the text and (via *rec_end) the
 * end offset of every target's records; the return value of herro_pool_correct is the text's length (< 0: HERRO_E_*, message in
 * herro_pool_last_error).  ids / descs: one C string per target (descs or its entries may be NULL), as for herro_job_fasta.
 * herro_pool_ctx exposes a context (e.g. for herro_set_precision, herro_model_describe); herro_pool_groups_taken says how many
 * groups context i took in the last call.  A client of this header only: every result is what the per-job calls give. */
typedef struct herro_pool herro_pool;
herro_pool* herro_pool_create(const int* device_ids, uint32_t n_ctx);
void herro_pool_destroy(herro_pool* pool);
const char* herro_pool_last_error(const herro_pool* pool);
uint32_t herro_pool_size(const herro_pool* pool);
herro_ctx* herro_pool_ctx(herro_pool* pool, uint32_t i);
int herro_pool_set_reads(herro_pool* pool, uint32_t n_reads, const uint8_t* seq, const uint8_t* qual, const uint64_t* off, const uint32_t* name_class);
int herro_pool_load_model(herro_pool* pool, const char* path);
int64_t herro_pool_correct(herro_pool* pool, uint32_t n_targets, const uint32_t* rids, const uint64_t* aln_off, const herro_alignment* alns,
                           uint32_t window_size, uint32_t batch_size, int batch_mode, uint32_t group_targets, const char* const* ids,
                           const char* const* descs);
const char* herro_pool_result(const herro_pool* pool, const uint64_t** rec_end);
uint32_t herro_pool_groups_taken(const herro_pool* pool, uint32_t i);
/* Alignments / targets the jobs of the last herro_pool_correct left out (herro_job_skipped summed over its groups). */
int herro_pool_skipped(const herro_pool* pool, uint64_t* n_alignments, uint64_t* n_targets);
/* Host-only test hook: a pool of n_ctx stand-in contexts that needs no device — context i "works" us_per_aln[i] microseconds per
 * alignment of a group, the FASTA of a target is ">id\n" + (rid % 7 + 1) bases; rid 0xfffffffe makes the group's herro_job_create fail
 * with HERRO_E_UNSUPPORTED, rid 0xfffffffd its herro_job_infer with HERRO_E_STATE; a target whose first alignment has qid == tid
 * counts as one skipped alignment.  Everything else (queue, two jobs in flight, merge, error paths) is herro_pool_correct's own code. */
herro_pool* herro_debug_pool_fake(uint32_t n_ctx, const uint32_t* us_per_aln);

#ifdef __cplusplus
}
#endif
#endif /* HERRO_AMD_H */
