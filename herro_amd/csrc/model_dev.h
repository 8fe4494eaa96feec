// model_dev.h — device-side view of the correction model and of one inference batch.
//
// The reference executes an opaque TorchScript file (inference.rs:162-163, 185) that is NOT part
// of /root/reference; only its I/O contract is (inference.rs:152-174).  The architecture below is
// the ASSUMED one (oracle/model_ref.py is its PyTorch twin): token embedding +
// quality channel -> two read-wise 1-D conv blocks along the window axis -> per-position linear
// over the 31 rows -> informative positions only -> Pre-LN Transformer encoder -> two heads.
// Model parity with the real weights is therefore "unpinned".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "job_dev.h"

namespace herro {

struct ModelHyper {  // header of the flat weight file (tools/export_weights.py)
  uint32_t magic;    // 'HRRO'
  uint32_t version;
  uint32_t rows;     // 31
  uint32_t kw;       // conv kernel width along the window axis (odd)
  uint32_t c1, c2;   // conv channels
  uint32_t d_model, n_heads, d_ff, n_layers;
  uint32_t n_tensors;
  float ln_eps;
};

// A [K,N] row-major weight kept in three forms: f32, and its bf16 hi/lo split (w ~= hi + lo).
struct Weight {
  const float* f32 = nullptr;
  const uint16_t* hi = nullptr;
  const uint16_t* lo = nullptr;
  // the same planes pre-shuffled into MFMA fragment order (N, K multiples of 32): element
  //   ((((n / 32) * 2 + jt) * (K / 32) + ks) * 64 + lane) * 8 + e  =  W[32 (n/32) + 8 (fr >> 2) + 4 jt + (fr & 3)][32 ks + 8 fg + e],
  // lane = 16 fg + fr — a wave's fragment load is one contiguous KiB instead of 16 scattered 64-byte pieces
  const uint16_t* phi = nullptr;
  const uint16_t* plo = nullptr;
  // f16 forms for precision modes 4 / 5 (model_h.hip): h16 = round-to-nearest f16 of the f32 weight, l16 = f16 of the
  // remainder (only the heads use it), ph16 = h16 in the fragment order of phi
  const uint16_t* h16 = nullptr;
  const uint16_t* l16 = nullptr;
  const uint16_t* ph16 = nullptr;
  // precision 6 (N % 32 == 0, K % 128 == 0; the encoder's proj / ff1 / ff2): OCP e4m3 of W * 2^(127 - s8) in the fragment order of
  // v_mfma_scale_f32_16x16x128_f8f6f4 — byte ((((n / 32) * 2 + jt) * (K / 128) + s) * 2 + half) * 1024 + lane * 16 + j  =
  // W[32 (n/32) + 8 (fr >> 2) + 4 jt + (fr & 3)][128 s + 32 fg + 16 half + j]; s8 = the E8M0 scale byte the instruction takes
  const uint8_t* p8 = nullptr;
  uint32_t s8 = 127;
  const float* bias = nullptr;  // [N] or null
  uint32_t K = 0, N = 0;
};

struct LayerW {
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  Weight qkv, proj, ff1, ff2;
};

struct ModelDev {
  ModelHyper h;
  const float* t1;      // [kw][12][c1] embedding folded through conv1 (+BN), tap-major
  const float* wq1;     // [kw][c1]     conv1 weights of the quality channel
  const float* b1;      // [c1]
  Weight conv1g;        // conv1 as a GEMM for k_conv_m (model_h.hip): [kw*32, c1] f16 hi / lo slots of table, quality weight and bias; ph16 only
  Weight conv2;         // [kw*c1, c2]  (BN folded), k = tap*c1 + c
  Weight fc;            // [rows*c2, d_model], k = row*c2 + c
  const float* pe_div;  // [d_model/2]
  const float* pe_tab;  // [pe_rows][d_model] the positional encoding of rows 0 .. pe_rows - 1, tabulated at load by the arithmetic k_layers_p used to run per token
  uint32_t pe_rows;     // (16 sincosf per lane and tile were 17 % of the stack's vector instructions); rows beyond the table are still computed in the kernel; null / 0: no table
  LayerW layer[16];
  const float *lnf_g, *lnf_b;   // null: no LayerNorm behind the last layer (final_norm == 0)
  Weight heads;         // [d_model, 16]: col 0 info, cols 1..5 bases
  // Variants of the family (round 6; "cfg" / "pe_table" of the weight file, herro_amd/model_io.py Hyper).  The fused / f16 kernels implement the defaults;
  // every other combination runs on the layer-by-layer kernels (model.hip launch_model / launch_model_s).
  uint32_t act;         // encoder feed-forward activation: 0 ReLU, 1 GELU (erf), 2 GELU (tanh approximation)
  uint32_t norm_first;  // 1 Pre-LN, 0 Post-LN
  uint32_t pe_kind;     // 0 sinusoidal of the row index, 1 learned table, 2 none
  uint32_t final_norm;  // 1: lnf_g / lnf_b behind the last layer
  const float* pe_learned;   // [pe_learned_rows][d_model]
  uint32_t pe_learned_rows;
};
inline bool model_default_variant(const ModelDev& M) { return M.act == 0 && M.norm_first == 1 && M.pe_kind == 0 && M.final_norm == 1; }

// One launch group = one or more inference batches back to back: windows are ragged planes
// [31][len] (stride lub), never padded in memory; the batch a window belongs to only matters through
// lmax[b] (the padding the reference's collate would add), so several batches share one launch.
struct BatchDev {
  uint32_t n_win;             // B
  uint32_t n_tok;             // N = sum of informative positions
  uint32_t max_win_tok;       // most informative positions in one window (0: unknown) — sizes the query-block dimension of k_attention_s
  const uint64_t* plane_off;  // [B] byte offset of the window's token/quality planes
  const uint32_t* plane_ld;   // [B] plane stride (lub)
  const uint32_t* len;        // [B] L' of each window
  const uint32_t* lmax;       // [B] max L' over the window's *batch* (collate pads to it, inference.rs:75-97)
  const uint32_t* tok_off;    // [B+1] first token of each window
  const uint64_t* sup_off;    // [B] element offset of the window's informative-row list
  const uint64_t* out_off;    // [B] element offset of the window's logits in the job buffers
  const uint64_t* rf_base;    // [B] first record of the window's receptive fields in rf_q (k_rows places them by an atomic, k_rfq at out_off); null: out_off
  const uint8_t* planes_b;    // token planes (used when rf_q is null: the stand-alone entry, receptive fields above 8 rows)
  const uint8_t* planes_q;    // raw qualities (complete planes; used when rf_q is null)
  const uint8_t* rf_q;        // the receptive fields only, compact (k_rfq): one 16-byte record per (token, read row) at [(out_off[b] + k) * 31 + row]:
                              // bytes 0..7 the tokens of rows tok_row - 2 (kw / 2) + i, bytes 8..15 their qualities; null: read the planes
  const uint32_t* sup_row;    // informative rows
  uint32_t n_tiles;           // token tiles of whole windows (<= 64 tokens each) for the fused stack; 0: not tileable
  const uint32_t* tile_tok0;  // [n_tiles+1] first token of each tile
  uint32_t n_tiles_q;         // tiles of <= 32 tokens behind them in the token stream (k_layers_p<., 2>: the short last round of a launch); 0: none
  const uint32_t* tile_tok0_q;  // [n_tiles_q+1]
  // windows of 65 .. 64 * FUSED_MAX_SIB informative rows (f16 stack): each one alone on ceil(rows / 64) consecutive 64-token
  // tiles at the HEAD of the token stream and of the 64-token grid — sibling tiles, which exchange their K / V fragments layer by layer (k_layers_p<., 4, true>)
  uint32_t n_tiles_b;
  const uint32_t* tile_tok0_b;  // [n_tiles_b+1]
  const uint32_t* tile_grp;     // [n_tiles_b] first tile of the window's group | tiles in the group << 20 | tokens of the window in the group's last tile << 24 (small windows may fill the rest of it)
  float* out_info;            // job-level [sum nsup]
  float* out_base;            // job-level [sum nsup][5]
};

// Everything the conv kernel needs to fetch the receptive field of one token, in one 32-byte record.
struct __attribute__((aligned(16))) TokMeta {
  uint64_t plane_off;  // byte offset of the window's token / quality planes
  uint32_t plane_ld;   // plane stride
  uint32_t tok_row;    // row of the token inside the window
  uint32_t len, lmax;  // L' of the window, max L' of its batch (both < 65536)
  uint32_t rf_idx;     // job-level index of the token (slot of its receptive-field qualities in BatchDev::rf_q)
  uint32_t pad1;
};

// The same for k_conv_m (model_h.hip), with the token's five receptive-field rows tok_row - 2 .. + 2 resolved once per token instead
// of once per (read row, lane): byte masks of the rows inside the window's real rows, default bytes of the others — pad token /
// quality 126 inside the batch padding [len, lmax) (inference.rs:86-97), token 12 (zero table row) / no quality (0xff) outside
// [0, lmax).  Rows 0..3 in the 32-bit words, row 4 in the packed bytes.
struct __attribute__((aligned(16))) TokCv {
  uint64_t plane_off;
  uint32_t ld_d1;      // plane stride | default token of row 4 << 16 | default quality of row 4 << 24
  uint32_t row_ok;     // tok_row | validity of conv1 positions tok_row - 1 .. + 1 (inside [0, lmax)) << 16 | mask of row 4 << 24
  uint32_t rf_idx;
  uint32_t mk0, dt0, dq0;
};

struct ModelScratch {  // sized for n_tok tokens
  uint32_t* tok_win;  // [N] window (batch-local) of each token
  uint32_t* tok_row;  // [N] row of each token
  uint64_t* tok_out;  // [N] element offset of each token's logits in the job buffers (k_build_tokens_h; the stack's last phase reads one word instead of three dependent ones)
  TokMeta* tok_meta;  // [N]
  TokCv* tok_cv;      // [N]
  float* y1;          // [N][31][kw][c1]
  float* y2;          // [N][31*c2]
  float* x;           // [N][d_model] residual stream
  float* hbuf;        // [N][d_model] normalised
  float* qkv;         // [N][3*d_model]
  float* att;         // [N][d_model]
  float* ff;          // [N][d_ff]
  float* logits;      // [N][16]
  // bf16x3 pipeline: activations that feed a GEMM are kept pre-split as bf16 hi/lo planes
  // (a ~= hi + lo, 16 mantissa bits) so the GEMM never converts in its inner loop
  uint16_t *y1_hi, *y1_lo;    // [N][31][kw][c1]
  uint16_t *y2_hi, *y2_lo;    // [N][31*c2]
  uint16_t *h_hi, *h_lo;      // [N][d_model]
  uint16_t *att_hi, *att_lo;  // [N][d_model]
  uint16_t *ff_hi, *ff_lo;    // [N][d_ff]
  // sibling tiles of the f16 stack (windows above 64 informative rows): K / V fragments [tile][layer parity][wave = head][8 fragments][64 lanes] x 16 B,
  // flags [tile] = layers published so far (zeroed before every launch), one sticky error word (a sibling never showed up)
  uint16_t* sib_kv;
  uint32_t* sib_flag;
  uint32_t* sib_err;
};

void launch_model(const ModelDev& M, const BatchDev& B, const ModelScratch& S, int precision,
                  hipStream_t st, KernelTimer* tm);
// f16-operand kernels (model_h.hip): terms = 2 -> precision 4, terms = 1 -> precision 5, terms = 3 -> precision 6 (f16 + an e4m3 remainder term), 21 -> precision 7 (proj on two
// terms, FF1 / FF2 on one), 12 -> precision 8 (proj on one, FF on two).  B must be tiled (n_tiles > 0).
bool model_h_supported(const ModelDev& M);        // the f16 encoder stack serves the model (d_model 256, 8 heads of 32, d_ff % 256 == 0, the default variant): precisions 4 .. 8 exist
bool model_h_conv_supported(const ModelDev& M);   // ... and the f16 conv / FC kernels too (kw 3, 64 -> 128 channels); otherwise the bf16x3 front end of model.hip feeds the f16 stack
void launch_front_generic(const ModelDev& M, const BatchDev& B, const ModelScratch& S, hipStream_t st, KernelTimer* tm);   // model.hip
bool model_h_f8_supported(const ModelDev& M);   // ... and every layer's proj / ff1 / ff2 has its e4m3 copy
int model_h_half_tiles(const ModelDev& M);   // qmode of plan_tiles: 0 no 32-token tiles, 1 for a short last round (default), 2 for every small window (HERRO_LAYERS_Q)
void launch_model_h(const ModelDev& M, const BatchDev& B, const ModelScratch& S, int terms, hipStream_t st, KernelTimer* tm);
void launch_pe_table(const float* pe_div, float* tab, uint32_t rows, uint32_t d_model, hipStream_t st);   // tab[row][2 i] = sin(row * pe_div[i]), [2 i + 1] = cos
#ifdef HERRO_PROF_BUILD
void model_h_prof_dump();   // phase cycles of k_layers_p (model_h.hip LP_MARK), printed by herro_destroy when HERRO_PROF=1
#endif
// [B,L,31] -> [B][31][L] planes (stand-alone entry only)
void launch_transpose_blr(const uint8_t* src, uint8_t* dst, uint32_t B, uint32_t L, hipStream_t st);

}  // namespace herro
