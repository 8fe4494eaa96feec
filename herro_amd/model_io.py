"""Correction-model parameters: seeded random init of the ASSUMED architecture and export to the
flat weight file `herro_load_model` reads (stands in for `tch::CModule::load_on_device`,
reference inference.rs:185).

The real HERRO model is a TorchScript archive that is not part of the reference checkout
(`.MISSING_LARGE_BLOBS`, README.md:56-66); only its I/O contract is visible
(inference.rs:152-174).  Until a model file is reachable, weights are random-init of the
architecture documented in DESIGN.md (§Model) and model parity is "unpinned".

Raw parameters follow PyTorch naming/shape conventions so that a converter from a real
state_dict is a renaming exercise.  Export folds BatchNorm (eval) into the conv weights and the
embedding table through conv1 (float64 arithmetic, rounded once to f32).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, asdict

import numpy as np

MAGIC = 0x4F525248  # 'HRRO' little-endian
VOCAB = 12          # BASES_MAP tokens 0..10 + padding 11 (inference.rs:15,23-31)
PAD = 11


@dataclass(frozen=True)
class Hyper:
    rows: int = 31
    emb: int = 6
    kw: int = 3
    c1: int = 64
    c2: int = 128
    d_model: int = 256
    n_heads: int = 8
    d_ff: int = 1024
    n_layers: int = 4
    ln_eps: float = 1e-5
    bn_eps: float = 1e-5
    # variants of the family an archive may hold (round 6; the f16 / fused kernels serve the defaults, the layer-by-layer kernels every combination):
    act: int = 0          # encoder feed-forward activation: 0 ReLU, 1 GELU (erf), 2 GELU (tanh approximation)
    norm_first: int = 1   # 1 Pre-LN (x + f(LN(x))), 0 Post-LN (LN(x + f(x)))
    pe: int = 0           # position: 0 sinusoidal of the row index, 1 learned table [pe_rows, d_model], 2 none
    final_norm: int = 1   # LayerNorm behind the last encoder layer
    pe_rows: int = 0      # rows of the learned table
    bn: int = 1           # BatchNorm behind the convolutions (0: none)


def random_raw_params(hp: Hyper = Hyper(), seed: int = 0x48455252) -> dict[str, np.ndarray]:
    """PyTorch-default-like init (uniform +-1/sqrt(fan_in)), non-trivial BN/LN statistics."""
    g = np.random.default_rng(seed)
    f32 = np.float32

    def uni(shape, fan_in):
        b = 1.0 / np.sqrt(fan_in)
        return g.uniform(-b, b, shape).astype(f32)

    p: dict[str, np.ndarray] = {}
    emb = g.normal(0, 1, (VOCAB, hp.emb)).astype(f32)
    emb[PAD] = 0  # padding_idx row
    p["embedding.weight"] = emb
    cin = hp.emb + 1
    p["conv1.weight"] = uni((hp.c1, cin, hp.kw, 1), cin * hp.kw)
    p["conv1.bias"] = uni((hp.c1,), cin * hp.kw)
    p["conv2.weight"] = uni((hp.c2, hp.c1, hp.kw, 1), hp.c1 * hp.kw)
    p["conv2.bias"] = uni((hp.c2,), hp.c1 * hp.kw)
    for n, c in ((("bn1", hp.c1), ("bn2", hp.c2)) if hp.bn else ()):
        p[f"{n}.weight"] = g.uniform(0.5, 1.5, c).astype(f32)
        p[f"{n}.bias"] = g.normal(0, 0.1, c).astype(f32)
        p[f"{n}.running_mean"] = g.normal(0, 0.1, c).astype(f32)
        p[f"{n}.running_var"] = g.uniform(0.5, 1.5, c).astype(f32)
    p["fc.weight"] = uni((hp.d_model, hp.rows * hp.c2), hp.rows * hp.c2)
    p["fc.bias"] = uni((hp.d_model,), hp.rows * hp.c2)
    D = hp.d_model
    for l in range(hp.n_layers):
        q = f"encoder.layers.{l}."
        p[q + "self_attn.in_proj_weight"] = uni((3 * D, D), D)
        p[q + "self_attn.in_proj_bias"] = uni((3 * D,), D)
        p[q + "self_attn.out_proj.weight"] = uni((D, D), D)
        p[q + "self_attn.out_proj.bias"] = uni((D,), D)
        p[q + "linear1.weight"] = uni((hp.d_ff, D), D)
        p[q + "linear1.bias"] = uni((hp.d_ff,), D)
        p[q + "linear2.weight"] = uni((D, hp.d_ff), hp.d_ff)
        p[q + "linear2.bias"] = uni((D,), hp.d_ff)
        for n in ("norm1", "norm2"):
            p[q + n + ".weight"] = g.uniform(0.8, 1.2, D).astype(f32)
            p[q + n + ".bias"] = g.normal(0, 0.05, D).astype(f32)
    if hp.final_norm:
        p["encoder.norm.weight"] = g.uniform(0.8, 1.2, D).astype(f32)
        p["encoder.norm.bias"] = g.normal(0, 0.05, D).astype(f32)
    if hp.pe == 1:
        p["pos_table"] = g.normal(0, 0.5, (hp.pe_rows, D)).astype(f32)
    p["info_head.weight"] = uni((1, D), D)
    p["info_head.bias"] = uni((1,), D)
    p["base_head.weight"] = uni((5, D), D)
    p["base_head.bias"] = uni((5,), D)
    return p


def pe_div_term(d_model: int) -> np.ndarray:
    """exp(-(2i) ln(10000)/d) in f32 — shared verbatim by the twin and the kernels."""
    i = np.arange(0, d_model, 2, dtype=np.float64)
    return np.exp(-i * np.log(10000.0) / d_model).astype(np.float32)


def fold(raw: dict[str, np.ndarray], hp: Hyper) -> dict[str, np.ndarray]:
    """Raw (PyTorch-shaped) parameters -> the tensors the HIP kernels consume."""
    f64 = np.float64
    out: dict[str, np.ndarray] = {}

    def bn_scale_shift(n):
        if f"{n}.weight" not in raw:      # no BatchNorm behind this convolution
            c = raw[("conv1" if n == "bn1" else "conv2") + ".weight"].shape[0]
            return np.ones(c, f64), np.zeros(c, f64)
        s = raw[f"{n}.weight"].astype(f64) / np.sqrt(raw[f"{n}.running_var"].astype(f64) + hp.bn_eps)
        return s, raw[f"{n}.bias"].astype(f64) - raw[f"{n}.running_mean"].astype(f64) * s

    s1, sh1 = bn_scale_shift("bn1")
    w1 = raw["conv1.weight"].astype(f64)[:, :, :, 0] * s1[:, None, None]  # [c1, 7, kw]
    b1 = raw["conv1.bias"].astype(f64) * s1 + sh1
    emb = raw["embedding.weight"].astype(f64)  # [12, emb]
    # t1[tap][tok][c] = sum_e w1[c][e][tap] * emb[tok][e]
    t1 = np.einsum("cet,ve->tvc", w1[:, : hp.emb, :], emb)
    out["t1"] = t1
    out["wq1"] = w1[:, hp.emb, :].T.copy()  # [kw][c1]
    out["b1"] = b1
    s2, sh2 = bn_scale_shift("bn2")
    w2 = raw["conv2.weight"].astype(f64)[:, :, :, 0] * s2[:, None, None]  # [c2, c1, kw]
    out["conv2.wt"] = w2.transpose(0, 2, 1).reshape(hp.c2, hp.kw * hp.c1)  # [N][k = tap*c1 + c]
    out["conv2.b"] = raw["conv2.bias"].astype(f64) * s2 + sh2
    out["fc.wt"] = raw["fc.weight"]      # [D][rows*c2], input index = row*c2 + c
    out["fc.b"] = raw["fc.bias"]
    out["pe_div"] = pe_div_term(hp.d_model)
    for l in range(hp.n_layers):
        q, o = f"encoder.layers.{l}.", f"L{l}."
        out[o + "ln1.g"], out[o + "ln1.b"] = raw[q + "norm1.weight"], raw[q + "norm1.bias"]
        out[o + "ln2.g"], out[o + "ln2.b"] = raw[q + "norm2.weight"], raw[q + "norm2.bias"]
        out[o + "qkv.wt"], out[o + "qkv.b"] = raw[q + "self_attn.in_proj_weight"], raw[q + "self_attn.in_proj_bias"]
        out[o + "proj.wt"], out[o + "proj.b"] = raw[q + "self_attn.out_proj.weight"], raw[q + "self_attn.out_proj.bias"]
        out[o + "ff1.wt"], out[o + "ff1.b"] = raw[q + "linear1.weight"], raw[q + "linear1.bias"]
        out[o + "ff2.wt"], out[o + "ff2.b"] = raw[q + "linear2.weight"], raw[q + "linear2.bias"]
    if hp.final_norm:
        out["lnf.g"], out["lnf.b"] = raw["encoder.norm.weight"], raw["encoder.norm.bias"]
    if hp.pe == 1:
        out["pe_table"] = raw["pos_table"]
    if (hp.act, hp.norm_first, hp.pe, hp.final_norm) != (0, 1, 0, 1):   # the defaults need no record: files of earlier rounds stay what they were
        out["cfg"] = np.array([hp.act, hp.norm_first, hp.pe, hp.final_norm], np.float32)
    heads_w = np.zeros((16, hp.d_model), np.float32)
    heads_b = np.zeros(16, np.float32)
    heads_w[0], heads_b[0] = raw["info_head.weight"][0], raw["info_head.bias"][0]
    heads_w[1:6], heads_b[1:6] = raw["base_head.weight"], raw["base_head.bias"]
    out["heads.wt"], out["heads.b"] = heads_w, heads_b
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in out.items()}


def write_flat(path: str, folded: dict[str, np.ndarray], hp: Hyper) -> None:
    with open(path, "wb") as f:
        f.write(struct.pack("<11If", MAGIC, 1, hp.rows, hp.kw, hp.c1, hp.c2, hp.d_model, hp.n_heads, hp.d_ff,
                            hp.n_layers, len(folded), hp.ln_eps))
        for name, t in folded.items():
            nb = name.encode()
            assert len(nb) < 32 and t.ndim <= 4
            dims = list(t.shape) + [0] * (4 - t.ndim)
            f.write(nb.ljust(32, b"\0"))
            f.write(struct.pack("<5I", t.ndim, *dims))
            f.write(np.ascontiguousarray(t, np.float32).tobytes())


def export(raw: dict[str, np.ndarray], hp: Hyper, path: str) -> None:
    write_flat(path, fold(raw, hp), hp)


def save_raw(path: str, raw: dict[str, np.ndarray], hp: Hyper) -> None:
    np.savez(path, __hyper__=np.array(list(asdict(hp).values()), dtype=np.float64), **raw)


def default_model_file(cache_dir: str, hp: Hyper = Hyper(), seed: int = 0x48455252) -> tuple[str, dict[str, np.ndarray]]:
    """Writes (once) the random-init flat file used by tests / smoke / bench; returns (path, raw)."""
    import os
    os.makedirs(cache_dir, exist_ok=True)
    raw = random_raw_params(hp, seed)
    var = "" if (hp.act, hp.norm_first, hp.pe, hp.final_norm, hp.bn, hp.n_heads, hp.d_ff) == (0, 1, 0, 1, 1, 8, 1024) else f"_v{hp.act}{hp.norm_first}{hp.pe}{hp.final_norm}{hp.bn}_{hp.n_heads}_{hp.d_ff}"
    path = os.path.join(cache_dir, f"herro_random_{seed:x}_{hp.kw}_{hp.c1}_{hp.c2}_{hp.d_model}_{hp.n_layers}{var}.hrro")
    if not os.path.exists(path):
        export(raw, hp, path + ".tmp")
        os.replace(path + ".tmp", path)
    return path, raw
