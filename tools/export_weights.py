#!/usr/bin/env python
"""TorchScript archive -> flat weight file for herro_load_model.

The reference loads its correction model with `tch::CModule::load_on_device(model_path, device)`
(reference inference.rs:185-186) and calls `forward(bases, quals, lens, indices)` (inference.rs:155-163); the
published file is `model_R10_v0.1.pt` (reference README.md:56-66), which is not part of the checkout.  This tool
is the bridge for the day such a file is at hand:

    python tools/export_weights.py model.pt model.hrro [--dump report.txt] [--no-verify]

1. `torch.jit.load` the archive; print / dump its forward code, the operator histogram of the inlined graph, the
   sub-module tree and every parameter / buffer with shape and dtype.
2. Recover the hyper-parameters from the SHAPES (embedding width, conv kernel / channels, d_model, heads, d_ff,
   layers), map the state dict to the names `herro_amd.model_io.fold` consumes, and check the graph against the
   architecture the HIP kernels implement (embedding + quality -> two Conv(k,1)+BatchNorm+ReLU blocks -> Linear over
   the 31 rows -> gather -> sinusoidal position -> Pre-LN Transformer encoder (ReLU) + final LayerNorm -> two
   heads) and its VARIANTS (round 6): BatchNorm or none, ReLU / GELU in the encoder, Pre- / Post-LN, a final LayerNorm or none, a sinusoidal / learned /
   absent position term, heads of 32 or 64, any odd conv width, up to 16 layers.  Anything else — an operator outside that family, a parameter no rule
   claims — stops the conversion with the list of what was not understood.  Nothing is guessed: what the shapes cannot tell is read from the module
   attributes or the graph, and the conversion check (step 4) runs the archive against the folded tensors in the variant that was read.
3. Fold (BatchNorm into the convs, the embedding through conv1) and write the flat file; print the FLOP-per-window
   formula evaluated on the recovered shapes ("the builder must print it with the formula").
4. --verify: run the archive on a small random batch on the CPU and compare with a numpy evaluation of the FOLDED
   tensors in the kernels' dataflow (receptive-field evaluation, collate padding) — a conversion check, <= 2e-5.
"""
from __future__ import annotations

import argparse
import os
import re
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from herro_amd import model_io  # noqa: E402

# aten / prim operators the assumed architecture may lower to (scripted or traced, eager or fused encoder path)
ALLOWED_OPS = {
    # data movement / shape
    "aten::to", "aten::_to_copy", "aten::contiguous", "aten::permute", "aten::transpose", "aten::reshape", "aten::view", "aten::flatten",
    "aten::unflatten", "aten::unsqueeze", "aten::squeeze", "aten::cat", "aten::stack", "aten::slice", "aten::select", "aten::index",
    "aten::index_select", "aten::gather", "aten::index_put_", "aten::index_put", "aten::copy_", "aten::clone", "aten::expand", "aten::expand_as",
    "aten::repeat", "aten::chunk", "aten::split", "aten::split_with_sizes", "aten::unbind", "aten::size", "aten::len", "aten::dim",
    "aten::zeros", "aten::ones", "aten::zeros_like", "aten::ones_like", "aten::full", "aten::full_like", "aten::empty", "aten::arange", "aten::tensor",
    "aten::masked_fill", "aten::masked_fill_", "aten::masked_select", "aten::where", "aten::bitwise_not", "aten::logical_not", "aten::__not__",
    "aten::tolist", "aten::Int", "aten::Float", "aten::Bool", "aten::item", "aten::ScalarImplicit", "aten::IntImplicit", "aten::FloatImplicit",
    "aten::detach", "aten::type_as", "aten::pad", "aten::constant_pad_nd", "aten::max", "aten::min", "aten::sum", "aten::any", "aten::all",
    "aten::eq", "aten::ne", "aten::lt", "aten::le", "aten::gt", "aten::ge", "aten::__is__", "aten::__isnot__", "aten::__getitem__", "aten::__contains__",
    "aten::append", "aten::list", "aten::format", "aten::warn", "aten::nested_to_padded_tensor", "aten::_nested_tensor_from_mask",
    "aten::_nested_tensor_from_mask_left_aligned", "aten::is_floating_point", "aten::requires_grad", "aten::is_grad_enabled", "aten::device", "aten::dtype",
    "aten::__and__", "aten::__or__", "aten::logical_and", "aten::logical_or", "aten::ceil", "aten::floor", "aten::floordiv", "aten::remainder",
    "aten::embedding", "aten::is_autocast_enabled", "aten::is_autocast_cpu_enabled", "aten::to_padded_tensor", "aten::triu", "aten::tril",
    "aten::is_nested", "aten::numel", "aten::is_cuda", "aten::is_cpu",
    # arithmetic of the model
    "aten::add", "aten::add_", "aten::sub", "aten::mul", "aten::mul_", "aten::div", "aten::neg", "aten::sqrt", "aten::rsqrt", "aten::pow",
    "aten::exp", "aten::log", "aten::sin", "aten::cos", "aten::relu", "aten::relu_", "aten::gelu", "aten::gelu_", "aten::erf", "aten::dropout", "aten::dropout_", "aten::feature_dropout",
    "aten::conv2d", "aten::conv1d", "aten::_convolution", "aten::convolution", "aten::batch_norm", "aten::linear", "aten::matmul", "aten::addmm",
    "aten::bmm", "aten::baddbmm", "aten::mm", "aten::t", "aten::layer_norm", "aten::native_layer_norm", "aten::softmax", "aten::_softmax",
    "aten::scaled_dot_product_attention", "aten::_native_multi_head_attention", "aten::_transformer_encoder_layer_fwd",
    "aten::multi_head_attention_forward",
}
FORBIDDEN_HINT = {
    "aten::silu": "SiLU activation", "aten::tanh": "tanh activation",
    "aten::sigmoid": "sigmoid inside the model", "aten::lstm": "recurrent layer", "aten::gru": "recurrent layer",
    "aten::max_pool2d": "pooling", "aten::avg_pool2d": "pooling", "aten::adaptive_avg_pool2d": "pooling", "aten::group_norm": "GroupNorm",
    "aten::instance_norm": "InstanceNorm", "aten::leaky_relu": "LeakyReLU", "aten::elu": "ELU",
}


class Unsupported(RuntimeError):
    pass


def load_archive(path: str):
    import torch
    m = torch.jit.load(path, map_location="cpu")
    m.eval()
    return m


def graph_ops(m) -> Counter:
    """operator histogram of the inlined forward graph (sub-blocks included)"""
    ops = Counter()

    def walk(block):
        for n in block.nodes():
            ops[n.kind()] += 1
            for b in n.blocks():
                walk(b)
    walk(m.inlined_graph)
    return ops


def describe(m) -> str:
    out = []
    try:
        out.append("==== forward code ====\n" + m.code)
    except Exception as e:  # pragma: no cover
        out.append(f"(no code: {e})")
    out.append("==== operator histogram (inlined graph) ====")
    for k, v in sorted(graph_ops(m).items()):
        out.append(f"  {k:50s} {v}")
    out.append("==== sub-modules ====")
    for name, sub in m.named_modules():
        out.append(f"  {name or '<root>':50s} {getattr(sub, 'original_name', type(sub).__name__)}")
    out.append("==== parameters ====")
    for name, p in m.named_parameters():
        out.append(f"  {name:60s} {tuple(p.shape)} {p.dtype}")
    out.append("==== buffers ====")
    for name, p in m.named_buffers():
        out.append(f"  {name:60s} {tuple(p.shape)} {p.dtype}")
    return "\n".join(out)


def check_graph(m) -> list[str]:
    ops = graph_ops(m)
    problems = []
    for k in sorted(ops):
        if k.startswith("prim::"):
            continue
        if k in FORBIDDEN_HINT:
            problems.append(f"{k} x{ops[k]}: {FORBIDDEN_HINT[k]}")
        elif k not in ALLOWED_OPS:
            problems.append(f"{k} x{ops[k]}: operator outside the assumed architecture")
    if not any(k in ops for k in ("aten::conv2d", "aten::conv1d", "aten::_convolution", "aten::convolution")):
        problems.append("no convolution in the graph")
    if not any(k in ops for k in ("aten::layer_norm", "aten::native_layer_norm", "aten::_transformer_encoder_layer_fwd")):
        problems.append("no LayerNorm / encoder layer in the graph")
    return problems


def gelu_kind(m):
    """0: no GELU in the graph; 1: erf form; 2: tanh approximation (the `approximate` argument of aten::gelu)"""
    kind = 0

    def walk(block):
        nonlocal kind
        for n in block.nodes():
            if n.kind() in ("aten::gelu", "aten::gelu_"):
                ins = list(n.inputs())
                ap = ins[1].toIValue() if len(ins) > 1 else "none"
                kind = max(kind, 2 if ap == "tanh" else 1)
            for b in n.blocks():
                walk(b)
    walk(m.inlined_graph)
    return kind


def recover(m):
    """state dict of the archive -> (Hyper, raw dict under model_io's canonical names).  Raises Unsupported.
    Variants of the family are RECOGNISED, not refused (round 6): BatchNorm behind the convolutions or none, ReLU / GELU in the encoder, Pre- / Post-LN,
    a final LayerNorm or none, a sinusoidal / learned / absent position term, heads of 32 or 64 — the conversion check then runs the archive against the
    folded tensors in exactly that variant, so a wrong reading cannot produce a file."""
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    ops = graph_ops(m)
    used: set[str] = set()
    notes: list[str] = []

    def find(patterns, pred=None, what=""):
        """the single tensor whose name matches one of the regexes (and pred(shape)); Unsupported if 0 or > 1"""
        hits = [k for k in sd if k not in used and any(re.search(p, k) for p in patterns) and (pred is None or pred(sd[k].shape))]
        if len(hits) != 1:
            raise Unsupported(f"{what}: expected exactly one tensor matching {patterns}, found {hits or 'none'}")
        used.add(hits[0])
        return hits[0], sd[hits[0]]

    # ---- embedding: [12, E]
    _, emb = find([r"emb\w*\.weight$"], lambda s: len(s) == 2 and s[0] == model_io.VOCAB, "token embedding (12 x E)")
    E = emb.shape[1]
    # ---- convolutions: the two weights of rank 3 / 4 in definition order; kernel along the window axis
    conv_w = [k for k in sd if re.search(r"\.weight$", k) and sd[k].ndim in (3, 4) and k not in used]
    if len(conv_w) != 2:
        raise Unsupported(f"expected two convolution weights, found {conv_w}")

    def conv_kernel(w, name):
        if w.ndim == 3:            # Conv1d over the window axis, rows folded into the batch
            return w[:, :, :, None]
        if w.shape[3] == 1:
            return w               # Conv2d, kernel (kw, 1), input [B, C, L, R] — the assumed layout
        if w.shape[2] == 1:        # Conv2d, kernel (1, kw), input [B, C, R, L]: the same map with the axes swapped
            notes.append(f"{name}: kernel (1, {w.shape[3]}) -> treated as ({w.shape[3]}, 1) over [B, C, L, R]")
            return np.ascontiguousarray(w.transpose(0, 1, 3, 2))
        raise Unsupported(f"{name}: kernel {w.shape[2:]} mixes read rows (the kernels convolve along the window axis only)")
    c1w, c2w = (conv_kernel(sd[k], k) for k in conv_w)
    used.update(conv_w)
    if c1w.shape[1] != E + 1:
        raise Unsupported(f"conv1 takes {c1w.shape[1]} input channels, expected embedding {E} + 1 quality channel")
    if c2w.shape[1] != c1w.shape[0] or c2w.shape[2] != c1w.shape[2]:
        raise Unsupported(f"conv stack shapes {c1w.shape} -> {c2w.shape} are not two blocks with one kernel width")
    kw, C1, C2 = c1w.shape[2], c1w.shape[0], c2w.shape[0]
    raw = {"embedding.weight": emb, "conv1.weight": c1w, "conv2.weight": c2w}
    for n, k in zip(("conv1", "conv2"), conv_w):
        bk = k[: -len("weight")] + "bias"
        if bk not in sd:
            raise Unsupported(f"{k} has no bias tensor {bk}")
        raw[f"{n}.bias"] = sd[bk]
        used.add(bk)
    # ---- batch norms: running_mean of size C1 / C2
    bns = [k[: -len("running_mean")] for k in sd if k.endswith("running_mean")]
    if bns and (len(bns) != 2 or sd[bns[0] + "running_mean"].shape != (C1,) or sd[bns[1] + "running_mean"].shape != (C2,)):
        raise Unsupported(f"expected BatchNorm({C1}) and BatchNorm({C2}) after the convolutions (or none), found {bns}")
    if not bns:
        notes.append("no BatchNorm behind the convolutions")
    for n, pfx in zip(("bn1", "bn2"), bns):
        for f in ("weight", "bias", "running_mean", "running_var"):
            raw[f"{n}.{f}"] = sd[pfx + f]
            used.add(pfx + f)
        used.add(pfx + "num_batches_tracked")
    # ---- per-position linear over the rows: [D, rows * C2]
    name, fcw = find([r"\.weight$", r"^weight$"], lambda s: len(s) == 2 and s[1] % C2 == 0 and s[1] // C2 == 31, "linear over the 31 rows (D x 31*C2)")
    D = fcw.shape[0]
    raw["fc.weight"], raw["fc.bias"] = fcw, sd[name[: -len("weight")] + "bias"]
    used.add(name[: -len("weight")] + "bias")
    # ---- encoder layers (torch.nn.TransformerEncoder naming)
    layer_ids = sorted({int(mm.group(1)) for k in sd for mm in [re.search(r"layers\.(\d+)\.self_attn\.in_proj_weight$", k)] if mm})
    if not layer_ids or layer_ids != list(range(len(layer_ids))):
        raise Unsupported("no torch.nn.TransformerEncoder layers (…layers.N.self_attn.in_proj_weight) in the state dict")
    pfx = next(k for k in sd if re.search(r"layers\.0\.self_attn\.in_proj_weight$", k))
    pfx = pfx[: pfx.index("layers.0.")]
    FF = None
    for l in layer_ids:
        q = f"{pfx}layers.{l}."
        for src, dst in (("self_attn.in_proj_weight", "self_attn.in_proj_weight"), ("self_attn.in_proj_bias", "self_attn.in_proj_bias"),
                         ("self_attn.out_proj.weight", "self_attn.out_proj.weight"), ("self_attn.out_proj.bias", "self_attn.out_proj.bias"),
                         ("linear1.weight", "linear1.weight"), ("linear1.bias", "linear1.bias"), ("linear2.weight", "linear2.weight"),
                         ("linear2.bias", "linear2.bias"), ("norm1.weight", "norm1.weight"), ("norm1.bias", "norm1.bias"),
                         ("norm2.weight", "norm2.weight"), ("norm2.bias", "norm2.bias")):
            if q + src not in sd:
                raise Unsupported(f"encoder layer {l}: missing {q + src}")
            raw[f"encoder.layers.{l}.{dst}"] = sd[q + src]
            used.add(q + src)
        if sd[q + "self_attn.in_proj_weight"].shape != (3 * D, D):
            raise Unsupported(f"encoder layer {l}: in_proj_weight {sd[q + 'self_attn.in_proj_weight'].shape} != (3*{D}, {D})")
        ff = sd[q + "linear1.weight"].shape[0]
        if FF not in (None, ff):
            raise Unsupported("encoder layers with different feed-forward widths")
        FF = ff
    final_norm = int(pfx + "norm.weight" in sd)
    if final_norm:
        raw["encoder.norm.weight"], raw["encoder.norm.bias"] = sd[pfx + "norm.weight"], sd[pfx + "norm.bias"]
        used.update({pfx + "norm.weight", pfx + "norm.bias"})
    else:
        notes.append("no LayerNorm behind the last encoder layer")
    # ---- heads: [1, D] and [5, D]
    k1, w1 = find([r"\.weight$"], lambda s: tuple(s) == (1, D), "info head (1 x D)")
    k5, w5 = find([r"\.weight$"], lambda s: tuple(s) == (5, D), "base head (5 x D)")
    raw["info_head.weight"], raw["info_head.bias"] = w1, sd[k1[: -len("weight")] + "bias"]
    raw["base_head.weight"], raw["base_head.bias"] = w5, sd[k5[: -len("weight")] + "bias"]
    used.update({k1[: -len("weight")] + "bias", k5[: -len("weight")] + "bias"})
    # ---- position: sinusoidal, either computed in the graph (sin / cos) or a fixed table buffer
    for k in list(sd):
        if k in used:
            continue
        v = sd[k]
        if v.ndim >= 1 and v.shape[-1] == D // 2 and v.size == D // 2:       # the div_term vector
            if not np.allclose(v.reshape(-1), model_io.pe_div_term(D), rtol=1e-6):
                raise Unsupported(f"{k}: position frequencies are not 10000^(-2i/d)")
            used.add(k)
        elif v.ndim >= 2 and v.shape[-1] == D:                                # a [max_len, D] table
            t = v.reshape(-1, D)
            pos = np.arange(t.shape[0], dtype=np.float64)[:, None] * model_io.pe_div_term(D).astype(np.float64)[None, :]
            if np.allclose(t[:, 0::2], np.sin(pos), atol=1e-5) and np.allclose(t[:, 1::2], np.cos(pos), atol=1e-5):
                notes.append(f"{k}: sinusoidal table, reproduced on the device by sin / cos of the row index")
            else:
                if "pos_table" in raw:
                    raise Unsupported(f"{k}: a second position table")
                raw["pos_table"] = t
                notes.append(f"{k}: learned position table of {t.shape[0]} rows (windows longer than that are refused at run time, as the archive would raise)")
            used.add(k)
    left = [k for k in sd if k not in used]
    if left:
        raise Unsupported("tensors no rule claims: " + ", ".join(f"{k}{tuple(sd[k].shape)}" for k in left))
    # ---- attributes that shapes cannot tell: heads, eps, norm_first, activation
    n_heads, ln_eps, bn_eps, norm_first = None, 1e-5, 1e-5, None
    for name, sub in m.named_modules():
        on = getattr(sub, "original_name", "")
        if on == "MultiheadAttention" and hasattr(sub, "num_heads"):
            n_heads = int(sub.num_heads)
        if on == "TransformerEncoderLayer":
            if hasattr(sub, "norm_first"):
                nf = int(bool(sub.norm_first))
                if norm_first not in (None, nf):
                    raise Unsupported("encoder layers that differ in norm_first")
                norm_first = nf
            else:
                notes.append(f"{name}: norm_first not readable from the archive: assuming Pre-LN (the conversion check decides)")
        if on == "LayerNorm" and hasattr(sub, "eps"):
            ln_eps = float(sub.eps)
        if on == "BatchNorm2d" and hasattr(sub, "eps"):
            bn_eps = float(sub.eps)
    if n_heads is None:
        mm = re.search(r"num_heads\s*=\s*(\d+)|, (\d+), # num_heads", m.code if hasattr(m, "code") else "")
        n_heads = int(next(g for g in mm.groups() if g)) if mm else D // 32
        notes.append(f"num_heads not readable from the archive: assuming head_dim 32 -> {n_heads} heads")
    gk = gelu_kind(m)
    has_sin = any(k in ops for k in ("aten::sin", "aten::cos"))
    pe = 1 if "pos_table" in raw else (0 if has_sin else 2)
    if pe == 2:
        notes.append("no sin / cos and no position table in the archive: no position term")
    if gk:
        notes.append("GELU (" + ("tanh approximation" if gk == 2 else "erf") + ") in the encoder's feed-forward")
    if norm_first == 0:
        notes.append("Post-LN encoder layers (norm_first = False)")
    hp = model_io.Hyper(rows=31, emb=E, kw=kw, c1=C1, c2=C2, d_model=D, n_heads=n_heads, d_ff=FF, n_layers=len(layer_ids),
                        ln_eps=ln_eps, bn_eps=bn_eps, act=gk, norm_first=1 if norm_first is None else norm_first, pe=pe, final_norm=final_norm,
                        pe_rows=raw["pos_table"].shape[0] if pe == 1 else 0, bn=int(bool(bns)))
    return hp, {k: np.ascontiguousarray(v, np.float32) for k, v in raw.items()}, notes


def flop_report(hp, mean_rows: float = 4710.0, mean_tokens: float = 15.2) -> str:
    """2*M*N*K of every matmul, per window.  Dense = what the TorchScript graph executes over all
    L' x 31 cells; receptive-field = what the kernels execute (conv / linear only where an informative row needs them)."""
    R, kw, C1, C2, D, FF, NL = hp.rows, hp.kw, hp.c1, hp.c2, hp.d_model, hp.d_ff, hp.n_layers
    cin = hp.emb + 1
    h = kw // 2
    per_cell_conv = 2 * kw * cin * C1 + 2 * kw * C1 * C2
    per_row_fc = 2 * R * C2 * D
    per_tok_layer = 2 * D * 3 * D + 2 * D * D + 4 * D * FF
    dense = mean_rows * (R * per_cell_conv + per_row_fc)
    rf = mean_tokens * (R * ((2 * h + 1) * 2 * kw * cin * C1 + 2 * kw * C1 * C2) + per_row_fc)
    enc = mean_tokens * NL * (per_tok_layer + 4 * mean_tokens * D) + mean_tokens * 2 * D * 6
    return "\n".join([
        "FLOP per window (2MNK per matmul):",
        f"  conv stack, per cell        2*{kw}*{cin}*{C1} + 2*{kw}*{C1}*{C2} = {per_cell_conv:,}",
        f"  linear over rows, per row   2*{R}*{C2}*{D} = {per_row_fc:,}",
        f"  encoder, per token & layer  2*{D}*{3 * D} + 2*{D}*{D} + 4*{D}*{FF} = {per_tok_layer:,}  (+ attention 4*T*{D})",
        f"  dense graph, L' = {mean_rows:.0f}:              conv+linear {dense / 1e9:8.2f} GFLOP",
        f"  receptive-field form, T = {mean_tokens:.1f}:      conv+linear {rf / 1e9:8.3f} GFLOP   ({dense / max(rf, 1):.0f}x less)",
        f"  encoder + heads, T = {mean_tokens:.1f}, {NL} layers:  {enc / 1e9:8.3f} GFLOP",
    ])


def verify(m, hp, raw, seed: int = 0) -> float:
    """archive (CPU, fp32) vs numpy evaluation of the folded tensors in the kernels' dataflow; returns max |diff|"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import model_numpy as MN
    F = model_io.fold(raw, hp)
    rng = np.random.default_rng(seed)
    B, L = 3, 40
    win_len = np.array([40, 33, 37])
    bases = rng.integers(0, 11, (B, L, 31)).astype(np.uint8)
    quals = rng.integers(33, 90, (B, L, 31)).astype(np.uint8)
    for b in range(B):
        bases[b, win_len[b]:], quals[b, win_len[b]:] = 11, 126
    idx = [np.array([0, 1, 5, 20, 38, 39]), np.array([0, 2, 31, 32]), np.array([3, 35, 36])]
    lens = np.array([len(i) for i in idx], np.int32)
    qn = torch.from_numpy(MN.norm_qual(quals))
    with torch.no_grad():
        ti, tb = m(torch.from_numpy(bases.astype(np.int32)), qn, torch.from_numpy(lens), [torch.from_numpy(i.astype(np.int32)) for i in idx])
    ni, nb = MN.forward(F, hp, bases, quals, lens, np.concatenate(idx).astype(np.int32), win_len=win_len)
    return float(max(np.abs(ti.numpy() - ni).max(), np.abs(tb.numpy() - nb).max()))


def convert(path: str, out: str, dump: str | None = None, do_verify: bool = True, quiet: bool = False):
    m = load_archive(path)
    report = describe(m)
    if dump:
        open(dump, "w").write(report + "\n")
    problems = check_graph(m)
    if problems:
        raise Unsupported("the archive's graph is not the architecture the kernels implement:\n  " + "\n  ".join(problems))
    hp, raw, notes = recover(m)
    kernel_limits = []
    if hp.d_model % hp.n_heads or hp.d_model // hp.n_heads not in (32, 64) or hp.d_model % 64 or (hp.kw * hp.c1) % 32 or (hp.rows * hp.c2) % 32 or hp.d_ff % 32 or hp.c2 % 16 or hp.n_layers > 16:
        kernel_limits.append("herro_load_model accepts head_dim 32 / 64, d_model % 64 == 0, kw*c1 / 31*c2 / d_ff multiples of 32, <= 16 layers")
    if kernel_limits:
        raise Unsupported("; ".join(kernel_limits) + f" — recovered {hp}")
    # the conversion check runs BEFORE anything is written: a file that fails it must not be left where it can be loaded
    err = None
    if do_verify:
        err = verify(m, hp, raw)
        if not quiet:
            print(f"verify: archive vs folded tensors, max |logit diff| = {err:.2e}")
        if not err <= 2e-5:
            raise Unsupported(f"conversion check failed: max |logit diff| {err:.3e} > 2e-5 — the graph computes something the mapping does not")
    elif any("not readable from the archive" in n for n in notes):
        raise Unsupported("an attribute the shapes cannot tell (norm_first / eps / num_heads) is not readable from the archive and the "
                          "conversion check is off (--no-verify): refusing to guess — " + "; ".join(notes))
    model_io.export(raw, hp, out + ".tmp")
    os.replace(out + ".tmp", out)
    if not quiet:
        print(f"recovered {hp}")
        for n in notes:
            print("note:", n)
        dflt = (hp.act, hp.norm_first, hp.pe, hp.final_norm) == (0, 1, 0, 1)
        fast = dflt and hp.kw == 3 and hp.c1 == 64 and hp.c2 == 128 and hp.d_model == 256 and hp.n_heads == 8 and hp.d_ff % 256 == 0
        print("kernel path:", "f16 MFMA conv / FC / encoder kernels (precision 4 .. 8, chosen by the load-time calibration)" if fast else
              "generic bf16x3 kernels, layer by layer (precision 1 / 3): " + ("shapes differ from the tuned ones" if dflt else "a variant of the family the fused kernels do not implement"))
        print(flop_report(hp))
        print(f"wrote {out} ({os.path.getsize(out):,} bytes)")
    return hp, raw, err


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("archive")
    ap.add_argument("out")
    ap.add_argument("--dump", help="write code / graph / module tree / parameter list here")
    ap.add_argument("--verify", action="store_true", help="(default) run the archive against the folded tensors before writing")
    ap.add_argument("--no-verify", action="store_true", help="skip the conversion check")
    a = ap.parse_args()
    try:
        convert(a.archive, a.out, a.dump, not a.no_verify)
    except Unsupported as e:
        print(f"export_weights: cannot convert {a.archive}:\n{e}", file=sys.stderr)
        raise SystemExit(2)


if __name__ == "__main__":
    main()
