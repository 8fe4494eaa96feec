"""ORACLE — TEST INFRASTRUCTURE ONLY.

Plain PyTorch fp32 twin of the ASSUMED correction-model architecture, executed densely exactly
the way the reference drives its TorchScript model (reference inference.rs:147-175):

    forward(bases:int32[B,L,31], quals:f32[B,L,31], lens:int32[B], indices:List[int32[len_i]])
        -> (info_logits f32[N], bases_logits f32[N,5]),  N = sum(lens)

PARITY UNPINNED: the real model (`model_R10_v0.1.pt`, Zenodo 12683277, reference
README.md:56-66) is not in /root/reference and its internals are not visible in the repo; the
reference's own model test is commented out and its fixtures are absent
(inference.rs:302-410).  This twin pins the HIP kernels to *an* fp32 PyTorch execution of the
documented architecture, not to the published weights.

The twin is deliberately dense (embedding -> conv over every cell -> linear at every position ->
gather) — it is what a TorchScript executor would do, and it is the check that the product's
receptive-field evaluation is exact, including the batch-padding cells (token 11, quality 126)
that `collate` adds (inference.rs:86-97).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

QUAL_MIN, QUAL_MAX = 33.0, 126.0  # inference.rs:16-17


def normalise_quals(q_u8: torch.Tensor) -> torch.Tensor:
    """inference.rs:19-21,153: QUAL_SCALE * quals - QUAL_OFFSET on an f32 tensor."""
    diff = float(np.float32(QUAL_MAX) - np.float32(QUAL_MIN))
    scale = 2.0 / diff
    offset = 2.0 * float(np.float32(QUAL_MIN)) / diff + 1.0
    return scale * q_u8.to(torch.float32) - offset


class HerroNet(nn.Module):
    def __init__(self, hp):
        super().__init__()
        self.hp = hp
        self.embedding = nn.Embedding(12, hp.emb, padding_idx=11)
        cin = hp.emb + 1
        pad = (hp.kw // 2, 0)
        # variants of the family (herro_amd.model_io.Hyper, round 6): BatchNorm or none, ReLU / GELU, Pre- / Post-LN, final LayerNorm or none, position term
        bn = getattr(hp, "bn", 1)
        self.conv1 = nn.Conv2d(cin, hp.c1, (hp.kw, 1), padding=pad)
        self.bn1 = nn.BatchNorm2d(hp.c1, eps=hp.bn_eps) if bn else nn.Identity()
        self.conv2 = nn.Conv2d(hp.c1, hp.c2, (hp.kw, 1), padding=pad)
        self.bn2 = nn.BatchNorm2d(hp.c2, eps=hp.bn_eps) if bn else nn.Identity()
        self.fc = nn.Linear(hp.rows * hp.c2, hp.d_model)
        act = getattr(hp, "act", 0)
        activation = {0: "relu", 1: "gelu"}.get(act) or (lambda t: F.gelu(t, approximate="tanh"))
        layer = nn.TransformerEncoderLayer(hp.d_model, hp.n_heads, hp.d_ff, dropout=0.0, activation=activation,
                                           layer_norm_eps=hp.ln_eps, batch_first=True, norm_first=bool(getattr(hp, "norm_first", 1)))
        self.encoder = nn.TransformerEncoder(layer, hp.n_layers, norm=nn.LayerNorm(hp.d_model, eps=hp.ln_eps) if getattr(hp, "final_norm", 1) else None,
                                             enable_nested_tensor=False)
        self.info_head = nn.Linear(hp.d_model, 1)
        self.base_head = nn.Linear(hp.d_model, 5)
        from herro_amd.model_io import pe_div_term  # a constant table, shared verbatim
        self.register_buffer("pe_div", torch.from_numpy(pe_div_term(hp.d_model)))
        self.pe_kind = getattr(hp, "pe", 0)
        if self.pe_kind == 1:
            self.pos_table = nn.Parameter(torch.zeros(hp.pe_rows, hp.d_model))

    def load_raw(self, raw: dict):
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in raw.items()}
        sd["pe_div"] = self.pe_div
        if getattr(self.hp, "bn", 1):
            for n in ("bn1", "bn2"):
                sd[f"{n}.num_batches_tracked"] = torch.tensor(0)
        self.load_state_dict(sd)
        return self

    def positional(self, idx: torch.Tensor) -> torch.Tensor:
        if self.pe_kind == 1:
            return self.pos_table[idx.long()]      # (an index beyond the table raises, as it would in an archive)
        if self.pe_kind == 2:
            return torch.zeros(idx.shape[0], self.hp.d_model)
        ang = idx.to(torch.float32)[:, None] * self.pe_div[None, :]  # f32 product, as on device
        pe = torch.zeros(idx.shape[0], self.hp.d_model)
        pe[:, 0::2] = torch.sin(ang)
        pe[:, 1::2] = torch.cos(ang)
        return pe

    @torch.no_grad()
    def forward(self, bases, quals, lens, indices):
        B, L, R = bases.shape
        x = torch.cat([self.embedding(bases.long()), quals[..., None]], dim=-1)  # [B,L,R,7]
        x = x.permute(0, 3, 1, 2)                                                # [B,7,L,R]
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))                                      # [B,C2,L,R]
        x = x.permute(0, 2, 3, 1).reshape(B, L, R * self.hp.c2)
        x = self.fc(x)                                                           # [B,L,D]
        lens_l = [int(v) for v in lens]
        tmax = max(lens_l) if lens_l else 0
        if tmax == 0:
            return torch.zeros(0), torch.zeros(0, 5)
        toks = torch.zeros(B, tmax, self.hp.d_model)
        mask = torch.ones(B, tmax, dtype=torch.bool)  # True = padding
        for i in range(B):
            if lens_l[i]:
                idx = indices[i].long()
                toks[i, : lens_l[i]] = x[i, idx] + self.positional(idx)
                mask[i, : lens_l[i]] = False
        # windows without informative positions never reach the model in the reference
        # (inference.rs:243); keep them out of attention to avoid all-masked rows
        keep = [i for i in range(B) if lens_l[i] > 0]
        y = self.encoder(toks[keep], src_key_padding_mask=mask[keep])
        y = y[~mask[keep]]
        return self.info_head(y).squeeze(-1), self.base_head(y)


    # ---------------------------------------------------------------------------------------------------------
    # The same function written with matmul / layer_norm / softmax only, on whatever device the parameters live
    # on.  This is what the -m gpu end-to-end test runs ON THE GPU at the BASELINE sizes (4096-bp windows, batch
    # 64 / 128): the dense CPU module above manages 0.4 windows/s there, and the conv / fused-attention back ends
    # (MIOpen, SDPA) are not something a parity check should depend on.  tests/test_model_design.py pins it to
    # forward() on the CPU.  Differences from forward(), none of them numerical beyond f32 rounding order:
    # conv = kw shifted matmuls over a zero-padded window axis, BatchNorm(eval) applied as its affine map,
    # the per-position linear evaluated on the gathered rows only (a row-wise op commutes with the gather).
    @torch.no_grad()
    def forward_gemm(self, bases, quals, lens, indices, win_chunk: int = 8):
        hp = self.hp
        assert (getattr(hp, "act", 0), getattr(hp, "norm_first", 1), getattr(hp, "pe", 0), getattr(hp, "final_norm", 1), getattr(hp, "bn", 1)) == (0, 1, 0, 1, 1), \
            "forward_gemm is written for the default variant (the end-to-end test's model); forward() serves every variant"
        dev = self.fc.weight.device
        B, L, R = bases.shape
        kw, h, D, H = hp.kw, hp.kw // 2, hp.d_model, hp.n_heads
        dh = D // H

        def bn(x, m):   # channel-last
            return (x - m.running_mean) / torch.sqrt(m.running_var + m.eps) * m.weight + m.bias

        def conv(x, m):  # x [b, L, R, cin] -> [b, L, R, cout]; Conv2d((kw, 1), padding (h, 0)) along L
            w = m.weight[:, :, :, 0]                        # [cout, cin, kw]
            xp = F.pad(x, (0, 0, 0, 0, h, h))               # zero rows before / after the window axis
            acc = None
            for t in range(kw):
                term = xp[:, t:t + L] @ w[:, :, t].T
                acc = term if acc is None else acc + term
            return acc + m.bias

        lens_l = [int(v) for v in lens]
        keep = [i for i in range(B) if lens_l[i] > 0]
        if not keep:
            return torch.zeros(0), torch.zeros(0, 5)
        tmax = max(lens_l)
        toks = torch.zeros(len(keep), tmax, D, device=dev)
        mask = torch.ones(len(keep), tmax, dtype=torch.bool, device=dev)
        for c0 in range(0, len(keep), win_chunk):
            sel = keep[c0:c0 + win_chunk]
            bb = bases[sel].to(dev).long()
            qq = quals[sel].to(dev)
            x = torch.cat([self.embedding(bb), qq[..., None]], dim=-1)       # [b, L, R, 7]
            x = F.relu(bn(conv(x, self.conv1), self.bn1))
            x = F.relu(bn(conv(x, self.conv2), self.bn2))                    # [b, L, R, c2]
            for k, i in enumerate(sel):
                idx = indices[i].to(dev).long()
                rows = x[k, idx].reshape(idx.shape[0], R * hp.c2)            # index = row * c2 + c, as forward()
                ang = idx.to(torch.float32)[:, None] * self.pe_div[None, :]
                pe = torch.zeros(idx.shape[0], D, device=dev)
                pe[:, 0::2] = torch.sin(ang)
                pe[:, 1::2] = torch.cos(ang)
                toks[c0 + k, :lens_l[i]] = rows @ self.fc.weight.T + self.fc.bias + pe
                mask[c0 + k, :lens_l[i]] = False
            del x
        y = toks
        neg = mask[:, None, None, :]
        for layer in self.encoder.layers:
            a = layer.self_attn
            h1 = F.layer_norm(y, (D,), layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
            qkv = h1 @ a.in_proj_weight.T + a.in_proj_bias
            q, k, v = (t.reshape(t.shape[0], tmax, H, dh).transpose(1, 2) for t in qkv.split(D, dim=-1))
            sc = (q * (dh ** -0.5)) @ k.transpose(-1, -2)
            sc = sc.masked_fill(neg, float("-inf"))
            o = (torch.softmax(sc, dim=-1) @ v).transpose(1, 2).reshape(-1, tmax, D)
            y = y + o @ a.out_proj.weight.T + a.out_proj.bias
            h2 = F.layer_norm(y, (D,), layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
            y = y + F.relu(h2 @ layer.linear1.weight.T + layer.linear1.bias) @ layer.linear2.weight.T + layer.linear2.bias
        n = self.encoder.norm
        y = F.layer_norm(y, (D,), n.weight, n.bias, n.eps)[~mask]
        return (y @ self.info_head.weight.T + self.info_head.bias).squeeze(-1), y @ self.base_head.weight.T + self.base_head.bias


def build(raw: dict, hp) -> HerroNet:
    torch.manual_seed(0)
    m = HerroNet(hp).load_raw(raw)
    m.eval()
    return m


def run_batch(model: HerroNet, bases_u8: np.ndarray, quals_u8: np.ndarray, lens: np.ndarray, indices_flat: np.ndarray,
              gemm: bool = False):
    """Drive the twin exactly like `inference` (inference.rs:147-175): raw u8 in, logits out.
    gemm=True evaluates forward_gemm (device-aware; the model may live on the GPU)."""
    b = torch.from_numpy(bases_u8.astype(np.int32))
    q = normalise_quals(torch.from_numpy(quals_u8))
    idx, o = [], 0
    for n in lens:
        idx.append(torch.from_numpy(indices_flat[o:o + int(n)].astype(np.int32)))
        o += int(n)
    with torch.no_grad():
        if gemm:
            info, base = model.forward_gemm(b, q, torch.from_numpy(lens.astype(np.int32)), idx)
        else:
            info, base = model(b, q, torch.from_numpy(lens.astype(np.int32)), idx)
    return info.cpu().numpy(), base.cpu().numpy()
