"""A torch.jit.script-able module with the reference's model interface
    forward(bases:int32[B,L,31], quals:f32[B,L,31], lens:int32[B], indices:List[int32[len_i]]) -> (info[N], bases[N,5])
(reference inference.rs:155-163) and the parameter names of oracle/model_ref.HerroNet.  Saved with torch.jit.save it
is the stand-in for `model_R10_v0.1.pt` that tools/export_weights.py is tested on: archive -> tool -> flat file ->
HIP logits against torch.jit.load(archive) on the CPU.  Test infrastructure only."""
from typing import List, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class HerroScript(nn.Module):
    __constants__ = ["pe_learned", "pe_none", "conv_tanh", "d_model"]   # (constant conditions: the branches a variant does not take are not compiled, their attributes need not exist)

    def __init__(self, hp, activation: str = "", norm_first=None, extra_param: bool = False, conv_act: str = "relu"):
        super().__init__()
        # the variant comes from hp (herro_amd.model_io.Hyper) unless the caller overrides it; conv_act = "tanh" makes a genuinely foreign graph
        activation = activation or {0: "relu", 1: "gelu"}[getattr(hp, "act", 0)]
        norm_first = bool(getattr(hp, "norm_first", 1)) if norm_first is None else norm_first
        self.pe_kind = int(getattr(hp, "pe", 0))
        self.pe_learned, self.pe_none = self.pe_kind == 1, self.pe_kind == 2
        self.conv_tanh = conv_act == "tanh"
        self.d_model = hp.d_model
        self.embedding = nn.Embedding(12, hp.emb, padding_idx=11)
        cin = hp.emb + 1
        pad = (hp.kw // 2, 0)
        self.conv1 = nn.Conv2d(cin, hp.c1, (hp.kw, 1), padding=pad)
        bn = getattr(hp, "bn", 1)
        self.bn1 = nn.BatchNorm2d(hp.c1, eps=hp.bn_eps) if bn else nn.Identity()
        self.conv2 = nn.Conv2d(hp.c1, hp.c2, (hp.kw, 1), padding=pad)
        self.bn2 = nn.BatchNorm2d(hp.c2, eps=hp.bn_eps) if bn else nn.Identity()
        self.fc = nn.Linear(hp.rows * hp.c2, hp.d_model)
        layer = nn.TransformerEncoderLayer(hp.d_model, hp.n_heads, hp.d_ff, dropout=0.0, activation=activation,
                                           layer_norm_eps=hp.ln_eps, batch_first=True, norm_first=norm_first)
        self.encoder = nn.TransformerEncoder(layer, hp.n_layers, norm=nn.LayerNorm(hp.d_model, eps=hp.ln_eps) if getattr(hp, "final_norm", 1) else None,
                                             enable_nested_tensor=False)
        if self.pe_kind == 1:
            self.pos_table = nn.Parameter(torch.zeros(int(hp.pe_rows), hp.d_model))
        self.info_head = nn.Linear(hp.d_model, 1)
        self.base_head = nn.Linear(hp.d_model, 5)
        from herro_amd.model_io import pe_div_term
        self.register_buffer("pe_div", torch.from_numpy(pe_div_term(hp.d_model)))
        if extra_param:
            self.mystery = nn.Parameter(torch.zeros(7, 3))

    def positional(self, idx: torch.Tensor) -> torch.Tensor:
        if self.pe_learned:
            return self.pos_table.index_select(0, idx)
        if self.pe_none:
            return torch.zeros(idx.shape[0], self.d_model)
        ang = idx.to(torch.float32).unsqueeze(1) * self.pe_div.unsqueeze(0)
        pe = torch.zeros(idx.shape[0], self.d_model)
        pe[:, 0::2] = torch.sin(ang)
        pe[:, 1::2] = torch.cos(ang)
        return pe

    def forward(self, bases: torch.Tensor, quals: torch.Tensor, lens: torch.Tensor, indices: List[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        B, L = bases.shape[0], bases.shape[1]
        x = torch.cat([self.embedding(bases.long()), quals.unsqueeze(-1)], dim=-1)   # [B,L,R,7]
        x = x.permute(0, 3, 1, 2)
        if self.conv_tanh:
            x = torch.tanh(self.bn1(self.conv1(x)))
        else:
            x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = x.permute(0, 2, 3, 1).reshape(B, L, -1)
        x = self.fc(x)
        lens_l: List[int] = lens.to(torch.int64).tolist()
        tmax = 0
        for n in lens_l:
            tmax = max(tmax, n)
        toks = torch.zeros(B, tmax, self.d_model)
        mask = torch.ones(B, tmax, dtype=torch.bool)
        for i in range(B):
            n = lens_l[i]
            if n > 0:
                idx = indices[i].long()
                toks[i, :n] = x[i].index_select(0, idx) + self.positional(idx)
                mask[i, :n] = False
        y = self.encoder(toks, src_key_padding_mask=mask)
        y = y[~mask]
        return self.info_head(y).squeeze(-1), self.base_head(y)


def save_archive(path: str, raw: dict, hp, **kw) -> None:
    """random-init (or `raw`) HerroScript -> TorchScript archive on disk"""
    torch.manual_seed(0)
    m = HerroScript(hp, **kw)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in raw.items()}
    sd["pe_div"] = m.pe_div
    if getattr(hp, "bn", 1):
        for n in ("bn1", "bn2"):
            sd[f"{n}.num_batches_tracked"] = torch.tensor(0)
    if kw.get("extra_param"):
        sd["mystery"] = m.mystery.detach()
    m.load_state_dict(sd)
    m.eval()
    torch.jit.save(torch.jit.script(m), path)
