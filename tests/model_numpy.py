"""numpy mirror of the HIP model dataflow (model.hip), fed with the *folded* tensors of the flat
weight file.  CPU check of the export folding, layout conventions and the receptive-field
evaluation against the dense PyTorch twin.  Test helper only."""
import numpy as np


def norm_qual(q):
    qs, qo = np.float32(2.0 / 93.0), np.float32(2.0 * 33.0 / 93.0 + 1.0)
    return (qs * q.astype(np.float32)).astype(np.float32) - qo


def layernorm(x, g, b, eps):
    m = x.mean(-1, keepdims=True)
    v = ((x - m) ** 2).mean(-1, keepdims=True)
    return (x - m) / np.sqrt(v + eps) * g + b


def forward(F, hp, bases, quals, lens, indices, win_len=None):
    """bases/quals u8 [B,L,31]; win_len[b] = true length (cells beyond are batch padding)."""
    B, L, R = bases.shape
    kw, c1, c2, D = hp.kw, hp.c1, hp.c2, hp.d_model
    h = kw // 2
    lmax = L
    toks = []
    o = 0
    for b in range(B):
        for k in range(int(lens[b])):
            toks.append((b, int(indices[o + k])))
        o += int(lens[b])
    N = len(toks)
    y1 = np.zeros((N, R, kw, c1), np.float32)
    t1, wq1, b1 = F["t1"], F["wq1"], F["b1"]
    for n, (b, l) in enumerate(toks):
        ln = L if win_len is None else int(win_len[b])
        for dl in range(kw):
            pos = l + dl - h
            if pos < 0 or pos >= lmax:
                continue
            v = np.tile(b1, (R, 1)).astype(np.float32)
            for t in range(kw):
                q = pos + t - h
                if q < 0 or q >= lmax:
                    continue
                if q < ln:
                    tok = bases[b, q, :].astype(np.int64)
                    qn = norm_qual(quals[b, q, :])
                else:
                    tok = np.full(R, 11)
                    qn = norm_qual(np.full(R, 126, np.uint8))
                v = v + t1[t][tok] + wq1[t][None, :] * qn[:, None]
            y1[n, :, dl, :] = np.maximum(v, 0)
    a2 = y1.reshape(N * R, kw * c1)
    y2 = np.maximum(a2 @ F["conv2.wt"].T + F["conv2.b"], 0).reshape(N, R * c2)
    x = y2 @ F["fc.wt"].T + F["fc.b"]
    rows = np.array([l for _, l in toks], np.float32)
    pe_kind, pre, act = getattr(hp, "pe", 0), bool(getattr(hp, "norm_first", 1)), getattr(hp, "act", 0)
    if pe_kind == 0:
        ang = (rows[:, None] * F["pe_div"][None, :]).astype(np.float32)
        x[:, 0::2] += np.sin(ang)
        x[:, 1::2] += np.cos(ang)
    elif pe_kind == 1:
        x += F["pe_table"][rows.astype(np.int64)]

    def activation(t):
        if act == 0:
            return np.maximum(t, 0)
        if act == 1:
            from scipy.special import erf
            return 0.5 * t * (1.0 + erf(t / np.sqrt(2.0)))
        return 0.5 * t * (1.0 + np.tanh(0.7978845608028654 * (t + 0.044715 * t ** 3)))
    H, dh = hp.n_heads, D // hp.n_heads
    starts = np.concatenate([[0], np.cumsum(lens)]).astype(int)
    for li in range(hp.n_layers):
        p = f"L{li}."
        hb = layernorm(x, F[p + "ln1.g"], F[p + "ln1.b"], hp.ln_eps) if pre else x
        qkv = hb @ F[p + "qkv.wt"].T + F[p + "qkv.b"]
        att = np.zeros_like(x)
        for b in range(B):
            s, e = starts[b], starts[b + 1]
            if e == s:
                continue
            for hd in range(H):
                q = qkv[s:e, hd * dh:(hd + 1) * dh] / np.sqrt(dh)
                k = qkv[s:e, D + hd * dh:D + (hd + 1) * dh]
                v = qkv[s:e, 2 * D + hd * dh:2 * D + (hd + 1) * dh]
                sc = q @ k.T
                sc = np.exp(sc - sc.max(-1, keepdims=True))
                att[s:e, hd * dh:(hd + 1) * dh] = (sc / sc.sum(-1, keepdims=True)) @ v
        x = x + att @ F[p + "proj.wt"].T + F[p + "proj.b"]
        if pre:
            hb = layernorm(x, F[p + "ln2.g"], F[p + "ln2.b"], hp.ln_eps)
        else:
            x = hb = layernorm(x, F[p + "ln1.g"], F[p + "ln1.b"], hp.ln_eps)
        ff = activation(hb @ F[p + "ff1.wt"].T + F[p + "ff1.b"])
        x = x + ff @ F[p + "ff2.wt"].T + F[p + "ff2.b"]
        if not pre:
            x = layernorm(x, F[p + "ln2.g"], F[p + "ln2.b"], hp.ln_eps)
    hb = layernorm(x, F["lnf.g"], F["lnf.b"], hp.ln_eps) if getattr(hp, "final_norm", 1) else x
    lg = hb @ F["heads.wt"].T + F["heads.b"]
    return lg[:, 0], lg[:, 1:6]
