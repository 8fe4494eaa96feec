"""not-gpu: the host half of herro_job_create (CIGAR -> binary ops, windowing, validation, descriptor layout, the
parallel merge) through the device-free hook, against descriptors derived independently from the oracle's
`extract_windows` (windowing.rs:44-273) and `get_query_region` rules (features.rs:97-108)."""
import re

import numpy as np
import pytest

import oracle_lib as O
from herro_amd import api, synth

OPC = {b"M": 0, b"I": 1, b"D": 2}


def _ops(cigar: bytes):
    out, byte_lo, byte_hi = [], [], []
    for m in re.finditer(rb"(\d+)([MID])", cigar):
        out.append((int(m.group(1)) << 2) | OPC[m.group(2)])
        byte_lo.append(m.start())
        byte_hi.append(m.end())
    return out, byte_lo, byte_hi


def _expected(sb, W, targets):
    ops, ow, win, tiles, tgt_off = [], [], [], [], [0]
    n_cls_total = 0
    for t in targets:
        rid = int(sb.tgt_rid[t])
        tlen = int(sb.off[rid + 1] - sb.off[rid])
        nwin = (tlen + W - 1) // W
        per_win = [[] for _ in range(nwin)]
        ins_sum = [0] * nwin
        cls_of = {}
        win_base = len(win)
        for a in range(int(sb.tgt_aln_off[t]), int(sb.tgt_aln_off[t + 1])):
            row = tuple(int(x) for x in sb.aln[a, :9])
            cig = sb.cigar(a)
            bops, blo, bhi = _ops(cig)
            rows = O.extract_windows(row, cig, nwin, W).tolist()
            op_base = len(ops)
            if rows:
                ops.extend(bops)
            qid, strand = row[0], row[4]
            if qid not in cls_of:
                cls_of[qid] = len(cls_of)
            for w, ts, qs, qe, b0, so, b1, eo in rows:
                lo = blo.index(b0)
                hi = bhi.index(b1) + 1
                qbeg = row[2] + qs if strand == 0 else row[3] - qe          # features.rs:97-108
                per_win[w].append(dict(win=win_base + w, qid=qid, cls=n_cls_total + cls_of[qid], tstart=ts, qbeg=qbeg, qlen=qe - qs,
                                       op_begin=op_base + lo, op_cnt=hi - lo, start_off=so, end_off=eo, strand=strand,
                                       wtstart=w * W, wlen=min(W, tlen - w * W)))
                for k in range(lo, hi):                                      # insertion bases of the slice (for the row bound)
                    if bops[k] & 3 == 1:
                        ins_sum[w] += bops[k] >> 2
        for w in range(nwin):
            wl = min(W, tlen - w * W)
            lub = (wl + min(ins_sum[w], 50 * wl) + 15) & ~15
            win.append(dict(rid=rid, wid=w, n_wids=nwin, tstart=w * W, win_len=wl, ow_begin=len(ow), ow_cnt=len(per_win[w]), lub=lub))
            ow.extend(per_win[w])
            tiles.extend((len(win) - 1, r0) for r0 in range(0, lub, 1024))
        n_cls_total += len(cls_of)
        tgt_off.append(len(win))
    return ops, ow, win, tiles, tgt_off


@pytest.mark.parametrize("W,tl,nt,kw,threads", [
    (64, 700, 5, dict(p_partial=0.4, flank_min=10, flank_max=40), "1"),
    (256, 1500, 7, dict(p_partial=0.3, flank_min=20, flank_max=60, p_long_indel=0.02), "4"),
    (4096, 2 * 4096 + 333, 3, dict(p_partial=0.2), "3"),
])
def test_descriptors_match_oracle_windowing(monkeypatch, W, tl, nt, kw, threads):
    monkeypatch.setenv("HERRO_HOST_THREADS", threads)
    sb = synth.generate(nt, tl, 10, seed=W + tl, **kw)
    lens = (sb.off[1:] - sb.off[:-1]).astype(np.uint32)
    c = api.HostContext(lens)
    job = api.job_from_synth(c, sb, W)
    got = c.job_arrays(job)
    ops, ow, win, tiles, tgt_off = _expected(sb, W, range(sb.n_targets))
    assert got["ops"].tolist() == ops
    assert got["tgt_win_off"].tolist() == tgt_off
    assert len(got["ow"]) == len(ow) and len(got["win"]) == len(win)
    scr = 0
    word_off = np.concatenate([[0], np.cumsum((lens.astype(np.uint64) + 31) // 32)])
    qual_off = np.concatenate([[0], np.cumsum(lens.astype(np.uint64))])
    for g, e in zip(got["ow"], ow):
        for k, v in e.items():
            assert int(g[k]) == v, (k, int(g[k]), v)
        scr += e["op_cnt"]
        rid = win[e["win"]]["rid"]
        assert int(g["t_woff"]) == int(word_off[rid]) and int(g["q_woff"]) == int(word_off[e["qid"]])
        assert int(g["q_qual_off"]) == int(qual_off[e["qid"]])
    # per-overlap scratch slices (op tables on the device): disjoint, covering [0, sum of op counts)
    iv = sorted((int(g["scr_off"]), int(g["op_cnt"])) for g in got["ow"])
    end = 0
    for o, n in iv:
        assert o == end
        end = o + n
    assert end == scr
    fin = row = pos = 0
    for g, e in zip(got["win"], win):
        for k, v in e.items():
            assert int(g[k]) == v, (k, int(g[k]), v)
        assert (int(g["fin_off"]), int(g["row_off"]), int(g["pos_off"])) == (fin, row, pos)
        fin += 31 * e["lub"]; row += e["lub"]; pos += W + 1
    assert list(zip(got["tile_win"].tolist(), got["tile_r0"].tolist())) == tiles
    job.close()
    c.close()


def test_rejections_without_a_device():
    sb = synth.generate(2, 600, 6, seed=1, flank_min=20, flank_max=40)
    lens = (sb.off[1:] - sb.off[:-1]).astype(np.uint32)
    c = api.HostContext(lens)
    rows = sb.aln[int(sb.tgt_aln_off[0]):int(sb.tgt_aln_off[1])].copy()
    cigs = [sb.cigar(a) for a in range(int(sb.tgt_aln_off[0]), int(sb.tgt_aln_off[1]))]
    off = np.array([0, len(rows)], np.uint64)
    # what parse_paf would have dropped (a second alignment of the same (query, target) pair, a self overlap) is left
    # out and counted, it no longer fails the job: the descriptors are those of the input without these alignments
    bad = np.concatenate([rows[:2], rows[:1], rows[2:]]); bad[2, 1:5] = rows[1, 1:5]      # rows[0]'s pair again, in third place
    selfo = rows[:1].copy(); selfo[0, 0] = selfo[0, 5]                                    # qid == tid
    bad = np.concatenate([bad, selfo])
    bad_cigs = cigs[:2] + [cigs[1]] + cigs[2:] + [cigs[0]]
    off_bad = np.array([0, len(bad)], np.uint64)
    jb = c.create_job(sb.tgt_rid[:1], bad, off_bad, bad_cigs, 128)
    jg = c.create_job(sb.tgt_rid[:1], rows, off, cigs, 128)
    assert jb.skipped() == (2, 0) and jg.skipped() == (0, 0)
    ab, ag = c.job_arrays(jb), c.job_arrays(jg)
    for k in ag:
        assert np.array_equal(ab[k], ag[k]), k
    jb.close(); jg.close()
    with pytest.raises(api.HerroError) as e:                       # a CIGAR op the reference panics on
        c.create_job(sb.tgt_rid[:1], rows, off, [cigs[0].replace(b"M", b"X", 1)] + cigs[1:], 128)
    assert e.value.code == -3
    with pytest.raises(api.HerroError):
        c.create_job(sb.tgt_rid[:1], rows, off, cigs, 8)           # window size out of range
    c.close()


def test_prepared_alignments_give_the_same_jobs():
    """api.PreparedAlignments (one resident herro_alignment array, jobs over target ranges by pointer offset — the bench's
    end_to_end leg) builds exactly the descriptors job_from_synth builds."""
    sb = synth.generate(6, 2000, 8, seed=3, flank_min=30, flank_max=60)
    lens = (sb.off[1:] - sb.off[:-1]).astype(np.uint32)
    c = api.HostContext(lens)
    prep = api.PreparedAlignments(sb)
    for t0, t1 in ((0, 6), (2, 5), (5, 6)):
        j1, j2 = prep.job(c, t0, t1, 256), api.job_from_synth(c, sb, 256, range(t0, t1))
        a1, a2 = c.job_arrays(j1), c.job_arrays(j2)
        assert all(np.array_equal(a1[k], a2[k]) for k in a1)
        j1.close(); j2.close()
    c.close()


def _hand_alignment(rng, tlen, widths):
    """one '+' alignment of query 1 onto target 0 from position 0: random M / I / D ops (never two insertions in a row,
    never a leading insertion), some matches long enough to need 5..7 digits; text written with `widths` zero padding"""
    ops, t, q, prev = [], 0, 0, "I"
    while t < tlen - 3000:
        ty = "M" if prev != "M" else rng.choice(["I", "D"])
        ln = int(rng.choice([1, 2, 7, 35, 120, 999, 1000, 9999, 10000, 12345])) if ty == "M" else int(rng.integers(1, 40))
        if ty == "M":
            ln = min(ln, tlen - 3000 - t + 1)
        ops.append((ln, ty))
        t += ln if ty != "I" else 0
        q += ln if ty != "D" else 0
        prev = ty
    if ops[-1][1] == "I":
        ops.pop(); q -= ops and 0  # (the popped insertion is re-counted below)
    t = sum(l for l, y in ops if y != "I"); q = sum(l for l, y in ops if y != "D")
    text = "".join(f"{l:0{int(rng.choice(widths))}d}{y}" for l, y in ops).encode()
    canon = "".join(f"{l}{y}" for l, y in ops).encode()
    row = np.array([[1, 40000, 0, q, 0, 0, tlen, 0, t]], np.uint32)
    return row, text, canon


@pytest.mark.parametrize("W", [64, 1000, 4096])
def test_cigar_text_decoder_corner_cases(W):
    """The two-stage text decoder of herro_job_create (windowing.hpp scan_cigar): 1..4 digit lengths (32-bit lane),
    5..7 digits (64-bit lane), 8+ digits / zero padding / short text (byte-wise path) all give the ops a regular
    expression reads, and the same descriptors as the canonical spelling; malformed text fails like CigarIter."""
    rng = np.random.default_rng(5)
    lens = np.array([40000, 40000], np.uint32)
    c = api.HostContext(lens)
    rid = np.array([0], np.uint32)
    off = np.array([0, 1], np.uint64)
    for widths in ([1], [1, 3], [1, 5, 7], [1, 8, 9], [4], [7], [12]):
        row, text, canon = _hand_alignment(rng, 40000, widths)
        j1 = c.create_job(rid, row, off, [text], W)
        j2 = c.create_job(rid, row, off, [canon], W)
        a1, a2 = c.job_arrays(j1), c.job_arrays(j2)
        assert a1["ops"].tolist() == _ops(canon)[0]
        assert len(a1["ow"]) > 0
        for k in a2:
            assert np.array_equal(a1[k], a2[k]), (widths, k)
        j1.close(); j2.close()
    # an alignment too short to give a window still has its text decoded (short strings take the byte-wise path)
    short = np.array([[1, 40000, 0, 5, 0, 0, 40000, 0, 5]], np.uint32)
    j = c.create_job(rid, short, off, [b"5M"], W)
    assert len(c.job_arrays(j)["ow"]) == 0
    j.close()
    row, text, canon = _hand_alignment(rng, 40000, [1])
    for bad, what in [(canon + b"12", "ends inside an op"), (b"M" + canon, "longer than 0"), (canon.replace(b"M", b"M0I", 1), "longer than 0"),
                      (canon.replace(b"M", b"m", 1), "Unexpected cigar operation"), (canon.replace(b"M", b":", 1), "Unexpected cigar operation"),
                      (canon.replace(b"M", b"\xc8", 1), "Unexpected cigar operation"), (canon.replace(b"D", b"=", 1), "Unexpected cigar operation"),
                      (b"99999999999M" + canon, "overflows 30 bits")]:
        with pytest.raises(api.HerroError) as e:
            c.create_job(rid, row, off, [bad], W)
        assert what in str(e.value), (bad[:30], str(e.value))
    c.close()


def test_token_tile_plan_packs_whole_windows_into_fewer_tiles():
    """herro_job_infer's window order for the fused stack: a permutation, every tile <= 64 tokens of whole windows, never more
    tiles than the batch order gives, and within 3 % of the bound ceil(tokens / 64) on the bench's row-count distribution."""
    rng = np.random.default_rng(5)
    for n, lo, hi in ((4096, 4, 31), (1000, 1, 65), (37, 60, 65), (1, 64, 65), (513, 1, 3)):
        cnt = rng.integers(lo, hi, n).astype(np.uint32)
        tiles_plain, order_plain = api.debug_tile_plan(cnt, packed=False)
        tiles, order = api.debug_tile_plan(cnt, packed=True)
        assert order_plain.tolist() == list(range(n))
        assert sorted(order.tolist()) == list(range(n))
        assert tiles <= tiles_plain
        # the greedy consecutive split over the packed order (what the kernels get)
        t, cur = 1, 0
        for c in cnt[order]:
            if cur + c > 64:
                t, cur = t + 1, 0
            cur += int(c)
        assert t == tiles
        assert tiles >= -(-int(cnt.sum()) // 64)
        if (lo, hi) == (4, 31):
            assert tiles <= 1.03 * cnt.sum() / 64 + 1
    with pytest.raises(api.HerroError):
        api.debug_tile_plan(np.array([3, 0, 2], np.uint32))
    with pytest.raises(api.HerroError):
        api.debug_tile_plan(np.array([65], np.uint32))


def _check_plan(cnt, n64, n32, order, tok):
    n = len(cnt)
    assert sorted(order.tolist()) == list(range(n))
    c = cnt[order].astype(np.int64)
    win_tok = np.concatenate([[0], np.cumsum(c)])
    assert tok[0] == 0 and tok[-1] == win_tok[-1] and len(tok) == n64 + n32 + 1
    assert np.all(np.isin(tok, win_tok)), "a tile boundary inside a window"
    size = np.diff(tok.astype(np.int64))
    assert np.all(size[:n64] <= 64) and np.all(size[n64:] <= 32) and np.all(size > 0)
    in_head = np.arange(n) < (np.searchsorted(win_tok, tok[n64]) if n64 + n32 else 0)
    assert np.all(c[~in_head] <= 32)
    return c, win_tok


def test_token_tile_plan_all_small_windows_in_half_tiles():
    """qmode 2 (the A/B plan of k_layers_q): a permutation; tiles are runs of WHOLE windows; the 64-token tiles come first, each
    opened by a window of 33..64 rows; every other window sits in a tile of <= 32 tokens; the 32-token tiles are within 4 % of
    ceil(tokens / 32) on the bench's distribution; the cost (a 64-token tile = two 32-token ones) never exceeds that of the
    64-token-only plan by more than a few percent."""
    rng = np.random.default_rng(6)
    for n, lo, hi, packed in ((4096, 4, 31, True), (2560, 4, 31, True), (1000, 1, 65, True), (1000, 1, 65, False), (37, 60, 65, True),
                              (64, 33, 34, True), (1, 64, 65, True), (1, 7, 8, True), (513, 1, 3, True), (300, 20, 45, True)):
        cnt = rng.integers(lo, hi, n).astype(np.uint32)
        n64, n32, order, tok = api.debug_tile_plan(cnt, packed=packed, qmode=2, bounds=True)
        c, win_tok = _check_plan(cnt, n64, n32, order, tok)
        first_win = np.searchsorted(win_tok, tok[:-1])
        assert np.all(c[first_win[:n64]] > 32), "a 64-token tile is opened by a large window"
        assert int((c > 32).sum()) == n64
        if (lo, hi) == (4, 31):
            assert n64 == 0 and n32 <= 1.04 * cnt.sum() / 32 + 1
        if packed:
            full, _ = api.debug_tile_plan(cnt, packed=True)
            assert 2 * n64 + n32 <= 1.04 * 2 * full + 1 + n64   # never materially worse than 64-token tiles only (smaller bins pack ~2 % looser)
    assert api.debug_tile_plan(np.array([], np.uint32), qmode=2)[:2] == (0, 0)


def test_token_tile_plan_short_last_round_in_half_tiles():
    """qmode 1 (herro_job_infer's default for the f16 stack): the 64-token plan, except that a last round filling at most half of
    the compute units is re-packed into 32-token tiles (one per compute unit, half the cost)."""
    rng = np.random.default_rng(7)
    for n, lo, hi, n_cu in ((2560, 4, 31, 256), (4096, 4, 31, 256), (700, 4, 31, 256), (100, 4, 31, 256), (3000, 4, 31, 304), (900, 20, 60, 64), (50, 40, 64, 8)):
        cnt = rng.integers(lo, hi, n).astype(np.uint32)
        full, _ = api.debug_tile_plan(cnt, packed=True)
        n64, n32, order, tok = api.debug_tile_plan(cnt, packed=True, qmode=1, n_cu=n_cu, bounds=True)
        _check_plan(cnt, n64, n32, order, tok)
        r = full % n_cu
        if r == 0 or 2 * r > n_cu:
            assert (n64, n32) == (full, 0)
        elif lo < 32:
            assert n64 % n_cu <= (cnt > 32).sum() and n64 <= full and n32 > 0
            assert n32 <= 2 * r + 2 + r // 8, (full, r, n64, n32)      # about two half tiles per tile of the short round
            rounds = lambda t64, t32: -(-t64 // n_cu) + 0.5 * -(-t32 // n_cu)
            assert rounds(n64, n32) < rounds(full, 0)
    # a launch like the bench's 2560 windows (625 tiles): 2 full rounds + a short one
    cnt = np.random.default_rng(1).integers(4, 27, 2560).astype(np.uint32)
    assert 600 <= api.debug_tile_plan(cnt, packed=True)[0] <= 640
    n64, n32, _ = api.debug_tile_plan(cnt, packed=True, qmode=1, n_cu=256)
    assert n64 == 512 and 150 <= n32 <= 256


def test_token_tile_plan_windows_above_64_rows_on_sibling_tiles():
    """f16 stack: a window of 65 .. 512 informative rows stays fused — on ceil(rows / 64) consecutive 64-token tiles at the head of the
    token stream (sibling tiles, one group per window), its last tile filled up with the best-fitting small windows; the other
    windows follow, planned as before, with the sibling tiles counted as busy compute units when the last round is sized."""
    rng = np.random.default_rng(9)
    for n_big, n_small, n_cu, packed in ((1, 0, 256, True), (3, 40, 256, True), (10, 1014, 256, True), (5, 507, 256, True), (40, 3, 64, True),
                                         (2, 2558, 256, True), (6, 300, 256, False)):
        big = rng.integers(65, 513, n_big)
        big[0] = 65
        if n_big > 1:
            big[1] = 512
        if n_big > 2:
            big[2] = 128
        small = rng.integers(4, 31, n_small)
        cnt = np.concatenate([big, small]).astype(np.uint32)
        cnt = cnt[rng.permutation(len(cnt))]
        nb, n64, n32, order, tok, grp = api.debug_tile_plan_sib(cnt, packed=packed, qmode=1, n_cu=n_cu)
        assert sorted(order.tolist()) == list(range(len(cnt)))
        assert nb == int(((cnt[cnt > 64].astype(np.int64) + 63) // 64).sum()) and len(grp) == nb
        so = cnt[order].astype(np.int64)
        win_tok = np.concatenate([[0], np.cumsum(so)])
        # the head of the stream: each large window (in the given order) on its own tiles of 64; its last tile may take small windows along
        assert order[so > 64].tolist() == np.flatnonzero(cnt > 64).tolist()
        k, w, guests = 0, 0, 0
        for _ in range(n_big):
            rows = so[w]
            assert rows > 64
            kk, last = -(-rows // 64), rows - 64 * (-(-rows // 64) - 1)
            for j in range(kk):
                assert tok[k + j] == win_tok[w] + 64 * j
                assert int(grp[k + j]) == (k | (kk << 20) | (last << 24))
            w += 1
            room = 64 - last
            while w < len(so) and so[w] <= 64 and win_tok[w + 1] <= tok[k + kk]:      # the guests of the last tile: whole windows
                room -= so[w]
                w += 1
                guests += 1
            assert room >= 0 and tok[k + kk] == win_tok[w]
            k += kk
        assert k == nb
        assert guests == 0 if not packed else (guests > 0 or n_small < 10)
        # behind them: whole small windows per tile, as a plan without sibling tiles would have them
        rest = so[w:].astype(np.uint32)
        assert (rest <= 64).all()
        _check_plan(rest, n64, n32, np.arange(len(rest), dtype=np.uint32), tok[nb:] - tok[nb])
        assert tok[-1] == cnt.sum()
        if len(rest) and packed:
            full, _ = api.debug_tile_plan(rest, packed=True)
            r = (full + nb) % n_cu                   # the sibling tiles head the grid of the 64-token tiles
            if r and 2 * r <= n_cu and r <= full:
                assert n32 > 0 and n64 <= full          # the round the sibling tiles make short goes into half tiles
            if r == 0 or 2 * r > n_cu:
                assert (n64, n32) == (full, 0)          # a last round more than half full stays in 64-token tiles
    with pytest.raises(api.HerroError):
        api.debug_tile_plan_sib(np.array([513], np.uint32))
    # the bench-like batch of tools/large_window_cost.py: 1014 small windows + 10 windows of 100 rows (20 sibling tiles, 28 free slots in every second one)
    cnt = np.clip(np.random.default_rng(3).normal(15.2, 4.5, 1024).round(), 4, 30).astype(np.uint32)
    cnt[:10] = 100
    nb, n64, n32, order, tok, grp = api.debug_tile_plan_sib(cnt, packed=True, qmode=1, n_cu=256)
    assert nb == 20 and (np.diff(tok[:21].astype(np.int64))[1::2] >= 36 + 20).all()      # the second tile of every pair took guests
    assert (nb + n64) % 256 == 0 and n32 > 0


def _check_sib_plan(cnt, packed, qmode, n_cu):
    """Invariants of a token-tile plan with sibling tiles, whatever the input: see the two tests above for what they mean."""
    cnt = np.asarray(cnt, np.uint32)
    nb, n64, n32, order, tok, grp = api.debug_tile_plan_sib(cnt, packed=packed, qmode=qmode, n_cu=n_cu)
    n = len(cnt)
    assert sorted(order.tolist()) == list(range(n))
    so = cnt[order].astype(np.int64)
    win_tok = np.concatenate([[0], np.cumsum(so)])
    assert len(tok) == nb + n64 + n32 + 1 and tok[0] == 0 and tok[-1] == so.sum()
    size = np.diff(tok.astype(np.int64))
    assert (size > 0).all() and (size[:nb + n64] <= 64).all() and (size[nb + n64:] <= 32).all()
    assert nb == int(((so[so > 64] + 63) // 64).sum())
    # sibling groups: consecutive tiles, the group word identical within a group, the window's tokens first
    k = 0
    w = 0
    while k < nb:
        g0, kk, last = int(grp[k]) & 0xfffff, (int(grp[k]) >> 20) & 15, int(grp[k]) >> 24
        assert g0 == k and 2 <= kk <= 8 and 1 <= last <= 64
        assert all(int(grp[k + j]) == int(grp[k]) for j in range(kk))
        while so[w] <= 64:                      # guests of the previous group
            w += 1
        rows = so[w]
        assert rows == 64 * (kk - 1) + last and tok[k] == win_tok[w]
        assert all(size[k + j] == 64 for j in range(kk - 1)) and size[k + kk - 1] >= last
        assert np.isin(tok[k + kk], win_tok)    # the guests are whole windows
        w += 1
        k += kk
    # everything behind the sibling tiles: tile boundaries on window boundaries
    assert np.isin(tok[nb:], win_tok).all()
    # a window of <= 64 rows never straddles tiles; a window above 32 rows is not in a 32-token tile
    t_of = np.searchsorted(tok, win_tok[:-1], side="right") - 1
    t_end = np.searchsorted(tok, win_tok[1:] - 1, side="right") - 1
    small = so <= 64
    assert (t_of[small] == t_end[small]).all()
    assert (so[(t_of >= nb + n64)] <= 32).all()


def test_token_tile_plan_with_sibling_tiles_property():
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=150, deadline=None)
    @given(st.lists(st.one_of(st.integers(1, 64), st.integers(1, 64), st.integers(65, 512)), min_size=0, max_size=400),
           st.booleans(), st.sampled_from([0, 1, 2]), st.sampled_from([1, 8, 64, 256, 304]))
    def run(cnt, packed, qmode, n_cu):
        _check_sib_plan(cnt, packed, qmode, n_cu)
    run()
    _check_sib_plan([512] * 300 + [1] * 50, True, 1, 256)      # more sibling tiles than compute units
    _check_sib_plan([65, 64, 64, 63, 1], True, 1, 256)


def test_limits_that_differ_from_the_reference_are_refused_loudly():
    """Inputs the reference accepts and this library does not come back as HERRO_E_UNSUPPORTED with a message — never as a
    wrong result: window sizes outside [16, 8192] (main.rs:69-74 takes any), more than 65535 windows in one read (`wid` is u16 in the
    reference as well: consensus.rs:22-33)."""
    UNSUPPORTED = -4
    lens = np.array([40000, 16 * 65536 + 1], np.uint32)
    c = api.HostContext(lens)
    rows = np.array([[0, 40000, 0, 3000, 0, 1, lens[1], 0, 3000]], np.uint32)
    for W in (8193, 15, 0, 1 << 20):
        with pytest.raises(api.HerroError) as e:
            c.create_job([1], rows, [0, 1], [b"3000M"], W)
        assert e.value.code == UNSUPPORTED and "window_size" in str(e.value), (W, str(e.value))
    with pytest.raises(api.HerroError) as e:          # 65537 windows of 16 bp
        c.create_job([1], rows, [0, 1], [b"3000M"], 16)
    assert e.value.code == UNSUPPORTED and "65535" in str(e.value), str(e.value)
    j = c.create_job([1], rows, [0, 1], [b"3000M"], 8192)      # the largest window size is served
    assert j.n_windows == -(-int(lens[1]) // 8192)
    j.close()
    c.close()


def test_two_threads_create_jobs_on_one_context():
    """The context's host pool takes one parallel section at a time: two threads building jobs on the same context get the same
    descriptors as one thread building them in turn."""
    import threading
    sb = synth.generate(48, 3000, 10, seed=77, flank_min=30, flank_max=60, p_partial=0.3)
    lens = (sb.off[1:] - sb.off[:-1]).astype(np.uint32)
    W = 512
    c = api.HostContext(lens)
    want = {}
    for k in range(4):
        j = api.job_from_synth(c, sb, W, range(12 * k, 12 * k + 12))
        want[k] = c.job_arrays(j)
        j.close()
    bad, errs = [], []

    def worker(ks):
        try:
            for _ in range(40):
                for k in ks:
                    j = api.job_from_synth(c, sb, W, range(12 * k, 12 * k + 12))
                    got = c.job_arrays(j)
                    j.close()
                    if got.keys() != want[k].keys() or any(got[n].tobytes() != want[k][n].tobytes() for n in got):
                        bad.append(k)
        except Exception as e:
            errs.append(e)
    th = [threading.Thread(target=worker, args=(ks,)) for ks in ((0, 1), (2, 3))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs and not bad
