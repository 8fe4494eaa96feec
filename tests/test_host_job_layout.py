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
            tiles.extend((len(win) - 1, r0) for r0 in range(0, lub, 256))
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
