"""TEST INFRASTRUCTURE ONLY — restatement of the reference's PAF ingest for the parity tests of
herro_paf_parse / herro_oec_read (product: herro_amd/csrc/ingest.cpp).  Never imported by the product.

Follows /root/reference/src/overlaps.rs:117-202 (`parse_paf`) and :292-323 (`read_batches`) line by line,
`bytes_to_u32` haec_io.rs:175-183.  Parity status: unpinned (the reference holds no test or fixture for
this function); behaviour is taken from reading the code."""
from __future__ import annotations


class ReferencePanic(Exception):
    pass


def bytes_to_u32(b: bytes) -> int:  # haec_io.rs:175-183 (release build: u32 arithmetic wraps)
    acc = 0
    for d in b:
        if not (48 <= d <= 57):
            raise ReferencePanic("Character is not a valid digit")
        acc = (acc * 10 + (d - 48)) & 0xFFFFFFFF
    return acc


def parse_paf(text: bytes, names: list[bytes], core: set[bytes] | None = None):
    """-> (targets in order of first appearance, {tid: [(qid,qlen,qstart,qend,strand,tid,tlen,tstart,tend,cigar)]})"""
    name_to_id = {}
    for i, n in enumerate(names):  # HashMap::collect: a repeated key keeps the last value
        name_to_id[n] = i
    processed = set()
    order, groups = [], {}
    pos = 0
    while pos < len(text):  # read_until(b'\n')
        e = text.find(b"\n", pos)
        end = len(text) if e < 0 else e + 1
        line = text[pos:end]
        pos = end
        data = line[:-1].split(b"\t")  # buffer[..len - 1]: the last byte goes, newline or not   (:135)
        it = iter(data)

        def nxt():
            try:
                return next(it)
            except StopIteration:
                raise ReferencePanic("called `Option::unwrap()` on a `None` value")
        qid = name_to_id.get(nxt())
        if qid is None:
            continue
        qlen, qstart, qend = bytes_to_u32(nxt()), bytes_to_u32(nxt()), bytes_to_u32(nxt())
        s = nxt()
        if len(s) == 0:
            raise ReferencePanic("index out of bounds")
        if s[0:1] == b"+":
            strand = 0
        elif s[0:1] == b"-":
            strand = 1
        else:
            raise ReferencePanic("Invalid strand character.")
        tstr = nxt()
        if core is not None and tstr not in core:
            continue
        tid = name_to_id.get(tstr)
        if tid is None:
            continue
        tlen, tstart, tend = bytes_to_u32(nxt()), bytes_to_u32(nxt()), bytes_to_u32(nxt())
        rest = list(it)
        if not rest:
            raise ReferencePanic("called `Option::unwrap()` on a `None` value")  # data.last().unwrap()
        last = rest[-1]
        if len(last) < 5:
            raise ReferencePanic("range start index 5 out of range")
        cigar = last[5:]
        if tid == qid:
            continue
        if (qid, tid) in processed:
            continue
        processed.add((qid, tid))
        if tid not in groups:
            groups[tid] = []
            order.append(tid)
        groups[tid].append((qid, qlen, qstart, qend, strand, tid, tlen, tstart, tend, cigar))
    return order, groups


def read_batch(decompressed: bytes, names: list[bytes], core: set[bytes] | None = None):
    """overlaps.rs:304-322 on the already-decompressed stream: '<n>\\n', n id lines, PAF."""
    e = decompressed.find(b"\n")
    end = len(decompressed) if e < 0 else e + 1
    n = 0
    for d in decompressed[:end][:-1]:
        n = (n * 10 + (d - 48)) & 0xFFFFFFFF
    pos = end
    for _ in range(n):
        if pos >= len(decompressed):
            break
        e = decompressed.find(b"\n", pos)
        pos = len(decompressed) if e < 0 else e + 1
    return parse_paf(decompressed[pos:], names, core)
