"""herro_fastx_read throughput on a synthetic plain FASTQ (four-line records of 5-35 kb) against HERRO_FASTX_THREADS: the C call alone
(what a host pays), best of three calls in this process.  usage: python tools/fastxrate.py [GB] [threads ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from herro_amd import io as hio  # noqa: E402

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
threads = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
path = "/tmp/fastxrate.fastq"
rng = np.random.default_rng(1)
base = np.frombuffer(b"ACGT", np.uint8)
with open(path, "wb") as f:
    for i in range(int(gb * 1e9 / 2 / 20000)):
        ln = int(rng.integers(5000, 35000))
        f.write(b"@read%d len=%d\n" % (i, ln))
        f.write(base[rng.integers(0, 4, ln)].tobytes())
        f.write(b"\n+\n")
        f.write(rng.integers(33, 90, ln).astype(np.uint8).tobytes())
        f.write(b"\n")
os.sync()                       # the file was just written: let the writeback finish before timing reads of it
size = os.path.getsize(path)
L = hio._lib()
ref = hio.read_fastx(path)
for t in threads:
    os.environ["HERRO_FASTX_THREADS"] = str(t)
    best = 1e9
    for _ in range(3):
        err = C.create_string_buffer(256)
        t0 = time.perf_counter()
        h = L.herro_fastx_read(path.encode(), 0, None, 0, err, 256)
        best = min(best, time.perf_counter() - t0)
        assert h, err.value
        n = L.herro_reads_count(h)
        L.herro_reads_free(h)
    r = hio.read_fastx(path)
    assert n == len(ref.ids) and r.ids == ref.ids and np.array_equal(r.off, ref.off) and np.array_equal(r.seq, ref.seq) and np.array_equal(r.qual, ref.qual)
    print(f"threads {t:2d}: {size / 1e9:.2f} GB, {n} reads in {best:.3f} s -> {size / best / 1e9:.2f} GB/s")
os.remove(path)
