#!/bin/bash
# round 6: what is the GPU doing during the end_to_end leg?  kernel + memory-copy trace of one run, busy fractions between the first and the last CIGAR scan; and the leg
# against the number of hardware queues.   usage: gpurun --timeout 1200 -- bash tools/r6_e2e_trace.sh tag
tag=${1:-r6t}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
q="--no-cpu-baseline --self-check 0 --strong-windows 0 --repeats 0 --steps 64 --warmup 32 --sustained 0 --sensitivity 0 --long-run-steps 0"
for hq in ${QUEUES:-4 8 16}; do
  GPU_MAX_HW_QUEUES=$hq timeout 200 python bench.py $q 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); e=d['end_to_end']; print('hw queues $hq value', round(d['value']), 'e2e', round(e['windows_per_s']), round(e['windows_per_s']/d['value'],3))" | tee -a $out/queues.txt
done
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/tr -o t -- python bench.py $q > $out/bench.json 2> $out/trace.err < /dev/null
kt=$(find $out/tr -name "*kernel_trace.csv" | head -1); mt=$(find $out/tr -name "*memory_copy_trace.csv" | head -1)
python - "$kt" "$mt" <<'PY' | tee $out/busy.txt
import csv, sys, re, collections
kt, mt = sys.argv[1], sys.argv[2]
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))) for r in csv.DictReader(open(kt))]
M = []
try:
    for r in csv.DictReader(open(mt)):
        M.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "?")), int(r.get("Size", 0) or 0)))
except Exception as e:
    print("no copy trace", e)
scans = sorted(s for s, e, n in K if "k_cigar_scan" in n)
# the timed e2e pass = the second half of the scans (the first half warms the feeders)
t0, t1 = scans[len(scans) // 2], scans[-1]
def union(iv):
    iv = sorted((max(s, t0), min(e, t1)) for s, e in iv if e > t0 and s < t1)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot
span = t1 - t0
print(f"e2e window {span/1e6:.1f} ms, {sum(1 for s in scans if s >= t0)} jobs")
print(f"any kernel running: {100*union([(s,e) for s,e,n in K])/span:.1f} %")
by = collections.defaultdict(int)
for s, e, n in K:
    if e > t0 and s < t1: by[n[:40]] += min(e, t1) - max(s, t0)
for n, v in sorted(by.items(), key=lambda kv: -kv[1])[:14]: print(f"   {n:42s} {100*v/span:6.1f} % of the window (sum of durations)")
for d in sorted(set(m[2] for m in M)):
    sel = [(s, e) for s, e, dd, _ in M if dd == d]
    byt = sum(b for s, e, dd, b in M if dd == d and e > t0 and s < t1)
    print(f"copies {d}: busy {100*union(sel)/span:.1f} %, {byt/1e6:.0f} MB -> {byt/span:.1f} GB/s over the window")
PY
rm -rf $out/tr
