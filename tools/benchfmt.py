import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("windows/s", round(d["value"]), "ms/step", round(d["ms_per_step"],3), d["stage_ms_per_step"], "roof", d["roofline"]["achieved"], d["roofline"]["frac"])
for k,v in d["kernels"].items(): print(f"  {k:16s} {v['avg_us']:8.1f} us x{v['calls']}")
