#!/usr/bin/env python
"""HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes: TCC has 4
slots, FETCH_SIZE costs 3 and WRITE_SIZE 2).  Units: rocprofv3 reports KiB.  gfx950 correction
(MI355X_MICROARCH.md §HBM): FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads -> doubled in
`hbm_bytes_corrected`; WRITE_SIZE is used as reported (uncalibrated).
usage: pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv GROUP out.json [PRECISION]"""
import collections
import csv
import json
import sys


def avg(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("herro::", "").split("<")[0]
            agg[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


f, w = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
rename = {"k_cols": "cols", "k_win": "win", "k_layout": "layout", "k_tokens": "tokens", "k_supgather": "supgather", "k_rfq": "rf_quals",
          "k_quals": "rf_quals", "k_consensus": "consensus", "k_cigar_scan": "cigar_scan", "k_patch_conv1_s": "patch_conv1", "k_conv_w": "conv_fused",
          "k_gemm_g": "fc_gemm", "k_layers": "layers_fused", "k_gemm_g256": "fc_gemm", "k_add_pe": "add_pe", "k_build_tokens": "build_tokens",
          "k_conv_h": "conv_fused", "k_conv_m": "conv_fused", "k_fc_h": "fc_gemm", "k_fc_r": "fc_gemm", "k_supoff": "supoff", "k_rows": "rows", "k_consensus_p": "consensus", "k_layers_p": "layers_fused", "k_build_tokens_h": "build_tokens"}
out = {"group": int(sys.argv[3]), "windows_per_launch": int(sys.argv[3]) * 128, "precision": int(sys.argv[5]) if len(sys.argv) > 5 else 4,
       "unit": "bytes per launch (per_window: the same divided by windows_per_launch — what bench.py scales to its own launch size)", "kernels": {}}
for k in sorted(set(f) | set(w)):
    fb, wb = f.get(k, 0.0) * 1024, w.get(k, 0.0) * 1024
    rec = {"fetch_bytes_raw": fb, "write_bytes": wb, "hbm_bytes_corrected": 2 * fb + wb,
           "hbm_bytes_corrected_per_window": (2 * fb + wb) / out["windows_per_launch"], "kernel": k}
    name = rename.get(k, k)
    # two kernels can share a bench name (the f16 kernels and the bf16x3 ones the few large windows take): keep the one that moves the bytes
    if name not in out["kernels"] or rec["hbm_bytes_corrected"] > out["kernels"][name]["hbm_bytes_corrected"]:
        out["kernels"][name] = rec
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_corrected"] / 1e6, 1) for k, v in out["kernels"].items()}))
