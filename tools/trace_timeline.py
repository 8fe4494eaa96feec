"""Prints the last N kernel dispatches of a rocprofv3 --kernel-trace CSV as a timeline (start / end in us relative to the first of them).
usage: python tools/trace_timeline.py <kernel_trace.csv> [N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"]
    name = name[:name.find("(")] if "(" in name else name
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:9.1f}  q{r.get("Queue_Id", "?"):>3}  grid {r.get("Grid_Size", r.get("Grid_Size_X", "?")):>8}  {name[-90:]}')
