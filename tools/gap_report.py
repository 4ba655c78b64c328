"""Busy / idle split of the LAST 1/parts of a rocprofv3 --kernel-trace CSV (e.g. the last of `parts` identical forwards):
   python tools/gap_report.py DIR parts   -> span, busy, idle, launches, and the largest gaps with the kernels around them."""
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
last = rows[-(len(rows) // parts):]
span = (last[-1][1] - last[0][0]) / 1e3
busy = sum(e - s for s, e, _ in last) / 1e3
gaps = sorted(((last[i + 1][0] - last[i][1]) / 1e3, last[i][2][:50], last[i + 1][2][:50]) for i in range(len(last) - 1))
print(json.dumps(dict(launches=len(last), span_us=round(span, 1), busy_us=round(busy, 1), idle_us=round(span - busy, 1),
                      gaps_over_10us=sum(g[0] > 10 for g in gaps), idle_in_gaps_over_10us=round(sum(g[0] for g in gaps if g[0] > 10), 1))))
for g in gaps[-8:]:
    print(round(g[0], 1), "us between", g[1], "->", g[2])
