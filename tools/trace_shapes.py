"""Per-shape kernel durations from a rocprofv3 --kernel-trace CSV (groups by kernel name + grid size)."""
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    acc[(n[:48], r["Grid_Size_X"], r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    v2 = sorted(v)[len(v) // 10: len(v) - len(v) // 10] or v
    # avg_us is the TRUE mean (what a roofline needs: total time / launches); trim_us drops the top and bottom 10 % (cold first launches)
    print(f"{k[0]:50s} grid {k[1]:>8s} n {len(v):5d} avg_us {sum(v) / len(v):9.1f} trim_us {sum(v2) / len(v2):9.1f} share {100 * sum(v) / tot:5.1f}%")
