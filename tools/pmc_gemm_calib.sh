#!/bin/bash
# Where the dominant GEMM's L2-miss traffic comes from: FETCH_SIZE / WRITE_SIZE (separate --pmc passes) of the ping-pong 256x192 tile on two
# isolated shapes, against the prediction "A once + the whole weight panel B once PER XCD (eight private L2s) + residual".
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_cal_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_cal_$c -o p -- python tools/gemm_sweep.py 6 "8192x1536x1536r;8192x3072x1536;8192x1536x8960r" > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
acc = {c: [] for c in ("FETCH_SIZE", "WRITE_SIZE")}
for c in acc:
    for f in glob.glob("/tmp/pmc_cal_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "gemm_pp_kernel<3, true" in r["Kernel_Name"]:
                acc[c].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
# the three shapes are launched in order, 3 warm-up + 3 x 20 timed launches each: split the dispatch sequence in thirds
out = {}
for c in acc:
    seq = [v for _, v in sorted(acc[c])]
    n = len(seq) // 3
    out[c] = [sum(seq[i * n:(i + 1) * n]) / max(1, n) for i in range(3)]
MB = lambda kib: kib * 1024 / 1e6
shapes = [("8192x1536x1536 + residual", 25.2 + 8 * 4.7 + 25.2, 25.2), ("8192x3072x1536 (512 tiles = two rounds: A is read once per round)", 2 * 25.2 + 8 * 9.4, 50.3), ("8192x1536x8960 + residual", 146.8 + 8 * 27.5 + 25.2, 25.2)]
res = []
for i, (name, pf, pw) in enumerate(shapes):
    res.append(dict(shape=name, fetch_MB_measured_x2=round(2 * MB(out["FETCH_SIZE"][i]), 1), fetch_MB_predicted_A_per_round_plus_8B_plus_res=round(pf, 1),
                    write_MB_measured=round(MB(out["WRITE_SIZE"][i]), 1), write_MB_predicted=pw))
open("$O/pmc_gemm_calib.json", "w").write(json.dumps(res, indent=1))
print(json.dumps(res, indent=1))
PY
