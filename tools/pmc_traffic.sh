#!/bin/bash
# HBM traffic of the dominant kernel per launch: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), no tracing domains.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_traffic_$c -o p -- python bench.py --steps 1 --warmup 0 --denoise-steps 4 --no-cpu-baseline > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob("$R/gpurun_out/pmc_traffic_%s/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if r["Counter_Name"] == c:
                acc[k] += float(r["Counter_Value"]); n[k] += 1
    for k in acc:
        if "gemm_nt_kernel<256, 192, 4, 2, 64, 2, false, 0, 1>" in k or "gemm_nt_kernel<256, 192, 4, 2, 32, 2, false, 0, 2>" in k or "attn_fwd_kernel3<128" in k:
            out.setdefault(k[:90], {})[c] = dict(avg_per_launch=acc[k] / n[k], launches=n[k])
print(json.dumps(out, indent=1))
PY
