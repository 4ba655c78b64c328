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
        if "gemm_pp_kernel" in k or "attn_fwd_plain_kernel" in k or "conv_halo_kernel" in k or "xattn_probs_kernel" in k:
            key = k.replace("void (anonymous namespace)::", "").replace("((anonymous namespace)::GemmP)", "").replace("((anonymous namespace)::AttnP)", "").replace("((anonymous namespace)::HaloP)", "").replace("((anonymous namespace)::XaP)", "").replace("(anonymous namespace)::", "")
            out.setdefault(key, {})[c + "_KiB_avg"] = acc[k] / n[k]
            out[key]["launches"] = n[k]
for k, v in out.items():   # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md section HBM): doubled
    v["hbm_bytes_per_launch"] = int((2 * v.get("FETCH_SIZE_KiB_avg", 0) + v.get("WRITE_SIZE_KiB_avg", 0)) * 1024)
open("$R/gpurun_out/pmc_traffic.json", "w").write(json.dumps(dict(kernels=out), indent=1))
print(json.dumps(out, indent=1))
PY
