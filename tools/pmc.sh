#!/bin/bash
# usage: tools/pmc.sh <tag> <kprof args...>   -> gpurun_out/pmc_<tag>_{a,b}.csv (two separate --pmc passes, no tracing domains)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --output-format csv -d $R/gpurun_out/pmc_${tag}_a -o p -- python tools/kprof.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU \
  --output-format csv -d $R/gpurun_out/pmc_${tag}_b -o p -- python tools/kprof.py "$@" > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for s in "ab":
    for f in glob.glob("$R/gpurun_out/pmc_${tag}_%s/*counter_collection.csv" % s):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "gemm_nt" in k or "attn_fwd" in k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            print(k)
            for c, v in sorted(d.items()):
                print(f"   {c:28s} {v / n[(k, c)]:.4g}  (avg per dispatch, {n[(k,c)]} dispatches)")
PY
