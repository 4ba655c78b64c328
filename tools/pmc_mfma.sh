#!/bin/bash
# MFMA-pipe busy fraction per kernel: one --pmc pass (no tracing domains) around a short bench run.
# util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs)   (gfx94x MfmaUtil formula; MI355X_MICROARCH.md §counters)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_mfma -o p -- \
  python bench.py --steps 1 --warmup 0 --denoise-steps 4 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$R/gpurun_out/pmc_mfma/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
out = {}
for k, d in acc.items():
    if "GRBM_GUI_ACTIVE" not in d or d["GRBM_GUI_ACTIVE"] == 0: continue
    util = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (d["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    out[k[:110]] = dict(dispatches=n[(k, "GRBM_GUI_ACTIVE")], gui_active_cycles_total=d["GRBM_GUI_ACTIVE"], mfma_busy_cycles_total=d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), mfma_util=round(util, 4))
ours = {k: v for k, v in out.items() if "anonymous namespace" in k and "at::native" not in k}   # every kernel of csrc/, not only the top rows
top = dict(sorted(ours.items(), key=lambda kv: -kv[1]["gui_active_cycles_total"])[:16])
open("$R/gpurun_out/pmc_mfma.json", "w").write(json.dumps(top, indent=1))
print(json.dumps(top, indent=1))
PY
