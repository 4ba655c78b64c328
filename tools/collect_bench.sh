#!/bin/bash
# The bench / rocprofv3 / PMC part of tools/collect_profiles.sh alone (PMC passes first, so that the bench line's `roofline.traffic` comes from
# THIS build's counters): gpurun --timeout 1800 -- 'bash tools/collect_bench.sh', then copy gpurun_out/prof_* into profiles/rN/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O $R/profiles/r5
bash tools/pmc_traffic.sh > /dev/null 2>&1; cp $O/pmc_traffic.json $O/prof_pmc_traffic.json; cp $O/pmc_traffic.json $R/profiles/r5/pmc_traffic.json; rm -rf $O/pmc_traffic_FETCH_SIZE $O/pmc_traffic_WRITE_SIZE
bash tools/pmc_mfma.sh > /dev/null 2>&1; cp $O/pmc_mfma.json $O/prof_pmc_mfma.json; rm -rf $O/pmc_mfma
python bench.py 2>/dev/null | tail -1 > $O/prof_bench_default_run.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/prof_bench_under_rocprof.json
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/prof_bench_kernel_stats.csv
python $R/tools/trace_shapes.py /tmp/prof_stats 45 > $O/prof_bench_kernel_shapes.txt
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --output-format csv -d /tmp/scene_tr -o s -- python $R/tools/scene_trace.py > /dev/null 2>&1
python $R/tools/scene_trace.py --report /tmp/scene_tr 60 > $O/prof_scene_trace.txt 2>&1
rm -rf /tmp/scene_tr
cd $R
for v in 1 0 1 0; do V3A_CTX_VO=$v python tools/dit_time.py 2>/dev/null | tail -1; done > $O/prof_dit_time_ctx_vo_on_off.jsonl
for v in 0 1 0 1; do V3A_FUSED_QKV=$v python tools/dit_time.py 2>/dev/null | tail -1; done > $O/prof_dit_time_fused_qkv_off_on.jsonl
python tools/qkv_time.py 2>/dev/null | tail -1 > $O/prof_qkv_time.jsonl
python tools/tile_time.py 2>/dev/null | grep "^{" > $O/prof_tile_time.jsonl
