#!/bin/bash
# One gpurun call that regenerates everything under profiles/rN (summaries only: the per-dispatch traces are deleted on the box).
#   gpurun --timeout 3000 -- 'bash tools/collect_profiles.sh'   then copy gpurun_out/prof_* into profiles/rN/
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
V3A_FULL_SIZE=${V3A_FULL_SIZE:-0} python -m pytest tests -q -m gpu -rP --durations=25 > $O/prof_gputest_stdout.log 2>&1   # incl. the production-size teacher-forced runs
tail -3 $O/prof_gputest_stdout.log
cp $O/parity.json $O/prof_parity.json
python bench.py 2>/dev/null | tail -1 > $O/prof_bench_default_run.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/prof_bench_under_rocprof.json
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/prof_bench_kernel_stats.csv
python $R/tools/trace_shapes.py /tmp/prof_stats 45 > $O/prof_bench_kernel_shapes.txt
rm -rf /tmp/prof_stats
# exactly one scene's worth of kernels, no bench extras (host-glue launches listed separately)
rocprofv3 --kernel-trace --output-format csv -d /tmp/scene_tr -o s -- python $R/tools/scene_trace.py > /dev/null 2>&1
python $R/tools/scene_trace.py --report /tmp/scene_tr 60 > $O/prof_scene_trace.txt 2>&1
rm -rf /tmp/scene_tr
cd $R
bash tools/pmc_traffic.sh > /dev/null 2>&1; cp $O/pmc_traffic.json $O/prof_pmc_traffic.json; rm -rf $O/pmc_traffic_FETCH_SIZE $O/pmc_traffic_WRITE_SIZE
bash tools/pmc_mfma.sh > /dev/null 2>&1; cp $O/pmc_mfma.json $O/prof_pmc_mfma.json; rm -rf $O/pmc_mfma
python tools/dit14b_time.py 2>/dev/null | tail -4 > $O/prof_dit14b.jsonl
python tools/sp_rank_time.py 2>/dev/null | tail -8 > $O/prof_sp_rank_time_1_3b.jsonl
python tools/sp_rank_time.py 14b 2>/dev/null | tail -8 > $O/prof_sp_rank_time_14b.jsonl
python tools/sp_rank_time.py 14b fp8 2>/dev/null | tail -8 > $O/prof_sp_rank_time_14b_fp8.jsonl
python tools/sp_rank_time.py views21 2>/dev/null | tail -8 > $O/prof_sp_rank_time_views21.jsonl
python tools/gemm_sweep.py 6,8,7 0,1,2,3,4,8 2>/dev/null | grep "^{" > $O/prof_gemm_sweep.jsonl
python tools/gemm_fp8_time.py 2>/dev/null | grep "^{" > $O/prof_gemm_fp8.jsonl
python tools/attn_time.py 2x12x4096 2x12x6144 2>/dev/null | grep '^{' > $O/prof_attn_time.jsonl
python tools/xprobs_time.py 2>/dev/null | grep '^{' > $O/prof_xprobs_time.jsonl
python tools/norm_time.py 2>/dev/null | grep '^{' > $O/prof_norm_time.jsonl            # the row passes beside the copy yardstick of the same box
python tools/voxel_time.py 2>/dev/null | grep '^{' > $O/prof_voxel_time.jsonl          # compact sort key (round 6)
python tools/gemm_sweep.py 6,14 0,1,2,3,8 2>/dev/null | grep "^{" > $O/prof_gemm_sweep_w4.jsonl   # ping-pong tile vs the one-wave-per-SIMD tile
python tools/conv_sweep.py 2>/dev/null | tail -2 > $O/prof_conv_sweep.txt
python tools/vae_time.py 2>/dev/null | tail -2 > $O/prof_vae_time.jsonl
python tools/recon_time.py 2>/dev/null | tail -2 > $O/prof_recon_time.jsonl
python tools/dpt_layers.py 13 f32 2>/dev/null | grep "^{" > $O/prof_dpt_layers_f32.jsonl
python tools/dpt_layers.py 13 bf16 2>/dev/null | grep "^{" | tail -1 > $O/prof_dpt_layers_bf16_total.jsonl
python tools/conv_split_sweep.py 2>/dev/null | grep "^{" > $O/prof_conv_split_sweep.jsonl
for v in 1 0 1 0; do V3A_CTX_VO=$v python tools/dit_time.py 2>/dev/null | tail -1; done > $O/prof_dit_time_ctx_vo_on_off.jsonl
ls -la $O | head -50
