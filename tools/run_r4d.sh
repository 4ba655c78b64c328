export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 600 python -m pytest -m gpu -q -x -rf tests/test_kernels_gpu.py -k "xattn or batched" > $O/r4d_k.log 2>&1; tail -8 $O/r4d_k.log
timeout 900 python -m pytest -m gpu -q -rf tests/test_dit_gpu.py tests/test_boundary_gpu.py tests/test_fullsize_gpu.py::test_config3_21_view_dit_forward_matches_oracle > $O/r4d_dit.log 2>&1; tail -25 $O/r4d_dit.log | cut -c1-400
for v in 1 0 1 0; do V3A_CTX_VO=$v timeout 300 python tools/dit_time.py 2>&1 | tail -1; done > $O/r4d_ab.log; cat $O/r4d_ab.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/scene_tr -o s -- python $GRAFT_REPO_ROOT/tools/scene_trace.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/scene_trace.py --report /tmp/scene_tr 30 > $O/r4d_scene_trace.txt 2>&1; head -60 $O/r4d_scene_trace.txt | cut -c1-200
rm -rf /tmp/scene_tr
