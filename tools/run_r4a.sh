export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "conv" > $O/r4a_conv_tests.log 2>&1; tail -5 $O/r4a_conv_tests.log
timeout 200 python tools/conv_sweep.py > $O/r4a_conv_sweep.log 2>&1; tail -3 $O/r4a_conv_sweep.log
timeout 200 python tools/vae_time.py > $O/r4a_vae_time.log 2>&1; tail -3 $O/r4a_vae_time.log
timeout 400 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>$O/r4a_bench.err | tail -1 > $O/r4a_bench.json; cat $O/r4a_bench.json
timeout 1100 python -m pytest tests -q -m gpu -rP > $O/r4a_gputest.log 2>&1; tail -5 $O/r4a_gputest.log
