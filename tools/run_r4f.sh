export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest -m gpu -q -x -rf tests/test_kernels_gpu.py -k "xattn or batched or row_sumsq or production_shapes" > $O/r4f_k.log 2>&1; tail -8 $O/r4f_k.log | cut -c1-300
timeout 1200 python -m pytest -m gpu -q -rf tests/test_dit_gpu.py tests/test_boundary_gpu.py tests/test_vae_gpu.py tests/test_recon_gpu.py tests/test_fullsize_gpu.py::test_full_size_vae_decode_matches_oracle tests/test_fullsize_gpu.py::test_full_size_reconstruction_matches_oracle tests/test_fullsize_gpu.py::test_config3_21_view_reconstruction_layout_matches_oracle > $O/r4f_dit.log 2>&1; tail -25 $O/r4f_dit.log | cut -c1-400
for v in 1 0 1 0; do V3A_CTX_VO=$v timeout 300 python tools/dit_time.py 2>&1 | tail -1; done > $O/r4f_ab.log; cat $O/r4f_ab.log
timeout 300 python tools/vae_time.py 2>&1 | tail -2
timeout 300 python tools/recon_time.py 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/r4f_bench.json; cat $O/r4f_bench.json | cut -c1-1500
