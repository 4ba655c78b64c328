export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 900 python -m pytest -m gpu -q -x -rf tests/test_recon_gpu.py tests/test_boundary_gpu.py tests/test_cli_gpu.py "tests/test_dit_gpu.py::test_sharded_forward_replays_from_a_hipgraph_bit_identically" tests/test_fullsize_gpu.py::test_full_size_vae_decode_matches_oracle tests/test_fullsize_gpu.py::test_config3_21_view_production_width_properties > $O/r4c_quick.log 2>&1; tail -15 $O/r4c_quick.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/scene_tr -o s -- python $GRAFT_REPO_ROOT/tools/scene_trace.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/scene_trace.py --report /tmp/scene_tr 70 > $O/r4c_scene_trace.txt 2>&1; head -75 $O/r4c_scene_trace.txt | cut -c1-220
rm -rf /tmp/scene_tr
cd $GRAFT_REPO_ROOT
OUT=$O/golden timeout 2200 bash tests/golden/make_fullsize_oracle.sh > $O/r4c_oracle_gen.log 2>&1; tail -30 $O/r4c_oracle_gen.log | cut -c1-600; ls -la $O/golden
