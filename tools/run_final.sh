export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O/golden
# the DiT full-depth digest with the current contract oracle (ctx_vo): ~2.5 minutes of host time
V3A_LIVE_ORACLE=1 V3A_WRITE_ORACLE=1 V3A_ORACLE_OUT=$O/golden timeout 900 python -m pytest -m gpu -q -rP -p no:cacheprovider tests/test_dit_gpu.py::test_full_depth_production_size_forward_matches_oracle > $O/final_dit_digest.log 2>&1; tail -4 $O/final_dit_digest.log | cut -c1-400
cp $O/golden/oracle_dit_full_depth_30_blocks_N4096.safetensors tests/golden/
bash tools/collect_profiles.sh
