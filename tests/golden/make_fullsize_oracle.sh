#!/bin/bash
# Regenerates the full-size oracle digests (tests/golden/oracle_*.safetensors) with the LIVE CPU oracle.  The three tests run the HIP path
# too, so this needs the GPU box (its 128 host threads also make the oracle ~20 minutes instead of hours):
#   gpurun --timeout 2400 -- 'OUT=gpurun_out/golden bash tests/golden/make_fullsize_oracle.sh'   then copy gpurun_out/golden/* here.
cd "$(dirname "$0")/../.."
V3A_LIVE_ORACLE=1 V3A_WRITE_ORACLE=1 V3A_ORACLE_OUT=${OUT:-tests/golden} python -m pytest -m gpu -q -rP -p no:cacheprovider \
  tests/test_fullsize_gpu.py::test_full_size_reconstruction_matches_oracle \
  tests/test_fullsize_gpu.py::test_config3_21_view_reconstruction_layout_matches_oracle \
  tests/test_dit_gpu.py::test_full_depth_production_size_forward_matches_oracle
