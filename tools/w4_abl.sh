#!/bin/bash
# time tile 14 (gemm_w4_kernel) of the shipped library and of every gpurun_abl/libv3a_*.so (development ablations) on a few shapes
SH="${SH:-8192x1536x8960r;8192x1536x1536r;8192x1536x1536;8192x1536x192;8192x8960x1536}"
echo "== shipped: tiles 6, 14"; python tools/gemm_sweep.py 6,14 "$SH"
for f in gpurun_abl/libv3a_*.so; do echo "== $f"; V3A_LIB=$f python tools/gemm_sweep.py 14 "$SH"; done
