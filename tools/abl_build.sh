#!/bin/bash
# Experiment builds of libvist3a_hip.so with -DV3A_PW_ABL=<n> in attention.hip (same ABI; select with V3A_LIB=<path>):
#   tools/abl_build.sh 1 2 4   ->  gpurun_abl/libvist3a_hip_abl1.so ...   (directory travels to the GPU box; it is git-ignored)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_abl
python -m vist3a_amd.build > /dev/null
for n in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DV3A_ATTN_EXPERIMENTAL -DV3A_PW_ABL=$n -c vist3a_amd/csrc/attention.hip -o gpurun_abl/attention_abl$n.o
  objs=$(ls vist3a_amd/csrc/build/*.o | grep -v "/attention.o")
  hipcc -shared -fPIC --offload-arch=gfx950 -o gpurun_abl/libvist3a_hip_abl$n.so $objs gpurun_abl/attention_abl$n.o
  rm gpurun_abl/attention_abl$n.o
  echo built gpurun_abl/libvist3a_hip_abl$n.so
done
