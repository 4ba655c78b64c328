#!/bin/bash
# Development A/B builds: compile ONE csrc file with extra -D flags and link it with the other in-tree objects into gpurun_abl/libv3a_<name>.so
# (same ABI; pick it up with V3A_LIB=...).  usage: tools/variant_build.sh <name> <file.hip> <flags...>
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p gpurun_abl /tmp/v3a_var
stem=$(basename $src .hip)
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result "$@" -c vist3a_amd/csrc/$src -o /tmp/v3a_var/${stem}_$name.o
objs=$(ls vist3a_amd/csrc/build/*.o | grep -v "/${stem}.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o gpurun_abl/libv3a_$name.so $objs /tmp/v3a_var/${stem}_$name.o
echo gpurun_abl/libv3a_$name.so
