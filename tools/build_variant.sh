#!/bin/bash
# A variant of libninpol_amd.so with ONE translation unit recompiled under extra flags (stamps builds, A/B candidates):
#   bash tools/build_variant.sh <name> <unit.hip> <extra hipcc flags...>     -> tools/_bin/lib_<name>.so
# The other objects are the in-tree ones (python -m ninpol_amd.build first).  Use with NINPOL_AMD_LIB=$PWD/tools/_bin/lib_<name>.so
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; unit=$2; shift 2
mkdir -p $R/tools/_bin
base=$(basename $unit .hip)
extra=""
case $base in
  kernels_idw_ls|grid_device) extra="-ffp-contract=off";;
  kernels_gls_hex8mf) extra="-mllvm -amdgpu-atomic-optimizer-strategy=None";;
esac
hipcc --offload-arch=gfx950 $extra "$@" -O3 -fPIC -std=c++17 -c $R/ninpol_amd/csrc/$base.hip -o $R/tools/_bin/${base}_$name.o
others=$(ls $R/ninpol_amd/csrc/_obj/*.o | grep -v "/$base.o")
g++ -shared -o $R/tools/_bin/lib_$name.so $R/tools/_bin/${base}_$name.o $others -L /opt/rocm/lib -lamdhip64 -lgomp -Wl,-rpath,/opt/rocm/lib
echo $R/tools/_bin/lib_$name.so
