#!/bin/bash
# development helper: recompile ONE unit of libninpol_amd.so and relink (python -m ninpol_amd.build recompiles all of them):
#   bash tools/rebuild_unit.sh kernels_gls_mfg.hip [extra hipcc flags]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
U=$1; shift
O=$R/ninpol_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 "$@" -O3 -fPIC -std=c++17 -c $R/ninpol_amd/csrc/$U -o $O/${U%.*}.o
g++ -shared -o $R/ninpol_amd/libninpol_amd.so $O/*.o -L /opt/rocm/lib -lamdhip64 -lgomp -Wl,-rpath,/opt/rocm/lib
ls -la $R/ninpol_amd/libninpol_amd.so
