#!/bin/bash
# A/B timing of the GLS block kernel (tetrahedron, wedge and mixed meshes) for library builds, in ONE GPU session:
#   bash tools/ab_block.sh tools/_bin/lib_A.so tools/_bin/lib_B.so ... [-- mesh names]
libs=(); meshes=(tet40 wedge60 mixed)
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; meshes=("$@"); break; fi
  libs+=("$1"); shift
done
for i in 1 2; do
  for lib in "${libs[@]}"; do
    NIN_METHODS=gls NINPOL_AMD_LIB=$PWD/$lib timeout -k 10 300 python tools/time_methods.py "${meshes[@]}" 2>/dev/null | grep "gls:" | sed "s|^|$lib  |"
  done
done
