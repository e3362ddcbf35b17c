#!/bin/bash
# A/B of hexahedron-mesh GLS (216^3, all-Dirichlet boundary and a Neumann plane) over library variants, in one GPU session:
#   bash tools/ab_hex8.sh [name ...]     (names of tools/_bin/lib_<name>.so; "base" = the in-tree library; NIN_* switches pass through)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  if [ "$v" = base ]; then unset NINPOL_AMD_LIB; else export NINPOL_AMD_LIB=$R/tools/_bin/lib_$v.so; fi
  echo "== $v"; timeout -k 10 200 python tools/time_neumann.py 216 2>&1 | grep "gls ms" | cut -c1-48
done
