#!/bin/bash
# A/B timing of one library under different environment switches, in ONE GPU session:
#   bash tools/ab_env.sh "NIN_MFW_LANE_COLUMNS=1" "" [-- mesh names]
envs=(); meshes=(tet40 wedge60 mixed)
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; meshes=("$@"); break; fi
  envs+=("$1"); shift
done
for i in 1 2; do
  for e in "${envs[@]}"; do
    env $e NIN_METHODS=gls timeout -k 10 300 python tools/time_methods.py "${meshes[@]}" 2>/dev/null | grep "gls:" | sed "s|^|[$e]  |"
  done
done
