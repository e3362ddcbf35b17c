#!/bin/bash
# PMC passes over the GLS kernels of one mesh family (default tet40): instruction mix and wait cycles.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CASE=${1:-tet40}
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_INSTS_SMEM"; do
  i=$((i+1))
  (cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_$CASE/p$i --output-format csv -- python3 tools/time_methods.py $CASE > $R/gpurun_out/pmc_$CASE/log$i.txt 2>&1) || { echo "pass $i failed"; tail -5 $R/gpurun_out/pmc_$CASE/log$i.txt; }
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmc_$CASE/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "gls" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        print(k, {a: f"{b:.4g}" for a, b in d.items()})
PY
