#!/bin/bash
# SQ counters of k_point with and without the fold tables (counters only; one pass per group; program directly after --)
set -o pipefail
R=$PWD; O=$R/gpurun_out/ppmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
GROUPS_=("SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY")
for mode in fold nofold; do
  g=0
  for set in "${GROUPS_[@]}"; do
    if [ $mode = nofold ]; then export MSFM_NO_FOLD=1; else unset MSFM_NO_FOLD; fi
    timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $O/$mode.$g -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-matching --no-extras > $O/$mode.$g.log 2>&1 || echo "pass $mode.$g failed"
    g=$((g + 1))
  done
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for mode in ("fold", "nofold"):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f"gpurun_out/ppmc/{mode}.*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("k_point"):
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(mode, {k: round(v / n[k]) for k, v in sorted(tot.items())}, "launches", dict(n).get("SQ_INSTS_LDS"))
PY
