#!/bin/bash
# Matrix-pipe / vector-pipe utilisation of the three MFMA kernels (k_knn2_i8, k_knn2_f16, k_panel_v2) from the SQ counters:
#   gpurun --timeout 900 -- 'bash scripts/mfma_util.sh'      then copy gpurun_out/mfma/r05_mfma_util.json into profiles/
# One rocprofv3 --pmc pass per counter group and workload (counters only: no --kernel-trace / --stats beside --pmc), the
# program directly after `--`.  Summarised by scripts/mfma_util_to_json.py.
set -o pipefail
R=$PWD; O=$R/gpurun_out/mfma; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
GROUPS_=("SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES")
run() {   # name, program + args
  local name=$1; shift
  local g=0
  for set in "${GROUPS_[@]}"; do
    timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $O/$name.$g -- "$@" > $O/$name.$g.log 2>&1 || echo "pass $name.$g failed"
    echo "$name group $g done"
    g=$((g + 1))
  done
}
run knn_i8 python3 $R/scripts/knn_only.py 48
run knn_f16 python3 $R/scripts/knn_float_only.py
run ba python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-matching --no-extras
cd $R
python3 scripts/mfma_util_to_json.py $O $O/r05_mfma_util.json
