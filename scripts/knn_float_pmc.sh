#!/bin/bash
# Hardware counters of the certified f16 matcher: gpurun -- 'bash scripts/knn_float_pmc.sh'
set -o pipefail
R=$PWD; O=$R/gpurun_out/knnfpmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  n=$(echo $set | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/$n -- python3 $R/scripts/knn_float_only.py > $O/$n.log 2>&1 || echo "pass $n failed"
  f=$(find $O/$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: [0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_knn2_f16' not in r['Kernel_Name']: continue
    a=acc[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for k,(n,v) in acc.items(): print("%-28s launches %d  mean %.4g" % (k,n,v/n))
PY
done
