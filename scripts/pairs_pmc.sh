#!/bin/bash
# Cache counters of the BA gather kernels: gpurun -- 'bash scripts/pairs_pmc.sh'
set -o pipefail
R=$PWD; O=$R/gpurun_out/pairspmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum SQ_WAVE_CYCLES" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY"; do
  n=$(echo $set | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/$n -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-matching > $O/$n.log 2>&1 || echo "pass $n failed"
  f=$(find $O/$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: [0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    name = 'k_pairs66' if 'k_pairs<6, 6' in k or 'k_pairs<(int)6, (int)6' in k else 'k_point' if k.startswith('k_point') else 'k_linearize' if 'k_linearize<true>' in k or 'k_linearize<(bool)1>' in k else 'k_backsub' if k.startswith('k_backsub') else 'k_ftf' if k.startswith('k_ftf') else None
    if not name: continue
    a=acc[(name,r['Counter_Name'])]; a[0]+=1; a[1]+=float(r['Counter_Value'])
for (k,c),(n,v) in sorted(acc.items()): print("%-12s %-32s launches %3d  mean %.4g" % (k,c,n,v/n))
PY
done
