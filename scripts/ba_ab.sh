#!/bin/bash
# A/B of environment switches on the BA leg in ONE box (wall clock, no per-class timers): scripts/ba_ab.sh "VAR=1" "OTHER=1" ...
mkdir -p gpurun_out
run() {
  env $1 timeout -k 10 300 python bench.py --steps ${STEPS:-40} --warmup 3 --no-cpu-baseline --no-matching --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.1f it/s  %.4f ms/step  one-shot %.2f ms (setup %.2f)' % ('$1', d['value'], d['ms_per_step'], d['ba_one_shot']['ms'], d['ba_one_shot']['setup_ms']))"
}
for rep in 1 2; do
  run "MSFM_DUMMY=0" || exit 1
  for v in "$@"; do run "$v" || exit 1; done
done
