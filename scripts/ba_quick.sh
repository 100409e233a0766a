#!/bin/bash
# quick BA bench at config 3: rate, final cost (bit-identity checks between variants) and the per-kernel times
python3 bench.py --steps ${STEPS:-20} --warmup 3 --no-extras --no-matching --no-cpu-baseline 2> gpurun_out/q.err > gpurun_out/q.json || exit 1
python3 - <<'P'
import json
d=json.loads(open("gpurun_out/q.json").read().strip().splitlines()[-1])
print(round(d["value"],1), "it/s", round(d["ms_per_step"],4), "ms", d["ba_cost"]["final"].hex(), "one-shot", round(d["ba_one_shot"]["iterations_per_s"],1))
print([(k["kernel"], round(k["ms_per_step"],4)) for k in d["ba_kernels"]], flush=True)
P
