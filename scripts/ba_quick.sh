#!/bin/bash
# BA leg only, with the per-class kernel timers: prints it/s, ms per step and the ten largest kernel classes
mkdir -p gpurun_out
MSFM_VERBOSE=1 MSFM_PROFILE=1 timeout -k 10 300 python bench.py --steps ${STEPS:-30} --warmup 3 --no-cpu-baseline --no-matching --no-extras > gpurun_out/b_quick.json 2> gpurun_out/b_quick.err || { tail -5 gpurun_out/b_quick.err; exit 1; }
grep -i "fold tables" gpurun_out/b_quick.err | head -1
python - <<PY
import json
d=json.loads(open("gpurun_out/b_quick.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
k={r["kernel"]: r["avg_launch_us"] / 1e3 * (1 if r["launches"] < 200 else r["launches"] / d["steps"]) for r in d.get("ba_kernels", [])}
print({a:round(b,4) for a,b in sorted(k.items(), key=lambda kv:-kv[1])[:11]})
PY
