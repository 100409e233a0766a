#!/bin/bash
# BA bench at config 3 for several k_ftf chunk lengths (experiments).
for c in ${CHUNKS:-1024 512 256 128 64}; do
  MSFM_FTF_CHUNK=$c python3 bench.py --steps 20 --warmup 3 --no-extras --no-matching --no-cpu-baseline 2> gpurun_out/fc$c.err > gpurun_out/fc$c.json || exit 1
  python3 - $c <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/fc{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("ftf chunk", sys.argv[1], round(d["value"],1), "it/s", [(k["kernel"], round(k["ms_per_step"],4)) for k in d["ba_kernels"] if k["kernel"] in ("ba_ftf","ba_point","ba_linearize")], flush=True)
P
done
