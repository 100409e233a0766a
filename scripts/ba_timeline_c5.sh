#!/bin/bash
# One LM iteration of config 3 as a timeline (kernel trace): kernel, duration, gap to the previous kernel's end
set -o pipefail
R=$PWD; O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --config 5 --window --steps 6 --warmup 2 --no-cpu-baseline --no-matching --no-extras > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tl/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete iteration: from the last-but-one k_point (mode 0 launches are the long ones) to the last
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_point") and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 5000]
a, b = idx[-3], idx[-2]
t_prev = int(rows[a - 1]["End_Timestamp"])
tot_k = tot_g = 0
out = []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append("%-34s %8.2f us   gap %6.2f us" % (r["Kernel_Name"][:34], (e - s) / 1e3, (s - t_prev) / 1e3))
    tot_k += e - s; tot_g += max(0, s - t_prev); t_prev = e
open("gpurun_out/tl/iteration.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:80]))
print("kernels %.1f us, gaps %.1f us, launches %d" % (tot_k / 1e3, tot_g / 1e3, b - a))
PY
