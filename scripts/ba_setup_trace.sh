#!/bin/bash
# Developer tool: kernel trace of the small fresh solves of scripts/ba_setup_laps.py - how many launches a set-up is, how long
# they run and how long the device idles between them.   gpurun -- 'bash scripts/ba_setup_trace.sh'
set -o pipefail
R=$PWD; O=$R/gpurun_out/setup_tl; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/scripts/ba_setup_laps.py > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/setup_tl/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last solve: from the last k_fill/k_scan-like burst... take the last 400 launches and cut at gaps > 2 ms
cut = 0
for i in range(len(rows) - 1, 0, -1):
    if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 1_500_000:
        cut = i
        break
last = rows[cut:]
t0, t1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
print("last solve: %d launches, %.3f ms from first start to last end, device busy %.3f ms" % (len(last), (t1 - t0) / 1e6, busy / 1e6))
# split set-up / iterations at the first k_point
ip = next(i for i, r in enumerate(last) if r["Kernel_Name"].startswith("k_point"))
su = last[:ip]
print("set-up part: %d launches, %.3f ms wall, busy %.3f ms" % (len(su), (int(su[-1]["End_Timestamp"]) - t0) / 1e6,
      sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in su) / 1e6))
gaps = sorted(((int(su[i + 1]["Start_Timestamp"]) - int(su[i]["End_Timestamp"])) / 1e3, su[i]["Kernel_Name"][:40], su[i + 1]["Kernel_Name"][:40]) for i in range(len(su) - 1))
print("largest gaps (us): ", gaps[-12:])
from collections import Counter
c = Counter(r["Kernel_Name"].split("(")[0][-45:] for r in su)
print(c.most_common(25))
PY
