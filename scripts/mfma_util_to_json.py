"""Summarises the SQ counter passes of scripts/mfma_util.sh into profiles/r05_mfma_util.json.

Per kernel (mean over its launches in each pass):
  cycles          = GRBM_GUI_ACTIVE / 8           (rocprofv3 sums the 8 XCDs; MI355X_MICROARCH.md "DVFS give-back")
  mfma_busy       = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)      (busy cycles are shader cycles per SIMD, summed)
  valu_busy       = 4 x SQ_ACTIVE_INST_VALU / (1024 x cycles)             (SQ_ACTIVE_INST_* count quad-cycles)
  valu_per_mfma   = SQ_INSTS_VALU / SQ_INSTS_MFMA  - 1                    (SQ_INSTS_VALU includes the MFMA instructions)
"""
import csv
import glob
import json
import os
import sys

KERNELS = {"knn_i8": "k_knn2_i8", "knn_f16": "k_knn2_f16", "ba": "k_chain"}


def main():
    root, out = sys.argv[1], sys.argv[2]
    res = {}
    for wl, pat in KERNELS.items():
        vals = {}
        for d in sorted(glob.glob(os.path.join(root, wl + ".*"))):
            if not os.path.isdir(d):
                continue
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                acc = {}
                for r in csv.DictReader(open(f)):
                    if pat not in r["Kernel_Name"]:
                        continue
                    a = acc.setdefault(r["Counter_Name"], [0, 0.0])
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
                for k, (n, v) in acc.items():
                    vals[k] = v / n
                    vals["_launches_" + k] = n
        if not vals:
            continue
        cyc = vals.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        e = dict(kernel=pat, counters_mean_per_launch={k: v for k, v in vals.items() if not k.startswith("_")},
                 launches=int(vals.get("_launches_GRBM_GUI_ACTIVE", 0)), cycles_per_launch=cyc)
        if cyc > 0:
            if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
                e["mfma_busy"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
            if "SQ_ACTIVE_INST_VALU" in vals:
                e["valu_busy"] = 4.0 * vals["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc)
        if vals.get("SQ_INSTS_MFMA"):
            e["valu_per_mfma"] = vals["SQ_INSTS_VALU"] / vals["SQ_INSTS_MFMA"] - 1.0
        res[wl] = e
    res["_source"] = ("rocprofv3 --pmc passes of scripts/mfma_util.sh (one counter group per pass, program directly after --): "
                      "scripts/knn_only.py 48 (2256 pairs, int8), scripts/knn_float_only.py (552 pairs, f16), "
                      "bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-matching --no-extras (k_chain: the persistent panel chain)")
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "metricsfm_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    res["_kernel_source_hash"] = h.hexdigest()[:16]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: {m: v.get(m) for m in ("mfma_busy", "valu_busy", "valu_per_mfma")} for k, v in res.items() if not k.startswith("_")}))


if __name__ == "__main__":
    main()
