"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/pmc_traffic.json.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-matching
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-matching
  python3 scripts/pmc_to_json.py /tmp/pmc_f /tmp/pmc_w profiles/pmc_traffic.json profiles/r01_pmc

bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
coalesced reads at 64 bytes (MI355X_MICROARCH.md, HBM section); 16-byte gathers are NOT under-counted, so for the gather
kernels (k_pairs) the corrected figure is an upper bound and the raw one a lower bound - both are stored."""
import csv
import glob
import json
import shutil
import sys

CLASSES = {  # bench.py kernel class -> substrings of the kernel names it launches
    "chol_panel_mfma": ["k_chain", "k_panel_v2"],
    "ba_schur_pairs": ["k_pairs<"],          # (not devsetup::k_pairs_of_points, the list builders of msfm_ba_create)
    "ba_point": ["k_point("],                # (not devsetup::k_point_keys / k_point_lengths)
    "ba_backsub": ["k_backsub"],
    "ba_ftf": ["k_ftf"],                     # (round 4: launched by itself only without the fold tables)
    "ba_sums": ["k_sums"],                   # per-camera sums + pair-list residue + zero fill of the reduced system in one launch
}


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return f, per


def main():
    fdir, wdir, out = sys.argv[1:4]
    prefix = sys.argv[4] if len(sys.argv) > 4 else None
    ff, fetch = load(fdir, "FETCH_SIZE")
    wf, write = load(wdir, "WRITE_SIZE")
    if prefix:
        shutil.copy(ff, prefix + "_fetch_size.csv")
        shutil.copy(wf, prefix + "_write_size.csv")
    res, raw = {}, {}
    src = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 3 --warmup 1 --no-cpu-baseline "
           "--no-matching`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts wide reads at 1/2, MI355X_MICROARCH.md "
           "HBM section); gather kernels are not under-counted: see bytes_per_launch_uncorrected")
    for cls, pats in CLASSES.items():
        names = [k for k in fetch if any(p in k for p in pats) and "devsetup::" not in k]
        if not names:
            continue
        # one launch of the class = one launch of each member kernel (k_pairs has three instantiations per assembly)
        f_sum = sum(sum(fetch[k]) / len(fetch[k]) for k in names)
        w_sum = sum(sum(write[k]) / len(write[k]) for k in names if k in write)
        res[cls] = dict(bytes_per_launch=(2 * f_sum + w_sum) * 1024, bytes_per_launch_uncorrected=(f_sum + w_sum) * 1024,
                        kernel=" + ".join(sorted(n.split("(")[0] for n in names)), source=src)
        for k in names:
            raw[k.split("(")[0]] = dict(FETCH_SIZE=sum(fetch[k]) / len(fetch[k]), WRITE_SIZE=sum(write.get(k, [0])) / max(1, len(write.get(k, [0]))),
                                        launches=len(fetch[k]))
    res["_raw_counters_per_launch_KB"] = raw
    # which kernels the counters belong to: bench.py attaches them to a live timing only when its own hash agrees
    import hashlib, os
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "metricsfm_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    res["_kernel_source_hash"] = h.hexdigest()[:16]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v["bytes_per_launch"] for k, v in res.items() if not k.startswith("_")}))


if __name__ == "__main__":
    main()
