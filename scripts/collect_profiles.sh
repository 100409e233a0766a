#!/bin/bash
# Collects what profiles/ holds, on the GPU box:  gpurun --timeout 1100 -- 'bash scripts/collect_profiles.sh'
# (run from the repo root; every step writes under gpurun_out/prof, copy the summaries into profiles/ afterwards).
set -e -o pipefail
R=$PWD
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O   # (the local copy under gpurun_out/ keeps older run directories: delete it before a new collection)
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/kt.log 2>&1
echo "kernel trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-matching --no-extras > $O/pmc_fetch.log 2>&1
echo "pmc fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-matching --no-extras > $O/pmc_write.log 2>&1
echo "pmc write done"
cd $R
python3 scripts/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json $O/pmc > $O/pmc_summary.txt 2>&1 || echo "pmc post-processing failed"
# the bench line comes after the counters so that it can attach them (same kernel sources: the hash inside matches)
# matrix-pipe utilisation of the three MFMA kernels (SQ counter passes; writes gpurun_out/mfma/r05_mfma_util.json)
bash scripts/mfma_util.sh > $O/mfma_util.log 2>&1 || echo "mfma_util failed"
cp gpurun_out/mfma/r05_mfma_util.json profiles/r05_mfma_util.json || true
echo "mfma util done"
cp $O/pmc_traffic.json profiles/pmc_traffic.json
python3 bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"; tail -c 300 $O/bench.json
python3 bench.py --config 5 --window --steps 20 --warmup 3 > $O/bench_c5_window.json 2> $O/bench_c5_window.err
echo "config 5 window done"
python3 scripts/extra_bench.py --c5 > $O/extra.json 2> $O/extra.err
echo "extra done"
find $O -name "*kernel_stats.csv" | head -3
