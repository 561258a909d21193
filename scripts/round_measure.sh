#!/bin/bash
# End-of-round measurement on one MI355X box: bench lines for the three sensor configs, CPU baseline
# sweeps, rocprofv3 kernel-trace summary and the two PMC passes (FETCH_SIZE / WRITE_SIZE separately).
# usage: scripts/round_measure.sh r02   (writes gpurun_out/r02/, then run scripts/collect_profiles.py r02 here)
R=${1:-r03}
set -o pipefail
mkdir -p gpurun_out/$R
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-extra-legs"
python bench.py --steps 60 --warmup 10 > gpurun_out/$R/bench_vls128.json 2> gpurun_out/$R/bench_vls128.err; echo "bench128 rc=$?"
python bench.py --steps 60 --warmup 10 --model 64 --no-extra-legs > gpurun_out/$R/bench_hdl64.json 2>/dev/null; echo "bench64 rc=$?"
python bench.py --steps 60 --warmup 10 --model 16 --no-extra-legs > gpurun_out/$R/bench_vlp16.json 2>/dev/null; echo "bench16 rc=$?"
python bench.py --steps 60 --warmup 10 $Q --no-profile > gpurun_out/$R/bench_vls128_noevents.json 2>/dev/null; echo "bench128 (no events) rc=$?"
python bench.py --steps 60 --warmup 10 $Q --causal > gpurun_out/$R/bench_vls128_causal.json 2>/dev/null; echo "bench128 (causal: no look-ahead) rc=$?"
python bench.py --steps 60 --warmup 10 $Q --resident > gpurun_out/$R/bench_vls128_resident.json 2>/dev/null; echo "bench128 (resident scans) rc=$?"
python bench.py --steps 60 --warmup 10 $Q --param MapsOnDevice=0 > gpurun_out/$R/bench_vls128_hostmaps.json 2>/dev/null; echo "bench128 (host maps) rc=$?"
python scripts/batch_sweep.py > gpurun_out/$R/batch_sweep.jsonl 2>/dev/null; echo "batch sweep rc=$?"
python scripts/cpu_baseline_sweep.py 16 20 > gpurun_out/$R/cpu_sweep_vlp16.log 2>&1; echo "cpu16 rc=$?"
python scripts/cpu_baseline_sweep.py 64 8 > gpurun_out/$R/cpu_sweep_hdl64.log 2>&1; echo "cpu64 rc=$?"
python scripts/cpu_baseline_sweep.py 128 6 > gpurun_out/$R/cpu_sweep_vls128.log 2>&1; echo "cpu128 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/trace -- python3 bench.py --steps 40 --warmup 8 $Q > gpurun_out/$R/trace_run.log 2>&1; echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/pmc_fetch -- python3 bench.py --steps 10 --warmup 4 $Q --no-profile > gpurun_out/$R/pmc_fetch_run.log 2>&1; echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/pmc_write -- python3 bench.py --steps 10 --warmup 4 $Q --no-profile > gpurun_out/$R/pmc_write_run.log 2>&1; echo "pmc write rc=$?"
# keep the merged-back payload small: drop the per-dispatch kernel traces
find gpurun_out/$R -name "*kernel_trace.csv" -delete
ls gpurun_out/$R
