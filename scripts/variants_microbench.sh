#!/bin/bash
# one box: dump a frame's match inputs with the tree library, then time every _variants/lib_*.so (and the tree) on them
mkdir -p gpurun_out/mb
timeout -k 10 200 python scripts/match_microbench.py dump /tmp/mb.npz > gpurun_out/mb/dump.log 2>&1 || { tail -5 gpurun_out/mb/dump.log; exit 1; }
for round in 1 2; do
  timeout -k 10 100 python scripts/match_microbench.py run /tmp/mb.npz 2>&1 | tail -1
  for v in _variants/lib_*.so; do
    LSA_LIB=$v timeout -k 10 100 python scripts/match_microbench.py run /tmp/mb.npz 2>&1 | tail -1
  done
done | tee gpurun_out/mb/results.txt
