#!/bin/bash
# usage: scripts/rocprof_top.sh <outdir> [bench args...]   -> prints the kernel table of a rocprofv3 kernel-trace run
out=$1; shift
mkdir -p gpurun_out/$out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out -- python3 bench.py --steps 10 --warmup 3 --cpu-frames 0 --no-profile "$@" > gpurun_out/$out/run.log 2>&1
f=$(find gpurun_out/$out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv,sys,re
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    n=re.sub(r"\(anonymous namespace\)::|void |HIP_vector_type<float, 4u>|lsa::","",r["Name"]).split("(")[0]
    print("%-38s calls %5s avg %8.1f us min %7.1f max %8.1f  %5s%%"%(n[:38],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3,r["Percentage"]))
PY
tail -1 gpurun_out/$out/run.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('fps %.1f'%d['value'], d['stage_ms_per_frame'])
except Exception as e: print('no json', e)"
