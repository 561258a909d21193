#!/bin/bash
# One GPU call of round 3: parity suite, a bench line and a rocprofv3 kernel table.  usage: scripts/r3_check.sh <tag> [pytest args]
T=${1:-r3}; shift
O=gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q "$@" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --batch-sequences 0 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('fps %.1f'%d['value'], 'causal', d.get('causal_no_lookahead',{}).get('value'), 'resident', d.get('replay_resident',{}).get('value'))
    print({k:round(v,3) for k,v in d['stage_ms_per_frame'].items()})
    print(d.get('roofline'))
except Exception as e: print('no json', e)
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-legs --no-profile > $O/trace_run.log 2>&1; echo "trace rc=$?"
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,re
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    n=re.sub(r"\(anonymous namespace\)::|void |HIP_vector_type<float, 4u>|lsa::","",r["Name"]).split("(")[0]
    print("%-40s calls %5s avg %8.1f us min %7.1f max %8.1f  %5s%%"%(n[:40],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3,r["Percentage"]))
PY
find $O -name "*kernel_trace.csv" -delete
