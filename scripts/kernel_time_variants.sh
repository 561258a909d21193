#!/bin/bash
# rocprofv3 average duration of the named kernels for the tree library and every _variants/lib_*.so (one box)
# usage: scripts/kernel_time_variants.sh "k_label k_compact" [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
K=$1; shift
for v in lidarslam_amd/liblidarslam_amd.so _variants/lib_*.so; do
  O=gpurun_out/ktv/$(basename $v .so); rm -rf $O; mkdir -p $O
  LSA_LIB=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-extra-legs --no-profile "$@" > $O/run.log 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$K" "$v" <<'PY'
import csv,sys,re
want=sys.argv[2].split(); out=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r"\(anonymous namespace\)::|void |lsa::","",r["Name"]).split("(")[0]
    if any(n.startswith(w) for w in want): out.append("%s %.1f us x%s"%(n[:28],float(r["AverageNs"])/1e3,r["Calls"]))
print(sys.argv[3], " | ".join(out))
PY
  find $O -name "*kernel_trace.csv" -delete
done
