#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the match kernel on the microbench inputs, tree library and every _variants/lib_*.so
mkdir -p gpurun_out/pmcmb; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python scripts/match_microbench.py dump /tmp/mb.npz > gpurun_out/pmcmb/dump.log 2>&1 || exit 1
for v in tree _variants/lib_*.so; do
  [ "$v" = tree ] && unset LSA_LIB || export LSA_LIB=$v
  timeout -k 10 100 python scripts/match_microbench.py run /tmp/mb.npz 2>&1 | tail -1
  for c in WRITE_SIZE FETCH_SIZE; do
    d=gpurun_out/pmcmb/$(basename $v .so)_$c; rm -rf $d
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 scripts/match_microbench.py run /tmp/mb.npz 3 > /dev/null 2>&1
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 - "$f" $c <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_search_all' in r['Kernel_Name']:
        k=r['Kernel_Name'].split('k_search_all')[1][:14]; acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
print('   ', sys.argv[2], {k:(v[0], round(v[1]/v[0])) for k,v in acc.items()})
PY
  done
done
find gpurun_out/pmcmb -name "*.csv" -delete
