#!/bin/bash
# The call kernels one after the other under rocprofv3 (run on the GPU box from the repo root): tools/call_kernel_stats.sh <scale> <tag>
#   gpurun_out/ks_<tag>_stats.csv = the kernel stats, gpurun_out/ks_<tag>_walks.txt = every walk_once_kernel dispatch (ns)
export TMPDIR=/tmp
cd /tmp && PANTAS_CALL_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ks_$2 -- python3 $GRAFT_REPO_ROOT/bench.py --scale $1 --no-cpu --no-e2e --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/ks_$2.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/ks_$2 -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/ks_$2_stats.csv
t=$(find gpurun_out/ks_$2 -name "*kernel_trace.csv" | head -1)
python3 - "$t" > gpurun_out/ks_$2_walks.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'walk_once' in r['Kernel_Name']:
        print(int(r['End_Timestamp']) - int(r['Start_Timestamp']), r.get('Grid_Size', r.get('Grid_Size_X', '')))
PY
rm -rf gpurun_out/ks_$2
