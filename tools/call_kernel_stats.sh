#!/bin/bash
# usage: ks.sh <scale> <tag>
export TMPDIR=/tmp
cd /tmp && PANTAS_CALL_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ks_$2 -- python3 $GRAFT_REPO_ROOT/bench.py --scale $1 --no-cpu --no-e2e --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/ks_$2.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/ks_$2 -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/ks_$2_stats.csv; rm -rf gpurun_out/ks_$2
