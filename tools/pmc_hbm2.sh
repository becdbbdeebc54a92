#!/bin/bash
# Second set of memory-side counters (run on the GPU box): read requests by size class and by destination.
#     tools/pmc_hbm2.sh <tag> <scale>
tag=$1; scale=${2:-s4}
export TMPDIR=/tmp
out=gpurun_out/hbm_${tag}
mkdir -p $out
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --scale $scale --steps 3 --warmup 1 --no-cpu --no-call --no-e2e > $out/$name.log 2>&1 || echo "pass $name failed" >> $out/failed.txt
}
run rdsz TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run rddram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_GMI_32B_sum
run wrdram TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum
python3 tools/pmc_summary.py $out > $out/summary2.txt
for d in rdsz rddram wrdram; do rm -rf $out/$d; done
grep -A10 "augment_short_p_kernel\|count_newlines_kernel\|planar_nodes_kernel" $out/summary2.txt
