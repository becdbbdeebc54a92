#!/bin/bash
# Memory-side request counters of the L2 (TCC_EA0_*), per kernel, in separate --pmc passes (run on the GPU box):
#     tools/pmc_hbm.sh <tag> <scale>
# What FETCH_SIZE / WRITE_SIZE are made of: read requests by size (all, the 32-byte ones), write requests (all, the 64-byte ones),
# atomics, and the L2's own request / hit counts.  tools/hbm_bytes.py turns the summary into exact bytes per launch, calibrated
# on a kernel of the SAME run whose bytes are known (count_newlines_kernel streams the GAF text once).
tag=$1; scale=${2:-s4}
export TMPDIR=/tmp
out=gpurun_out/hbm_${tag}
mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_REQ[A-Za-z_]*\|TCC_HIT[A-Za-z_]*\|TCC_MISS[A-Za-z_]*\|TCC_BUBBLE[A-Za-z_]*" | sort -u > $out/tcc_counters_available.txt
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --scale $scale --steps 3 --warmup 1 --no-cpu --no-call --no-e2e > $out/$name.log 2>&1 || echo "pass $name failed" >> $out/failed.txt
}
run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run at TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum
run l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 tools/pmc_summary.py $out > $out/summary.txt
for d in rd wr at l2 fetch write; do rm -rf $out/$d; done
cat $out/summary.txt | grep -A12 "augment_short_p_kernel\|count_newlines_kernel\|planar_nodes_kernel\|augment_rare_agg"
