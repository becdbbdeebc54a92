#!/bin/bash
# Address translation and vector-cache latency counters of the bench's kernels (run on the GPU box): tools/pmc_tlb.sh <tag> <scale>
tag=$1; scale=${2:-s4}
export TMPDIR=/tmp
d=gpurun_out/pmc_tlb_${tag}
mkdir -p $d
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $d/$name -- python3 bench.py --scale $scale --steps 3 --warmup 1 --no-cpu --no-call --no-e2e > $d/$name.log 2>&1
}
run tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
run lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
run stall TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_PENDING_STALL_CYCLES_sum
python3 tools/pmc_summary.py $d > $d/summary.txt
grep -A14 "augment_short_p_kernel" $d/summary.txt | cut -c1-100
rm -rf $d/tlb $d/lat $d/stall
