#!/bin/bash
# SQ counters of the tokeniser kernels (run on the GPU box): tools/pmc_tok.sh <tag> <scale>
tag=$1; scale=$2
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_${tag}
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}/$name -- python3 bench.py --scale $scale --steps 2 --warmup 1 --no-cpu --no-call --no-e2e > gpurun_out/pmc_${tag}/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python3 tools/pmc_summary.py gpurun_out/pmc_${tag} > gpurun_out/pmc_${tag}/summary.txt
rm -rf gpurun_out/pmc_${tag}/sq1 gpurun_out/pmc_${tag}/sq2
