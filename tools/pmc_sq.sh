#!/bin/bash
# The two SQ counter passes of tools/pmc.sh only (instruction mix, waits, LDS), for several kernel variants in one call:
#   tools/pmc_sq.sh <tag> <scale> <variant> [<variant> ...]        (run on the GPU box)
tag=$1; scale=$2; shift 2
export TMPDIR=/tmp
for v in "$@"; do
  export PANTAS_MAIN_KERNEL=$v
  d=gpurun_out/pmc_${tag}_v$v
  mkdir -p $d
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $d/sq1 -- python3 bench.py --scale $scale --steps 4 --warmup 1 --no-cpu --no-call --no-e2e > $d/sq1.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $d/sq2 -- python3 bench.py --scale $scale --steps 4 --warmup 1 --no-cpu --no-call --no-e2e > $d/sq2.log 2>&1 || exit 1
  python3 tools/pmc_summary.py $d > $d/summary.txt
  echo "== variant $v"; grep -A18 "augment_short_p_kernel\|augment_short_w_kernel" $d/summary.txt | head -20
  rm -rf $d/sq1 $d/sq2
done
