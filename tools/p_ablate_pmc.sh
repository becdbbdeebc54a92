#!/bin/bash
# Where the position-parallel kernel's instructions are: its timing-only instances (124 kept() is nothing, 125 kept() and later() nothing,
# 126 nothing behind the kept mask) beside the real one, instruction counts from one --pmc pass each.  Builds the library with
# EXPERIMENTS=1 on the GPU box first (the shipped build has no such instances) -- results of these runs are wrong by construction.
#   bash tools/p_ablate_pmc.sh [scale]
scale=${1:-s4}
export TMPDIR=/tmp
make -C pantas_amd/csrc -j16 EXPERIMENTS=1 OUT=/tmp/libpantas_exp.so OBJDIR=/tmp/obj_exp > /tmp/exp_build.log 2>&1 || { tail -20 /tmp/exp_build.log; exit 1; }
cp pantas_amd/libpantas_gpu.so /tmp/libpantas_plain.so
cp /tmp/libpantas_exp.so pantas_amd/libpantas_gpu.so
mkdir -p gpurun_out/pabl
for v in 120 124 125 126; do
  PANTAS_MAIN_KERNEL=$v rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pabl/v$v -- python3 bench.py --scale $scale --steps 3 --warmup 1 --no-cpu --no-call --no-e2e --full-line > gpurun_out/pabl/v$v.log 2>&1 || { tail -5 gpurun_out/pabl/v$v.log; }
  python3 tools/pmc_summary.py gpurun_out/pabl/v$v > gpurun_out/pabl/v$v.txt
  echo "== variant $v"; grep -A9 "augment_short_p_kernel" gpurun_out/pabl/v$v.txt | cut -c1-70
  python3 - <<P
import json
try:
    d=json.loads(open("gpurun_out/pabl/v$v.log").read().strip().splitlines()[-1]); print("   short-walk ms under the profiler:", d["kernel_ms"]["augment_main_short_walk"])
except Exception as e: print("   (no bench line:", e, ")")
P
  rm -rf gpurun_out/pabl/v$v
done
cp /tmp/libpantas_plain.so pantas_amd/libpantas_gpu.so
