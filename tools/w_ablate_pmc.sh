#!/bin/bash
# Needs a library built with the timing-only instances: make -C pantas_amd/csrc clean && make -C pantas_amd/csrc -j8 EXPERIMENTS=1
# (the shipped build refuses PANTAS_ABLATE); build the plain library again afterwards.
# Instruction counts and time of the wide-record kernel (augment_short_w_kernel, ABL instance) with parts switched off, one --pmc pass each
# (PANTAS_ABLATE bits: 32 nothing queued, 512 a kept node is one probe and one atomic, 256 a kept node is nothing; 288 = the bare walk).
# Timing only: results are wrong by construction.  Run on the GPU box: bash tools/w_ablate_pmc.sh
export TMPDIR=/tmp
mkdir -p gpurun_out/wabl
for a in 0 32 512 288; do
  PANTAS_ABLATE=$a rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/wabl/a$a -- python3 bench.py --scale s4 --steps 3 --warmup 1 --no-cpu --no-call --no-e2e > gpurun_out/wabl/a$a.log 2>&1 || exit 1
  python3 tools/pmc_summary.py gpurun_out/wabl/a$a | grep -A9 augment_short_w > gpurun_out/wabl/a$a.txt
  echo "== ablate $a"; cat gpurun_out/wabl/a$a.txt; grep -o '"augment_main_short_walk": [0-9.]*' gpurun_out/wabl/a$a.log
done
