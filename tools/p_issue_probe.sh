#!/bin/bash
# Is the position-parallel kernel bound by instruction issue?  Its timing-only instances with 32 more vector (127) / scalar (128)
# instructions per tile against the real one (120), timed by tools/ab_variants.py on one resident workload (results identical: the
# extra instructions feed nothing).  Builds the library with EXPERIMENTS=1 on the GPU box first.   bash tools/p_issue_probe.sh [scale]
scale=${1:-s4}
make -C pantas_amd/csrc -j16 EXPERIMENTS=1 OUT=/tmp/libpantas_exp.so OBJDIR=/tmp/obj_exp > /tmp/exp_build.log 2>&1 || { tail -20 /tmp/exp_build.log; exit 1; }
cp pantas_amd/libpantas_gpu.so /tmp/libpantas_plain.so
cp /tmp/libpantas_exp.so pantas_amd/libpantas_gpu.so
python3 tools/ab_variants.py --scale $scale --variants ${VARIANTS:-120,127,128,120,127,128} --steps 8 ${NOCHECK:+--no-check} | cut -c1-200
cp /tmp/libpantas_plain.so pantas_amd/libpantas_gpu.so
