#!/bin/bash
# `pantas call` / `pantas augment` at S4 with more formatting threads than granted cores (the makers of pieces block while the one writer writes)
python tools/e2e_scale.py --scale s4 --keep --heartbeat gpurun_out/hb.txt --out gpurun_out/wt_first.json > gpurun_out/wt_first.log 2>&1 || exit 1
pre=$(ls /dev/shm/pantas_e2e_s4_*.gfa | grep -v aug | head -1); pre=${pre%.gfa}
for t in 16 24 32 16 24 32; do
  PANTAS_WRITER_THREADS=$t PANTAS_TIMING=1 python -m pantas_amd.call --rca 3 --novel $pre.aug.gfa $pre.gtf > /dev/shm/wt_out.csv 2> gpurun_out/wt_call_$t.err || exit 1
  echo "call threads $t: $(grep -E 'tables built|events written|writer thread' gpurun_out/wt_call_$t.err | sed 's/.*cli call: //;s/\[timing\] //' | tr '\n' ';')"
  rm -f /dev/shm/wt_out.csv
done
for t in 16 32; do
  PANTAS_WRITER_THREADS=$t PANTAS_TIMING=1 python -m pantas_amd.augment $pre.gaf $pre.gfa > /dev/shm/wt_out.gfa 2> gpurun_out/wt_aug_$t.err || exit 1
  echo "augment threads $t: $(grep -E 'counters on host|GFA written|writer thread' gpurun_out/wt_aug_$t.err | sed 's/.*cli augment: //;s/\[timing\] //' | tr '\n' ';')"
  rm -f /dev/shm/wt_out.gfa
done
rm -f /dev/shm/pantas_e2e_s4_*
