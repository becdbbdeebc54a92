#!/bin/bash
# S5 `pantas augment` wall with the GFA loader's device memory parked (default) and freed at once, alternating, on one set of input files.
python tools/e2e_scale.py --scale s5 --blocks 4 --keep --heartbeat gpurun_out/hb.txt --out gpurun_out/s5_ab_first.json > gpurun_out/s5_ab_first.log 2>&1 || exit 1
pre=$(ls /dev/shm/pantas_e2e_s5_*.gfa | grep -v aug | head -1); pre=${pre%.gfa}
echo "files: $pre"
for i in 1 2; do
  for mode in park free; do
    if [ $mode = free ]; then export PANTAS_FREE_DEVICE_AT_ONCE=1; else unset PANTAS_FREE_DEVICE_AT_ONCE; fi
    PANTAS_TIMING=1 python -m pantas_amd.augment $pre.gaf $pre.gfa > /dev/shm/s5_ab_out.gfa 2> gpurun_out/s5_ab_${mode}_$i.err || exit 1
    echo "$mode $i: $(grep -E 'graph loaded|counters on host|GFA written' gpurun_out/s5_ab_${mode}_$i.err | sed 's/.*cli augment: //' | tr '\n' ';')"
    rm -f /dev/shm/s5_ab_out.gfa
  done
done
rm -f /dev/shm/pantas_e2e_s5_*
