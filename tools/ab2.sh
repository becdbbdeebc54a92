#!/bin/bash
scale=${1:-s3}
for cfg in "0 4" "18 8" "9 8" "20 8" "22 8" "18 8"; do
  set -- $cfg
  out=$(PANTAS_MAIN_KERNEL=$1 python bench.py --scale $scale --steps $2 --warmup 2 --no-cpu --no-call 2>&1 | tail -1)
  echo "variant=$1 steps=$2 $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["kernel_ms"]; print("avg=%.3f min=%.3f max=%.3f all=%s" % (k["augment_main"], k["augment_main_min"], k["augment_main_max"], k["augment_main_all"]))')"
done
