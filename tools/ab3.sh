#!/bin/bash
# A/B of main-kernel variants in one session: tools/ab3.sh <scale> <variant> [<variant> ...]
scale=${1:-s3}; shift
for v in "$@"; do
  out=$(PANTAS_MAIN_KERNEL=$v python bench.py --scale $scale --steps 8 --warmup 2 --full-line --no-cpu --no-call 2>&1 | grep '^{"metric' | tail -1)
  echo "variant=$v $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["kernel_ms"]; print("avg=%.3f min=%.3f max=%.3f value=%.3g" % (k["augment_main"], k["augment_main_min"], k["augment_main_max"], d["value"]) + " short=%.3f" % k.get("augment_main_short_walk", 0))')"
done
