#!/bin/bash
# A/B of the augment main-kernel variants on one box: tools/ab.sh <scale> [lines]
scale=${1:-s3}
for cfg in "0 0" "3 0" "4 0" "5 0"; do
  set -- $cfg
  out=$(PANTAS_MAIN_KERNEL=$1 PANTAS_NO_BUCKET=$([ "$2" = 1 ] && echo 1) python bench.py --scale $scale --steps 10 --warmup 2 --no-cpu --no-call 2>&1 | tail -1)
  echo "variant=$1 nobucket=$2 $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("main_ms=%.3f rare_ms=%.3f reads/s=%.3g frac=%.4f tok_s=%.2f" % (d["kernel_ms"]["augment_main"], d["kernel_ms"]["augment_rare"], d["value"], d["roofline"]["frac"], d["host"]["gaf_tokenize_s"]))')"
done
