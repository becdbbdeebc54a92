#!/bin/bash
# Needs a library built with the timing-only instances: make -C pantas_amd/csrc clean && make -C pantas_amd/csrc -j8 EXPERIMENTS=1
# (the shipped build refuses PANTAS_ABLATE); build the plain library again afterwards.
# timing-only ablations of the main kernel (results are wrong by construction): tools/ablate.sh <scale>
scale=${1:-s3}
for a in ${ABLATIONS:-0 1 2 4 8 12 16 20 31}; do
  out=$(PANTAS_ABLATE=$a python bench.py --scale $scale --steps 6 --warmup 2 --full-line --no-cpu --no-call --no-e2e 2>&1 | tail -1)
  echo "ablate=$a $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("main_ms=%.3f" % d["kernel_ms"]["augment_main"])')"
done
