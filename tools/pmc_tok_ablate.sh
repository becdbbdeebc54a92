#!/bin/bash
# SQ counters of every launch of parse_once_kernel in tools/tok_ablate.py (run on the GPU box): tools/pmc_tok_ablate.sh <tag> <scale> <stops>
tag=$1; scale=$2; stops=$3
export TMPDIR=/tmp
d=gpurun_out/pmc_${tag}
mkdir -p $d
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY --output-format csv -d $d/raw -- python3 tools/tok_ablate.py --scale $scale --stops $stops > $d/run.log 2>&1
python3 - "$d" "$stops" <<'PY'
import csv, glob, sys, collections
d, stops = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for path in glob.glob(f"{d}/raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "parse_once" not in (r.get("Kernel_Name") or ""):
            continue
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
labels = ["warm", "real"] + list(stops)
with open(f"{d}/summary.txt", "w") as f:
    for lab, i in zip(labels, ids):
        c = rows[i]
        f.write(f"stop {lab}: " + "  ".join(f"{k.replace('SQ_', '')}={v:.4g}" for k, v in sorted(c.items())) + "\n")
print(open(f"{d}/summary.txt").read())
PY
rm -rf $d/raw
