#!/bin/bash
# One measurement set for profiles/ (run on the GPU box from the repo root): tools/profile_round.sh <tag>
#   default bench line; rocprofv3 kernel stats of the same command; HBM counters in separate --pmc passes.
tag=${1:-run}
out=gpurun_out/prof_${tag}
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
cp gpurun_out/bench_details_n1.json $out/bench_default_details.json      # the whole object behind the short line
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu --no-e2e > $out/bench_under_rocprof.json 2> $out/stats.err || exit 1
# (with the call kernels since round 5: SURVEY 8-d asks for their bytes "for completeness")
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-e2e > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $out/pmc/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-e2e > $out/pmc_write.log 2>&1 || exit 1
python3 tools/pmc_summary.py $out/pmc > $out/pmc_summary.txt
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/stats $out/pmc      # the raw traces are large; the summaries stay
ls -la $out
