python -m pytest tests/test_gpu_tokenize.py -x -q 2>&1 | tail -2
for a in 0 24 20 28; do
  PANTAS_TOK_ALIGN_MIN=$a python bench.py --scale s4 --steps 5 --warmup 2 --full-line --no-cpu --no-call --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['tokenizer']
print('align_min $a', 'parse_s', round(t['measure_and_scans_s'],5), 'bucket_s', round(t['bucket_s'],5), 'index', round(t['line_index_s'],5), 'incl_bucket', d['reads_per_s_incl_bucket'], 'main', d['kernel_ms']['augment_main'])"
done
