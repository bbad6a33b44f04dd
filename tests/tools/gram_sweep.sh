#!/bin/sh
# rows/s and HBM fraction of gram_kernel<n> over 1e9 rows for a few column counts
for n in "$@"; do
  python bench.py --num-cols $n --rows 1e9 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['metric'], '%.3e rows/s' % d['value'], 'kernel %.2f ms  %.0f GB/s  frac %.3f' % (r['avg_kernel_ms'], r['achieved'], r['frac']))"
done
