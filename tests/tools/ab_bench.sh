#!/bin/sh
# Same-box A/B of two library builds on any bench configuration:
#   sh tests/tools/ab_bench.sh <libA> <libB> [bench args...]
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in $A $B; do
    COFACTOR_LIB=$L python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], '%.3e rows/s  %s %.3f ms' % (d['value'], d['roofline']['kernel'], d['roofline']['avg_kernel_ms']))"
  done
done
