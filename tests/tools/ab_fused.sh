#!/bin/sh
# A/B of two builds of the library on the SAME box (box-to-box variance of the VALU-bound fused
# kernel is up to 40 %): alternates COFACTOR_LIB between $1 and $2 three times.
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in $A $B; do
    COFACTOR_LIB=$L python bench.py --num-cols 10 --cat-cols 10 --rows 1e8 --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], '%.3e rows/s  kernel %.3f ms' % (d['value'], d['roofline']['avg_kernel_ms']))"
  done
done
