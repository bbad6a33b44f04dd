#!/bin/sh
# usage: sweep_wgs.sh "4 6 8" [extra bench args]
W="$1"; shift
for w in $W; do
  COFACTOR_GRAM_WGS_PER_CU=$w timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python tests/tools/bench_line.py "wgs/CU=$w"
done
