#!/bin/sh
# rows/s of 10_10 at several keys per column, with and without the matrix-core sums pass.  sh tests/tools/bench_keys.sh
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
for k in 24 32 48 64; do
  for off in 0 1; do
    COFACTOR_NO_SUMS_MFMA=$off timeout -k 10 200 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-calibration --total-rows 5e7 --num-cols 10 --cat-cols 10 --keys $k 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('keys $k no_mfma=$off', '%.3g rows/s' % d['value'], '%.2f ms/step' % d['ms_per_step'])"
  done
done
