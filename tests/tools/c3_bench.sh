#!/bin/sh
# C3 (sum_to_triple_10_10, 1e8 rows, 16 keys) under each one-pass kernel.  sh tests/tools/c3_bench.sh <tag> [prefs]
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
TAG=$1
PREFS=${2:-"1 2 3"}
for p in $PREFS; do
  COFACTOR_FUSED=$p timeout -k 10 200 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-calibration --total-rows 1e8 --num-cols 10 --cat-cols 10 --keys 16 > $R/gpurun_out/${TAG}_c3_pref$p.log 2>&1 || { echo "pref $p FAILED"; tail -5 $R/gpurun_out/${TAG}_c3_pref$p.log; continue; }
  python3 -c "
import json
d=json.loads(open('$R/gpurun_out/${TAG}_c3_pref$p.log').read().strip().splitlines()[-1])
r=d['roofline']
print('pref $p', '%.3g rows/s' % d['value'], '%.2f ms/step' % d['ms_per_step'], r['kernel'], '%.3f ms x %d' % (r['avg_kernel_ms'], r['launches']), 'frac %.3f' % r['frac'], 'err %.2g' % d['check']['max_rel_err_vs_torch_fp64'])"
done
