#!/bin/sh
# fused2 modes without pair accumulators: workgroups per CU.  sh tests/tools/bench_f2wgs.sh
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
run() {
  name=$1; shift
  timeout -k 10 200 python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-calibration "$@" > $R/gpurun_out/f2w.log 2>&1 || { echo "$name FAILED"; tail -3 $R/gpurun_out/f2w.log; return 0; }
  python3 -c "
import json
d=json.loads(open('$R/gpurun_out/f2w.log').read().strip().splitlines()[-1])
r=d['roofline']
print('$name', '%.3g rows/s' % d['value'], '%.2f ms/step' % d['ms_per_step'], r['kernel'], '%.2f ms x %d' % (r['avg_kernel_ms'], r['launches']))"
}
for w in 1 2 3; do
  COFACTOR_F2_WGS=$w run nb_10_10_wgs$w --total-rows 1e8 --num-cols 10 --cat-cols 10 --nb
  COFACTOR_F2_WGS=$w run 20_20_wgs$w --total-rows 5e7 --num-cols 20 --cat-cols 20
  COFACTOR_F2_WGS=$w run 20_10_wgs$w --total-rows 5e7 --num-cols 20 --cat-cols 10
done
