#!/bin/sh
# Does the offset of the columns modulo the channel interleave matter?  sh tests/tools/stagger_sweep.sh
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
for st in 0 256 1024 4096 16384 65536 69632; do
  COFACTOR_BENCH_STAGGER=$st timeout -k 10 200 python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-calibration > $R/gpurun_out/gs.log 2>&1 || { echo "stagger=$st FAILED"; tail -3 $R/gpurun_out/gs.log; continue; }
  python3 -c "
import json
d=json.loads(open('$R/gpurun_out/gs.log').read().strip().splitlines()[-1])
r=d['roofline']
print('stagger=$st', '%.3g rows/s' % d['value'], 'kernel %.2f ms' % r['avg_kernel_ms'], '%.0f GB/s' % r['achieved'])"
done
