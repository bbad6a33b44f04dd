#!/bin/sh
# gram_kernel at 20_0 / 1e9 rows: tile order x workgroups per CU.  sh tests/tools/gram_grid_sweep.sh
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
for ch in 0 1; do for w in 2 3 4 6 8 16; do
  COFACTOR_GRAM_CHUNKED=$ch COFACTOR_GRAM_WGS_PER_CU=$w timeout -k 10 200 python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-calibration > $R/gpurun_out/gs.log 2>&1 || { echo "chunked=$ch wgs=$w FAILED"; tail -3 $R/gpurun_out/gs.log; continue; }
  python3 -c "
import json
d=json.loads(open('$R/gpurun_out/gs.log').read().strip().splitlines()[-1])
r=d['roofline']
print('chunked=$ch wgs/cu=$w', '%.3g rows/s' % d['value'], 'kernel %.2f ms' % r['avg_kernel_ms'], '%.0f GB/s' % r['achieved'])"
done; done
