#!/bin/sh
# gram_dma_kernel at 20_0 / 1e9 rows: waves x ring depth x workgroups per CU.  sh tests/tools/gram_dma_sweep.sh
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
for cfg in "4 2 3" "8 2 1" "8 2 2" "8 2 3" "8 3 2" "8 4 1"; do
  set -- $cfg
  COFACTOR_GRAM_DMA_WAVES=$1 COFACTOR_GRAM_DMA_RING=$2 COFACTOR_GRAM_DMA_WGS=$3 timeout -k 10 200 python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-calibration > $R/gpurun_out/gs.log 2>&1 || { echo "waves=$1 ring=$2 wgs=$3 FAILED"; tail -3 $R/gpurun_out/gs.log; continue; }
  python3 -c "
import json
d=json.loads(open('$R/gpurun_out/gs.log').read().strip().splitlines()[-1])
r=d['roofline']
print('waves=$1 ring=$2 wgs/cu=$3', '%.3g rows/s' % d['value'], 'kernel %.2f ms' % r['avg_kernel_ms'], '%.0f GB/s' % r['achieved'], d.get('check'))"
done
