#!/bin/sh
# rows/s and HBM fraction of the dense kernel over 1e9 rows for a few column counts, gram_kernel and
# its LDS-DMA variant.  sh tests/tools/gram_sweep.sh 1 4 13 16 20
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
for n in "$@"; do for dma in 0 1; do
  COFACTOR_GRAM_DMA=$dma timeout -k 10 200 python3 $R/bench.py --num-cols $n --rows 1e9 --steps 5 --warmup 2 --no-cpu-baseline --no-calibration > $R/gpurun_out/gs.log 2>&1 || { echo "n=$n dma=$dma FAILED"; tail -2 $R/gpurun_out/gs.log; continue; }
  tail -1 $R/gpurun_out/gs.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('n=$n dma=$dma', '%.3e rows/s' % d['value'], 'kernel %.2f ms  %.0f GB/s  frac %.3f' % (r['avg_kernel_ms'], r['achieved'], r['frac']))"
done; done
