#!/bin/sh
# gram_kernel (COFACTOR_GRAM_RING=0) against gram_ring_kernel (=1) over 1e9 rows.  sh tests/tools/gram_ring_sweep.sh 1 4 13 16 20
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
for n in "$@"; do for ring in 0 1; do
  COFACTOR_GRAM_RING=$ring timeout -k 10 200 python3 $R/bench.py --num-cols $n --rows 1e9 --steps 6 --warmup 2 --no-cpu-baseline --no-calibration > $R/gpurun_out/gs.log 2>&1 || { echo "n=$n ring=$ring FAILED"; tail -2 $R/gpurun_out/gs.log; continue; }
  tail -1 $R/gpurun_out/gs.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('n=$n ring=$ring', '%.3e rows/s' % d['value'], 'kernel %.2f ms  %.0f GB/s  frac %.3f' % (r['avg_kernel_ms'], r['achieved'], r['frac']), 'err %.1e' % d['check']['max_rel_err_vs_torch_fp64'])"
done; done
