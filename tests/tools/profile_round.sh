#!/bin/sh
# Collects the rocprofv3 evidence kept under profiles/ for one round: kernel-trace stats, separate
# FETCH_SIZE / WRITE_SIZE passes and SQ counters for the bench configurations.  On the GPU box:
#   sh tests/tools/profile_round.sh r02
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
TAG=$1
cd /tmp && export TMPDIR=/tmp
KEY=""; KROWS=""
run() { # name, extra bench args...   (KEY / KROWS: the traffic.json entry this configuration feeds)
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${name}_trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-calibration "$@" > $R/gpurun_out/${TAG}_${name}_trace.log 2>&1 || return 1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_${name}_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calibration --no-check "$@" > /dev/null 2>&1 || return 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_${name}_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calibration --no-check "$@" > /dev/null 2>&1 || return 1
  # condensed on the box (the raw CSVs are large): summaries go to gpurun_out/profiles_<tag>/
  mkdir -p $R/gpurun_out/profiles_${TAG}
  PROFILES_OUT=$R/gpurun_out/profiles_${TAG} python3 $R/profiles/summarize.py ${TAG} $R/gpurun_out/${TAG}_${name}_trace $R/gpurun_out/${TAG}_${name}_fetch $R/gpurun_out/${TAG}_${name}_write ${name} "$KEY" "$KROWS" > /dev/null
  grep '^{' $R/gpurun_out/${TAG}_${name}_trace.log | tail -1 > $R/gpurun_out/profiles_${TAG}/${TAG}_bench_${name}.log
  rm -rf $R/gpurun_out/${TAG}_${name}_trace $R/gpurun_out/${TAG}_${name}_fetch $R/gpurun_out/${TAG}_${name}_write $R/gpurun_out/${TAG}_${name}_trace.log
  echo "profiled $name"
}
if [ "$2" = "k64" ]; then
  run 10_10_k64 --total-rows 5e7 --num-cols 10 --cat-cols 10 --keys 64 && run 10_10_k32 --total-rows 5e7 --num-cols 10 --cat-cols 10 --keys 32 && echo k64 collected
  exit 0
fi
if [ "$2" = "wide" ]; then
  run 20_20 --total-rows 5e7 --num-cols 20 --cat-cols 20 && echo wide collected
  exit 0
fi
if [ "$2" = "rest" ]; then
  run 10_10_k64 --total-rows 5e7 --num-cols 10 --cat-cols 10 --keys 64 && run 20_20 --total-rows 5e7 --num-cols 20 --cat-cols 20 \
    && run 10_10_k1000 --total-rows 5e7 --num-cols 10 --cat-cols 10 --keys 1000 && echo rest collected
  exit 0
fi
KEY=gram_kernel_20_0 KROWS=1000000000 run 20_0 && KEY=fused3_kernel_10_10 KROWS=100000000 run 10_10 --total-rows 1e8 --num-cols 10 --cat-cols 10 && KEY=fused2_kernel_nb_10_10 KROWS=100000000 run nb_10_10 --total-rows 1e8 --num-cols 10 --cat-cols 10 --nb \
  && KEY="" run 10_10_k64 --total-rows 5e7 --num-cols 10 --cat-cols 10 --keys 64 && run 20_20 --total-rows 5e7 --num-cols 20 --cat-cols 20 \
  && run 20_10 --total-rows 5e7 --num-cols 20 --cat-cols 10 && run 16_0 --total-rows 1e9 --num-cols 16 \
  && run 10_10_k1000 --total-rows 1e8 --num-cols 10 --cat-cols 10 --keys 1000 \
  && sh $R/tests/tools/sq_counters.sh ${TAG}_gram_20_0 gram_kernel \
  && sh $R/tests/tools/sq_counters.sh ${TAG}_fused3_10_10 fused3_kernel --total-rows 1e8 --num-cols 10 --cat-cols 10 \
  && sh $R/tests/tools/sq_counters.sh ${TAG}_fused2_nb_10_10 fused2_kernel --total-rows 1e8 --num-cols 10 --cat-cols 10 --nb \
  && echo profiles collected
