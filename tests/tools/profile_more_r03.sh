R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; key=$2; krows=$3; shift 3
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_${name}_trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-calibration "$@" > $R/gpurun_out/r03_${name}_trace.log 2>&1 || return 1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r03_${name}_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calibration --no-check "$@" > /dev/null 2>&1 || return 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r03_${name}_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-calibration --no-check "$@" > /dev/null 2>&1 || return 1
  mkdir -p $R/gpurun_out/profiles_r03b
  cp $R/profiles/traffic.json $R/gpurun_out/profiles_r03b/traffic.json 2>/dev/null
  PROFILES_OUT=$R/gpurun_out/profiles_r03b python3 $R/profiles/summarize.py r03 $R/gpurun_out/r03_${name}_trace $R/gpurun_out/r03_${name}_fetch $R/gpurun_out/r03_${name}_write ${name} "$key" "$krows" > /dev/null
  grep '^{' $R/gpurun_out/r03_${name}_trace.log | tail -1 > $R/gpurun_out/profiles_r03b/r03_bench_${name}.log
  rm -rf $R/gpurun_out/r03_${name}_trace $R/gpurun_out/r03_${name}_fetch $R/gpurun_out/r03_${name}_write $R/gpurun_out/r03_${name}_trace.log
  echo "profiled $name"
}
run nb_10_10 nb_ring_kernel_nb_10_10 100000000 --total-rows 1e8 --num-cols 10 --cat-cols 10 --nb
run 1_0 gram_kernel_1_0 1000000000 --total-rows 1e9 --num-cols 1
run 2_0 gram_kernel_2_0 1000000000 --total-rows 1e9 --num-cols 2
sh $R/tests/tools/sq_counters.sh r03_nbring_10_10 nb_ring_kernel --total-rows 1e8 --num-cols 10 --cat-cols 10 --nb
