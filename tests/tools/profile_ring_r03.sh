# round-3 profiles of the ring ops and the GROUP BY pool: kernel stats of ring_bench.py, kernel stats +
# HBM counters of the segmented GROUP BY (20_0, 1e4 groups, 1e8 rows) and of multiply (2_2 x 2_2, 2e6 rows)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_r03c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ringtrace -- python3 $R/tests/tools/ring_bench.py > $O/r03_bench_ring_ops.json 2> /dev/null || exit 1
PROFILES_OUT=$O python3 $R/profiles/summarize.py r03 $R/gpurun_out/ringtrace /nonexistent /nonexistent ring_ops > /dev/null
rm -rf $R/gpurun_out/ringtrace
one() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${name}_t -- python3 "$@" > $O/r03_bench_${name}.log 2>/dev/null || return 1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${name}_f -- python3 "$@" > /dev/null 2>&1 || return 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${name}_w -- python3 "$@" > /dev/null 2>&1 || return 1
  PROFILES_OUT=$O python3 $R/profiles/summarize.py r03 $R/gpurun_out/${name}_t $R/gpurun_out/${name}_f $R/gpurun_out/${name}_w $name > /dev/null
  rm -rf $R/gpurun_out/${name}_t $R/gpurun_out/${name}_f $R/gpurun_out/${name}_w
  echo "profiled $name"
}
one groups_20_0_G1e4 $R/tests/tools/groups_bench.py 20 1e4 1e8 && one multiply_2_2x2_2 $R/tests/tools/mul_bench.py 2e6 4
rm -f $O/traffic.json
ls $O
