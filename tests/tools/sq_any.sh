#!/bin/sh
# SQ counters of one kernel of any python program (two rocprofv3 --pmc passes).
#   sh tests/tools/sq_any.sh <tag> <kernel-substring> <script.py> [args...]   -> gpurun_out/<tag>_sq.json
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
TAG=$1; KSUB=$2; shift 2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/${TAG}_sq_a -- python3 "$@" > $R/gpurun_out/${TAG}_sq_a.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/${TAG}_sq_b -- python3 "$@" > $R/gpurun_out/${TAG}_sq_b.log 2>&1 || exit 1
python3 $R/tests/tools/sq_summary.py $R/gpurun_out/${TAG}_sq.json "$KSUB" $R/gpurun_out/${TAG}_sq_a $R/gpurun_out/${TAG}_sq_b
rm -rf $R/gpurun_out/${TAG}_sq_a $R/gpurun_out/${TAG}_sq_b $R/gpurun_out/${TAG}_sq_a.log $R/gpurun_out/${TAG}_sq_b.log
