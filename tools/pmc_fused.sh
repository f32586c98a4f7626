#!/bin/bash
# Two rocprofv3 --pmc passes over one fused launch of the headline workload; summary JSON on stdout.
# usage: tools/pmc_fused.sh <tag>     (run on the GPU box from the repo root)
set -e
export TMPDIR=/tmp
TAG=${1:-x}
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 tools/prof_run.py ${PIPE:-3} 1 > $OUT/p1.out 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 tools/prof_run.py ${PIPE:-3} 1 > $OUT/p2.out 2>&1
python3 tools/pmc_summary.py $OUT $TAG
