#!/bin/bash
# instruction-mix passes for k_fused (headline workload)
export TMPDIR=/tmp
KNAME=${KNAME:-k_queue}
OUT=gpurun_out/pmc_mix_${1:-x}; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH --output-format csv -d $OUT/p1 -- python3 tools/prof_run.py ${PIPE:-3} 1 > $OUT/p1.out 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- python3 tools/prof_run.py ${PIPE:-3} 1 > $OUT/p2.out 2>&1
python3 - <<PY
import csv,glob
c={}
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KNAME" in r["Kernel_Name"]: c[r["Counter_Name"]]=c.get(r["Counter_Name"],0)+float(r["Counter_Value"])
trips=291502459/64
for k in sorted(c): print(f"{k:28s} {c[k]:.4g}  per-64-seg {c[k]/trips:.1f}")
PY
