#!/bin/bash
# Diagnostic counter passes over one launch sequence of a pipeline (development aid): where do a kernel's wavefronts wait?
# usage: tools/pmc_diag.sh <tag> <pipe> <kernel substring>     (TRC_LIB selects the build)
export TMPDIR=/tmp
TAG=${1:-x}; PIPE=${2:-1}; K=${3:-k_bounce}
OUT=gpurun_out/diag_$TAG; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL GRBM_GUI_ACTIVE TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d $OUT/p1 -- python3 tools/prof_run.py $PIPE 1 > $OUT/p1.out 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/p2 -- python3 tools/prof_run.py $PIPE 1 > $OUT/p2.out 2>&1 || exit 1
python3 - <<PY
import csv, glob
c = {}
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$K" in r["Kernel_Name"]:
            c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
g = c.get("GRBM_GUI_ACTIVE", 0) / 2 / 8     # (two passes) / XCDs = shader cycles
for k in sorted(c): print(f"{k:36s} {c[k]:.6g}   per shader cycle {c[k] / g:.4g}")
PY
