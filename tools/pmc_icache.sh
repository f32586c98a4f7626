#!/bin/bash
# Instruction-cache and issue-stall counters of one k_queue launch (development aid): tools/pmc_icache.sh <tag>
export TMPDIR=/tmp
TAG=${1:-x}; OUT=gpurun_out/pmc_ic_$TAG; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 tools/prof_run.py 3 1 > $OUT/p1.out 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES --output-format csv -d $OUT/p2 -- python3 tools/prof_run.py 3 1 > $OUT/p2.out 2>&1
python3 - <<PY
import csv, glob
for p in ("p1", "p2"):
    c = {}
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_queue" in r["Kernel_Name"]: c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    for k in sorted(c): print(p, k, "%.4g" % c[k])
PY
