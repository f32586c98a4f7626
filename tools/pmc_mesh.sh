#!/bin/bash
# rocprofv3 passes over configs[4] (tools/mesh_bench.py): kernel stats, then SQ counters of k_mesh_brute.
# usage: tools/pmc_mesh.sh <tag> [level=8] [w=1024] [h=1024]
export TMPDIR=/tmp
TAG=${1:-x}; LV=${2:-8}; W=${3:-1024}; H=${4:-1024}
OUT=gpurun_out/mesh_$TAG; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/mesh_bench.py $LV $W $H 1 > $OUT/stats.out 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1 -- python3 tools/mesh_bench.py $LV $W $H 0 > $OUT/p1.out 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 tools/mesh_bench.py $LV $W $H 0 > $OUT/p2.out 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read())
c = {}
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mesh_brute" in r["Kernel_Name"]:
            c.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k in sorted(c): print(f"{k:24s} n={len(c[k])} mean {sum(c[k])/len(c[k]):.5g}")
PY
