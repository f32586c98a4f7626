#!/bin/bash
# VALU / SALU / VMEM instruction counts of k_queue per 64 live segments for several library builds (development aid).
#   usage (on the GPU box): tools/pmc_ab.sh name=path/to/lib.so ...      (path '-' = the in-tree build)
export TMPDIR=/tmp
for spec in "$@"; do
  v=${spec%%=*}; path=${spec#*=}
  OUT=gpurun_out/pmc_ab_$v; rm -rf $OUT; mkdir -p $OUT
  LIB=""; [ "$path" != "-" ] && LIB=$PWD/$path
  TRC_LIB=$LIB timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 tools/prof_run.py 3 1 100 > $OUT/p1.out 2>&1 || { echo "$v: rocprofv3 failed"; tail -5 $OUT/p1.out; exit 1; }
  python3 - <<PY
import csv, glob, re
c = {}
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_queue" in r["Kernel_Name"]: c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
m = re.search(r"rep 0: ([\\d.]+) ms, (\\d+) segs", open("$OUT/p1.out").read())
segs = int(m.group(2)); t = segs / 64
print("$v: per 64 segs: VALU %.1f SALU %.1f VMEM rd %.1f wr %.1f LDS %.1f | lane util %.1f%% | VALU busy quad-cycles/inst %.2f" % (
    c["SQ_INSTS_VALU"] / t, c["SQ_INSTS_SALU"] / t, c["SQ_INSTS_VMEM_RD"] / t, c["SQ_INSTS_VMEM_WR"] / t, c["SQ_INSTS_LDS"] / t,
    100 * c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64), c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]))
PY
done
