#!/bin/bash
# VALU instructions per 64 segments of k_queue as the scene grows object by object (development aid):
# TRC_NOBJ=5 walls, 6 +light cube, 7/8 +spheres, 9 +icosahedron.
export TMPDIR=/tmp
for n in 5 6 7 8 9; do
  OUT=gpurun_out/pmc_nobj_$n; rm -rf $OUT; mkdir -p $OUT
  TRC_NOBJ=$n timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 tools/prof_run.py 3 1 100 > $OUT/p1.out 2>&1
  python3 - <<PY
import csv, glob, re
c = {}
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_queue" in r["Kernel_Name"]: c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
m = re.search(r"rep 0: ([\d.]+) ms, (\d+) segs", open("$OUT/p1.out").read())
segs = int(m.group(2))
print("nobj $n: segs", segs, "VALU/64seg %.1f" % (c["SQ_INSTS_VALU"] / (segs / 64)), "SALU/64seg %.1f" % (c["SQ_INSTS_SALU"] / (segs / 64)),
      "lane util %.1f%%" % (100 * c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)))
PY
done
