#!/bin/bash
# Dynamic VALU cost of one branch of k_queue's shading: builds that evaluate the branch twice (tools/build_variant.sh
# dup_trans -DTRC_DUP_TRANS, dup_diff -DTRC_DUP_DIFF, dup_rng -DTRC_DUP_RNG) against the in-tree build; the difference in
# SQ_INSTS_VALU per 64 segments is what the branch costs per trip of the loop.   usage: tools/pmc_dup.sh  (on the GPU box)
export TMPDIR=/tmp
for v in base dup_trans dup_diff dup_rng; do
  OUT=gpurun_out/pmc_dup_$v; rm -rf $OUT; mkdir -p $OUT
  LIB=""; [ $v != base ] && LIB=$PWD/tracer_amd/variants/libtracer_$v.so
  TRC_LIB=$LIB timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 tools/prof_run.py 3 1 100 > $OUT/p1.out 2>&1
  python3 - <<PY
import csv, glob, re
c = {}
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_queue" in r["Kernel_Name"]: c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
m = re.search(r"rep 0: ([\\d.]+) ms, (\\d+) segs", open("$OUT/p1.out").read())
segs = int(m.group(2))
print("$v: VALU/64seg %.1f" % (c["SQ_INSTS_VALU"] / (segs / 64)), "SALU/64seg %.1f" % (c["SQ_INSTS_SALU"] / (segs / 64)),
      "lane util %.1f%%" % (100 * c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)))
PY
done
