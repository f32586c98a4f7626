#!/bin/bash
# LDS and wait counters of k_queue for several library builds (development aid; run on the GPU box).
#   tools/pmc_lds.sh name=path/to/lib.so ...      (path '-' = the in-tree build)
export TMPDIR=/tmp
for spec in "$@"; do
  v=${spec%%=*}; path=${spec#*=}
  OUT=gpurun_out/pmc_lds_$v; rm -rf $OUT; mkdir -p $OUT
  LIB=""; [ "$path" != "-" ] && LIB=$PWD/$path
  TRC_LIB=$LIB timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -- python3 tools/prof_run.py 3 1 100 > $OUT/p1.out 2>&1 || { echo "$v: rocprofv3 failed"; tail -5 $OUT/p1.out; exit 1; }
  python3 - <<PY
import csv, glob
c = {}
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_queue" in r["Kernel_Name"]: c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
v = c["SQ_INSTS_VALU"]
print("$v:", " ".join(f"{k[3:]}/VALU={c[k] / v:.4f}" for k in sorted(c) if k != "SQ_INSTS_VALU"))
PY
done
