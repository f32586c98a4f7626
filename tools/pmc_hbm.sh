#!/bin/bash
# HBM traffic of one launch of the headline workload: FETCH_SIZE and WRITE_SIZE in separate rocprofv3
# --pmc passes (MI355X_MICROARCH.md, HBM section: KB units; FETCH_SIZE x2 on gfx950 for wide reads).
# usage: tools/pmc_hbm.sh <tag>
export TMPDIR=/tmp
OUT=gpurun_out/pmc_hbm_${1:-x}; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- python3 tools/prof_run.py ${PIPE:-3} 1 > $OUT/p1.out 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p2 -- python3 tools/prof_run.py ${PIPE:-3} 1 > $OUT/p2.out 2>&1
python3 - <<PY
import csv, glob, json
c = {}
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        c.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        c[k][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for k, v in c.items():
    out[k] = {"FETCH_SIZE_KB_raw": v.get("FETCH_SIZE"), "WRITE_SIZE_KB_raw": v.get("WRITE_SIZE"),
              "hbm_read_MB_gfx950_corrected_x2": (v.get("FETCH_SIZE") or 0) * 2 / 1024,
              "hbm_write_MB": (v.get("WRITE_SIZE") or 0) / 1024}
json.dump({"workload": "cornell 512x512 mode 2, 300 iterations, one launch", "per_kernel": out}, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
