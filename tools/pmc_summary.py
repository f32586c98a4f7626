import csv, glob, json, re, sys
out_dir, tag = sys.argv[1], sys.argv[2]
import os
kname = os.environ.get("KNAME", "k_queue")
c = {}
for f in glob.glob(out_dir + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
            c["VGPR_Count"] = int(r["VGPR_Count"]); c["Accum_VGPR_Count"] = int(r["Accum_VGPR_Count"]); c["SGPR_Count"] = int(r["SGPR_Count"])
            c["Scratch_Size"] = int(r["Scratch_Size"])
m = re.search(r"rep 0: ([\d.]+) ms, (\d+) segs", open(out_dir + "/p1.out").read())
ms, segs = float(m.group(1)), int(m.group(2))
cyc = c["GRBM_GUI_ACTIVE"] / 8
d = {"kernel_ms_under_profiler": ms, "live_segments": segs,
     "effective_clock_GHz": cyc / (ms * 1e-3) / 1e9,
     "VALUBusy_gfx94x_formula_pct": 100 * c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
     "valu_issue_cycles_per_inst_per_simd": cyc * 1024 / c["SQ_INSTS_VALU"],
     "valu_lane_utilisation_pct": 100 * c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64),
     "valu_insts_per_64_segments": c["SQ_INSTS_VALU"] / (segs / 64),
     "salu_insts_per_64_segments": c["SQ_INSTS_SALU"] / (segs / 64)}
json.dump({"kernel": "trc::" + kname, "tag": tag,
           "workload": "cornell 512x512 mode 2, 300 iterations, one launch",
           "command": "tools/pmc_fused.sh (two rocprofv3 --pmc passes of python3 tools/prof_run.py 2 1)",
           "counters": c, "derived": d}, open(out_dir + "/summary.json", "w"), indent=1)
print(json.dumps(d, indent=1))
