#!/bin/bash
# rocprofv3 passes over configs[4] (tools/mesh_bench.py, mode 1 only: every k_mesh_brute dispatch of a pass is the same
# 1024x1024 frame): kernel stats, then ONE SQ counter pass per TRC_OPT_MESH_ACCEL setting (0 = bounding-sphere pre-test,
# 2 = every pair through the exact arithmetic), summarised per dispatch -> $OUT/summary.json (profiles/<tag>_mesh_pmc.json).
# usage: tools/pmc_mesh.sh <tag> [level=8] [w=1024] [h=1024]
export TMPDIR=/tmp
TAG=${1:-x}; LV=${2:-8}; W=${3:-1024}; H=${4:-1024}
OUT=gpurun_out/mesh_$TAG; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/mesh_bench.py $LV $W $H 1 > $OUT/stats.out 2>&1 || exit 1
for A in 0 2; do
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/a$A -- python3 tools/mesh_bench.py $LV $W $H 0 -1 $A > $OUT/a$A.out 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, json, os, re, sys
sys.path.insert(0, "tools")
out = {"kernel": "trc::k_mesh_brute", "tag": "$TAG",
       "workload": "BASELINE.json configs[4]: icosphere level $LV, ${W}x${H}, mode 1 (tools/mesh_bench.py)",
       "command": "tools/pmc_mesh.sh $TAG: one rocprofv3 --pmc pass per TRC_OPT_MESH_ACCEL setting; every k_mesh_brute dispatch of a pass is the same frame",
       "units": {"SQ_ACTIVE_INST_VALU": "quad-cycles (4 shader cycles), summed over SIMDs", "GRBM_GUI_ACTIVE": "cycles, summed over the 8 XCDs"},
       "by_mesh_accel": {}}
import hashlib
here = os.getcwd()
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(here, "tracer_amd", "csrc", "*.h")) + [os.path.join(here, "tracer_amd", "csrc", "kernels.hip")]):
    h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
out["kernel_source_digest"] = h.hexdigest()[:16]
for a in ("0", "2"):
    per = {}          # dispatch id -> counter -> value
    for f in glob.glob("$OUT/a%s/**/*counter_collection.csv" % a, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_mesh_brute" in r["Kernel_Name"]:
                d = per.setdefault(r["Dispatch_Id"], {})
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                d["_vgpr"] = int(r["VGPR_Count"]); d["_lds"] = int(r.get("LDS_Block_Size", 0) or 0); d["_name"] = r["Kernel_Name"][:60]
    txt = open("$OUT/a%s.out" % a).read()
    pairs = [int(x) for x in re.findall(r"mode 1 rep \d+: [\d.]+ ms, tri_tests (\d+)", txt)]
    ms = [float(x) for x in re.findall(r"mode 1 rep \d+: ([\d.]+) ms", txt)]
    n = len(per)
    if not n or not pairs: continue
    mean = lambda k: sum(d[k] for d in per.values()) / n
    cyc = mean("GRBM_GUI_ACTIVE") / 8
    insts = mean("SQ_INSTS_VALU")
    any_d = next(iter(per.values()))
    out["by_mesh_accel"][a] = {
        "kernel": any_d["_name"], "dispatches_profiled": n, "vgprs": any_d["_vgpr"], "lds_bytes": any_d["_lds"],
        "pairs_per_frame": int(pairs[-1]), "frame_ms_under_profiler_not_used": ms,
        "valu_insts_per_dispatch": insts, "valu_insts_per_64_pairs": insts / (pairs[-1] / 64.0),     # wave64 instructions per 64 (ray, face) pairs = per-lane instructions per pair
        "salu_insts_per_dispatch": mean("SQ_INSTS_SALU"), "lds_insts_per_dispatch": mean("SQ_INSTS_LDS"),
        "shader_cycles_per_dispatch": cyc,
        "simd_cycles_per_valu_inst": cyc * 1024 / insts,
        "valu_issue_utilisation": {"measured_2.5_cycles_per_wave64_inst": 2.5 * insts / (cyc * 1024), "guide_2.0_cycles_per_wave64_inst": 2.0 * insts / (cyc * 1024)},
        "valu_lane_utilisation_pct": 100 * mean("SQ_THREAD_CYCLES_VALU") / (mean("SQ_ACTIVE_INST_VALU") * 64),
        "wave_occupancy_mean_waves_per_simd": mean("SQ_WAVE_CYCLES") * 4 / (cyc * 1024)}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read())
print(json.dumps(out, indent=1))
PY
