"""tools/pmc_summary.py <out_dir> <tag> -- summary of the rocprofv3 --pmc passes of tools/pmc_fused.sh for one kernel.

What the derived block means (it must reconcile with the un-profiled run):
  * duration: GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the XCD was busy, so / 8 = shader cycles of the
    launch (MI355X_MICROARCH.md, DVFS); the wall time rocprofv3 reports while counting is inflated and is NOT used.
    The clock is what the kernel's own wavefronts measured in the same process (s_memtime / s_memrealtime, printed by
    tools/prof_run.py), so kernel_ms_from_counters = cycles / clock is comparable with the un-profiled HIP-event time.
  * SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES are in QUAD-cycles (4 shader cycles), summed over the SIMDs: a wave64 VALU
    instruction is counted for about one quad (active_quads_per_valu_inst ~ 1.03) although a SIMD accepts the next
    instruction of ANOTHER wave after 2.5 cycles (tools/micro/issue_rate.hip). rocprof's VALUBusy formula
    (100 * sum / CU_NUM / GRBM_GUI_ACTIVE) therefore saturates at 4 / 2.5 = 160 %, not 100 %: it is reported as it is,
    and valu_issue_utilisation = 2.5 / (SIMD cycles per VALU instruction) is the fraction of the issue ceiling in use.
  * valu_lane_utilisation = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64): rocprof's VALUUtilization.
"""
import csv, glob, json, os, re, sys
out_dir, tag = sys.argv[1], sys.argv[2]
kname = os.environ.get("KNAME", "k_queue")
pipe = os.environ.get("PIPE", "3")
VALU_ISSUE_FLOOR = 2.5          # shader cycles a SIMD needs per wave64 VALU instruction (tools/micro/issue_rate.hip)
N_SIMD, N_XCD = 1024, 8
c = {}
for f in glob.glob(out_dir + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
            c["VGPR_Count"] = int(r["VGPR_Count"]); c["Accum_VGPR_Count"] = int(r["Accum_VGPR_Count"]); c["SGPR_Count"] = int(r["SGPR_Count"])
            c["Scratch_Size"] = int(r["Scratch_Size"]); c["LDS_Block_Size"] = int(r.get("LDS_Block_Size", 0) or 0)
txt = open(out_dir + "/p1.out").read()
m = re.search(r"rep 0: ([\d.]+) ms, (\d+) segs.*?paths (\d+), shader_mhz ([\d.]+)", txt)
ms_prof, segs, paths, mhz = float(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4))
cyc = c["GRBM_GUI_ACTIVE"] / N_XCD
per_inst = cyc * N_SIMD / c["SQ_INSTS_VALU"]
d = {"live_segments": segs, "paths": paths,
     "shader_cycles_of_the_launch": cyc, "in_kernel_clock_MHz": mhz,
     "kernel_ms_from_counters": cyc / (mhz * 1e3) if mhz else None,
     "kernel_ms_reported_under_profiler_not_used": ms_prof,
     "valu_insts_per_64_segments": c["SQ_INSTS_VALU"] / (segs / 64),
     "salu_insts_per_64_segments": c["SQ_INSTS_SALU"] / (segs / 64),
     "simd_cycles_per_valu_inst": per_inst,
     "valu_issue_floor_cycles": VALU_ISSUE_FLOOR,
     "valu_issue_utilisation": VALU_ISSUE_FLOOR / per_inst,
     "active_quads_per_valu_inst": c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"],
     "VALUBusy_rocprof_formula_pct": 100 * c["SQ_ACTIVE_INST_VALU"] / 256 / cyc,
     "VALUBusy_formula_saturates_at_pct": 100 * 4 / VALU_ISSUE_FLOOR,
     "valu_lane_utilisation_pct": 100 * c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64),
     "wave_occupancy_mean_waves_per_simd": c["SQ_WAVE_CYCLES"] * 4 / (cyc * N_SIMD) if "SQ_WAVE_CYCLES" in c else None}
def kernel_source_digest():
    """sha256 over the device code (tracer_amd/csrc/*.h, kernels.hip): bench.py flags a profile taken from other sources."""
    import glob, hashlib, os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(here, "tracer_amd", "csrc", "*.h")) + [os.path.join(here, "tracer_amd", "csrc", "kernels.hip")]):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


json.dump({"kernel": "trc::" + kname, "tag": tag, "kernel_source_digest": kernel_source_digest(),
           "workload": "cornell 512x512 mode 2, 300 iterations, one launch",
           "command": f"tools/pmc_fused.sh {tag} (two rocprofv3 --pmc passes of: python3 tools/prof_run.py {pipe} 1)",
           "units": {"SQ_ACTIVE_INST_VALU": "quad-cycles (4 shader cycles), summed over SIMDs", "SQ_WAVE_CYCLES": "quad-cycles",
                     "SQ_BUSY_CYCLES": "cycles per shader engine", "GRBM_GUI_ACTIVE": "cycles, summed over the 8 XCDs"},
           "counters": c, "derived": d}, open(out_dir + "/summary.json", "w"), indent=1)
print(json.dumps(d, indent=1))
