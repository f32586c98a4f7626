#!/usr/bin/env python3
"""bench.py -- headline benchmark of the per-bounce path-tracing loop on MI355X.

Metric (BASELINE.json): Mpath-segments/s, Cornell box 512x512 path trace (mode 2), 300
iterations. A path segment is one closest-hit evaluation for a live path inside one bounce
pass (reference src/raytrace.cu:119-127); segments are counted on the device.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one full 300-iteration render of the 512x512 Cornell box per GPU, inputs (scene,
camera) resident on the device. For N > 1 (launched by torch.distributed.run, one rank per
GPU) rank r renders iterations r+1, r+1+N, ... (300 of them: sample-index partition, weak
scaling) and every step ends with ONE sum-reduce of the film buffers to rank 0 over RCCL,
followed by the display transform on rank 0 -- the only exchange the path has
(tracer_amd.multi.render_partitioned, the product's own multi-GPU entry).

Rank 0 prints one JSON line. Beside the contract's fields it carries
  roofline        the dominant kernel (k_queue) against the FP32 vector peak and against the measured VALU issue ceiling
  film_check      N == 1: sha1 of the film and LDR image the timed steps left behind against the value recorded from the
                  reference host build (tests/golden/meta.json); N > 1: the reduced film against the one-rank film
  cpu_baseline    the reference host build (oracle/_ref) on the host cores, one process per core (bounded sample) [N == 1]
  multi_gpu       N > 1: duration of the film sum-reduce, the one-GPU rate on the same workload, efficiency_vs_n1
  extra_configs   short runs of BASELINE.json configs[2], [3], [4] and of the compacted pipeline [N == 1]
  config4_strong  configs[3]'s shape, 3840x2160, a fixed number of samples split over the N GPUs (strong scaling)
Only the `cpu_baseline` leg touches oracle/.
"""
import argparse
import json
import os
import statistics
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

W = H = 512
ITERS = 300
SEED = 1

# algorithmic FP32 flop of the reference's arithmetic on the default Cornell
# scene (9 object tests / segment, brute-force 20-triangle mesh behind an AABB).
FLOP_PER_SEGMENT = 880.0
FLOP_PER_PATH = 118.0          # ray generation (~103) + accumulate (~15)
FLOP_PER_TRI_TEST = 74.0       # one full Moeller-Trumbore test, Ray.inl:189-226 (configs[4])
BYTES_PER_SEGMENT_REF = 108.0  # reference buffer layout, SURVEY.md 8(d)
PEAK_FP32_TFLOPS = 157.3       # MI355X FP32 vector peak: v_pk_fma_f32, packed AND fused (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
N_SIMD = 1024                  # 256 CUs x 4
# Two issue-cost constants, every utilisation below is printed against BOTH, labelled:
#   2.5 cycles per wave64 instruction: tools/micro/issue_rate.hip on this chip -- streams of independent v_add_f32 /
#       v_mul_f32 / v_mov_b32 issue at 5.0 cycles from one wave and 2.5 per SIMD from two, and (round 3,
#       profiles/r03_issue_rate_occupancy.txt) stay at 2.5 by the wall clock at 4, 6 and 8 resident waves: a SIMD serves its
#       two oldest wavefronts and the younger ones wait (3 waves take as long as 4). Mixed streams reach 2.3 at 8 waves.
#   2.0 cycles: the microarchitecture guide's figure (MI355X_MICROARCH.md, constants table). k_mesh_brute<false> -- unfused
#       multiplies, adds, compares and LDS broadcasts, 6.9 resident waves per SIMD -- sustains 2.07 by the PMC counters
#       (profiles/r03_d_mesh_pmc.json), so 2.5 is NOT a hard floor for a real instruction mix: price against 2.0.
# The numerical contract (-ffp-contract=off) allows no fusing, so the ceiling of this path is one add-or-multiply per lane
# per issue slot.
VALU_ISSUE_CYCLES = 2.5
VALU_ISSUE_CYCLES_GUIDE = 2.0
ISSUE_CONSTANTS = (("measured_2.5_cycles_per_wave64_inst", VALU_ISSUE_CYCLES), ("guide_2.0_cycles_per_wave64_inst", VALU_ISSUE_CYCLES_GUIDE))
ISSUE_CONSTANT_SOURCES = {"measured_2.5_cycles_per_wave64_inst": "tools/micro/issue_rate.hip on this chip, single-instruction streams at 2 to 8 waves per SIMD "
                                                                 "(profiles/r02_issue_rate.txt, profiles/r03_issue_rate_occupancy.txt); not a hard floor: "
                                                                 "k_mesh_brute<false> sustains 2.07 (profiles/r03_d_mesh_pmc.json)",
                          "guide_2.0_cycles_per_wave64_inst": "/opt/skills/guides/MI355X_MICROARCH.md, constants table: the bound to price against"}
C4_W, C4_H, C4_TOTAL_ITERS = 3840, 2160, 512      # config4_strong: an eighth of configs[3]'s 4096 samples per pixel


def latest_profile(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(HERE, "profiles", pattern)))
    return files[-1] if files else None


def measured_valu_mix():
    """(VALU wave-instructions per 64 live segments, live segments of the profiled launch, file) of k_queue from the
    latest committed rocprofv3 PMC pass (tools/pmc_fused.sh -> profiles/*_pmc_k_queue.json)."""
    f = latest_profile("*_pmc_k_queue.json")
    if not f:
        return None, None, None, None
    try:
        j = json.load(open(f))
        d = j["derived"]
        return float(d["valu_insts_per_64_segments"]), int(d["live_segments"]), os.path.basename(f), j.get("kernel_source_digest")
    except (KeyError, ValueError):
        return None, None, None, None


def kernel_source_digest():
    """As tools/pmc_summary.py records it: sha256 over the device code, tracer_amd/csrc/*.h and kernels.hip (the sources travel with the library)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(HERE, "tracer_amd", "csrc", "*.h")) + [os.path.join(HERE, "tracer_amd", "csrc", "kernels.hip")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def measured_hbm_traffic(kernel_substr, pattern, whole_call=False):
    """HBM bytes per launch from the committed PMC passes (tools/pmc_hbm.sh: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE,
    separate passes). whole_call: every kernel and fill of the call except the display transform."""
    f = latest_profile(pattern)
    if not f:
        return None, None
    try:
        per = json.load(open(f))["per_kernel"]
        if whole_call:
            mb = sum(v["hbm_read_MB_gfx950_corrected_x2"] + v["hbm_write_MB"] for k, v in per.items()
                     if "tonemap" not in k and "copyBuffer" not in k)
        else:
            k = next(v for name, v in per.items() if kernel_substr in name)
            mb = k["hbm_read_MB_gfx950_corrected_x2"] + k["hbm_write_MB"]
        return int(mb * 1048576), os.path.basename(f)
    except (KeyError, ValueError, StopIteration):
        return None, None


def cpu_processes(kind, n_procs, iters_each):
    """`n_procs` child processes (oracle/cpu_worker.py: numpy + ctypes only, no GPU), each rendering `iters_each`
    iterations of the 512x512 workload on a disjoint iteration stride with CPU checker `kind`, all released at one
    instant. Returns (Mseg/s over the span from the release to the last process's end, segments, seconds)."""
    import subprocess
    start_at = time.time() + 4.0                       # interpreter + numpy start-up of the children
    worker = os.path.join(HERE, "oracle", "cpu_worker.py")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, worker, kind, str(W), str(H), str(SEED), str(1 + r), str(n_procs),
                               str(iters_each), repr(start_at)], stdout=subprocess.PIPE, env=env, text=True)
             for r in range(n_procs)]
    res = []
    for p in procs:
        out, _ = p.communicate(timeout=600)
        if p.returncode != 0:
            raise RuntimeError(f"cpu_worker exited with {p.returncode}")
        res.append(json.loads(out.strip().splitlines()[-1]))
    segs = sum(r["live_segments"] for r in res)
    t_begin = min(start_at, min(r["t0"] for r in res))   # a process that missed the release started late: counted from the release
    span = max(r["t1"] for r in res) - t_begin
    return segs / span / 1e6, segs, span


def cpu_baseline():
    """The reference's CPU path on this box's host cores over a bounded sample of the same workload.

    value: oracle/_ref/libtracer_ref.so -- the reference's own Ray/Object/Material/Mesh/Utils .inl + GLM compiled by
    oracle/Makefile, single-threaded by construction -- run as one process per core on disjoint iteration strides
    (kind "reference"). Beside it the plain-C restatement (bit-identical to that build; OpenMP over pixels and
    single-threaded). Without oracle/_ref (a tree that never saw /root/reference) the restatement is the value, kind "port"."""
    from oracle.pyoracle import Oracle, available
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    o = Oracle("c")
    o.set_math_mode(Oracle.MATH_LIBM)
    s = o.scene_create(0)
    cam = o.camera_default(W, H)
    # the restatement, single thread: 4 iterations of the 512x512 workload
    t = time.perf_counter()
    r = o.pathtrace(s, W, H, cam, SEED, 1, 4, want_pbo=False, nthreads=1)
    v1 = r["stats"]["live_segments"] / (time.perf_counter() - t) / 1e6
    # the restatement, OpenMP over pixels: 32 iterations
    n_mt = 32
    t = time.perf_counter()
    r = o.pathtrace(s, W, H, cam, SEED, 1, n_mt, want_pbo=False, nthreads=threads)
    vm = r["stats"]["live_segments"] / (time.perf_counter() - t) / 1e6
    port = {"value": round(vm, 3), "unit": "Mseg/s", "cores": threads, "single_thread_value": round(v1, 3),
            "sample": f"{n_mt} of {ITERS} iterations, OpenMP over pixels ({threads} threads); single thread: 4 iterations"}
    if not available("ref"):
        return {"value": port["value"], "unit": "Mseg/s", "cores": threads, "kind": "port",
                "sample": f"cornell {W}x{H} mode 2, " + port["sample"], "single_thread_value": port["single_thread_value"],
                "note": "oracle/_ref (the reference host build) is not in this tree: the plain-C restatement was timed"}
    iters_each = 12
    v_one, _, _ = cpu_processes("ref", 1, iters_each)
    v_all, segs, span = cpu_processes("ref", threads, iters_each)
    return {"value": round(v_all, 3), "unit": "Mseg/s", "cores": threads, "kind": "reference",
            "sample": f"cornell {W}x{H} mode 2, {threads * iters_each} of {ITERS} iterations: {threads} processes of the reference host "
                      f"build (oracle/_ref/libtracer_ref.so, single-threaded), {iters_each} iterations each on disjoint iteration strides, "
                      f"released together; {segs} live segments in {span:.2f} s",
            "single_thread_value": round(v_one, 3),
            "parallel_efficiency": round(v_all / (threads * v_one), 3),
            "port": port}


def check_film_against_reference(film, tr):
    """N == 1, untimed: the film and the LDR image the timed steps left behind -- seed 1, iterations 1..300 of the
    512x512 Cornell box -- against the sha1 that tools/make_golden.py recorded in tests/golden/meta.json from the
    reference host build with the kernels' sin/cos/acos/pow interposed (oracle/_ref/libtracer_ref_pm.so; 2.5 minutes on
    one CPU core, in the build container). Byte for byte; raytrace.cu:162-177 is the running sum it pins."""
    import hashlib
    import numpy as np
    try:
        meta = json.load(open(os.path.join(HERE, "tests", "golden", "meta.json")))
        want_film = meta["hashes"]["mode2_512_seed1_film300_pm"]
        want_ldr = meta["hashes"]["mode2_512_seed1_ldr300_pm"]
    except (OSError, KeyError, ValueError) as e:
        return {"ok": None, "note": f"no pinned 300-iteration film in tests/golden/meta.json ({e!r})"}
    host = film.detach().cpu().numpy()
    got_film = hashlib.sha1(np.ascontiguousarray(host).tobytes()).hexdigest()
    got_ldr = hashlib.sha1(np.ascontiguousarray(tr.read_ldr()).tobytes()).hexdigest()
    mean = [float(x) for x in host.reshape(-1, 3).astype(np.float64).mean(0) / ITERS]
    return {"against": "reference host build + portable sin/cos/acos/pow (oracle/_ref/libtracer_ref_pm.so), tests/golden/meta.json",
            "iterations": ITERS, "film_sha1": got_film, "film_sha1_expected": want_film, "ldr_sha1": got_ldr,
            "ldr_sha1_expected": want_ldr, "film_mean": [round(x, 6) for x in mean],
            "ok": bool(got_film == want_film and got_ldr == want_ldr)}


def unfused_peak_tflops(mhz, issue_cycles=VALU_ISSUE_CYCLES):
    return 64 * N_SIMD * mhz * 1e6 / issue_cycles / 1e12


def issue_utilisation(valu_insts_total, kernel_ms, mhz):
    """VALU issue utilisation of a launch = the SIMD cycles its wave64 VALU instructions need at the SIMD's best issue rate
    / the SIMD cycles it ran, for both issue constants."""
    cycles = kernel_ms * 1e-3 * mhz * 1e6
    return {name: round(valu_insts_total / N_SIMD * c / cycles, 4) for name, c in ISSUE_CONSTANTS}


def mesh_profile():
    """configs[4]: VALU instructions per (ray, face) pair of k_mesh_brute from the committed PMC pass (tools/pmc_mesh.sh ->
    profiles/*_mesh_pmc.json, one entry per TRC_OPT_MESH_ACCEL setting), and the reference's flop per pair weighted by where
    Ray.inl:189-226 leaves (tools/mesh_exit_rates.py -> profiles/*_mesh_exit_rates.json, counted by the oracle once)."""
    prof, rates = None, None
    f = latest_profile("*_mesh_pmc.json")
    if f:
        try:
            prof = json.load(open(f)); prof["_file"] = os.path.basename(f)
        except ValueError:
            prof = None
    f = latest_profile("*_mesh_exit_rates.json")
    if f:
        try:
            rates = json.load(open(f)); rates["_file"] = os.path.basename(f)
        except ValueError:
            rates = None
    return prof, rates


def timed_render(tr, cam, seed, iters, reps, pipeline=None):
    """`reps` calls of `iters` iterations; (min kernel ms, median kernel ms, counters of one call)."""
    if pipeline is not None:
        tr.set_pipeline(pipeline)
    tr.pathtrace(cam, seed, 1, iters); tr.synchronize()           # warm-up (allocations, code load)
    ms = []
    for _ in range(reps):
        tr.pathtrace(cam, seed, 1, iters)
        ms.append(tr.last_kernel_ms())
    tr.reset_counters(); tr.pathtrace(cam, seed, 1, iters); cnt = tr.counters()
    return min(ms), statistics.median(ms), cnt


def extra_configs(ta, k_queue_ms=None):
    """Short runs of the other BASELINE.json configurations on one GPU (rank 0, N == 1). Rates are per call, from HIP
    events on the launch stream; rooflines use the nominal algorithmic costs of SURVEY.md 8(d)."""
    out = []

    def seg_entry(name, config, w, h, variant, iters, reps, pipeline):
        sc = ta.Scene(variant)
        tr = ta.Tracer(w, h)
        tr.upload_scene(sc.objects_ptr, sc.size)
        k_min, k_med, cnt = timed_render(tr, ta.camera_default(w, h), SEED, iters, reps, pipeline)
        seg, paths, mhz = cnt["live_segments"], cnt["paths"], cnt.get("shader_mhz")
        tf = (seg * FLOP_PER_SEGMENT + paths * FLOP_PER_PATH) / (k_min * 1e-3) / 1e12
        e = {"name": name, "config": config, "width": w, "height": h, "iterations": iters,
             "pipeline": {1: "compacted", 3: "work-queue"}[pipeline],
             "kernel_ms_min": round(k_min, 4), "kernel_ms_median": round(k_med, 4),
             "value": round(seg / k_min / 1e3, 1), "unit": "Mseg/s", "live_segments": seg,
             "roofline": {"bound": "valu", "achieved": round(tf, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(tf / PEAK_FP32_TFLOPS, 4), "flop_per_segment": FLOP_PER_SEGMENT}}
        if mhz:
            e["roofline"]["frac_unfused_at_clock"] = {name: round(tf / unfused_peak_tflops(mhz, c), 4) for name, c in ISSUE_CONSTANTS}
        tr.close()
        return e

    out.append(seg_entry("glass icosahedron + thin lens, 1024x1024", "BASELINE.json configs[2] (64 of its 1024 iterations)",
                         1024, 1024, 1, 64, 3, ta.PIPE_QUEUE))
    out.append(seg_entry("cornell 3840x2160", "BASELINE.json configs[3] (20 of its 4096 iterations, one GPU)",
                         C4_W, C4_H, 0, 20, 3, ta.PIPE_QUEUE))
    e = seg_entry("cornell 512x512, compacted wavefront pipeline", "BASELINE.json configs[1] through TRC_PIPE_COMPACTED: per-bounce ray "
                  "state in HBM, ballot compaction (the north_star's HBM evidence)", W, H, 0, ITERS, 5, ta.PIPE_COMPACTED)
    t_bytes, src = measured_hbm_traffic(None, "*_compacted_hbm_summary.json", whole_call=True)
    if t_bytes:
        gbs = t_bytes / (e["kernel_ms_min"] * 1e-3) / 1e9
        e["hbm"] = {"traffic": t_bytes, "traffic_source": src, "achieved_GBs": round(gbs, 1), "peak_GBs": PEAK_HBM_GBS,
                    "hbm_frac": round(gbs / PEAK_HBM_GBS, 4),
                    "algorithmic_bytes": int(e["live_segments"] * 80 + 12 * W * H * ITERS * 2),
                    "algorithmic_note": "40 B of ray state read and 40 B written per live segment (the reference's per-bounce buffers, "
                                        "raytrace.cu:107-160) + 12 B per path staged and read back once"}
        if k_queue_ms:
            # the same closest-hit arithmetic bounds this pipeline as it bounds k_queue: its HBM rate cannot exceed the
            # traffic over the time the arithmetic alone takes (the work-queue pipeline's kernel time, which moves no ray state)
            e["hbm"]["hbm_frac_ceiling_set_by_the_valu"] = round(t_bytes / (k_queue_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
    out.append(e)

    # configs[0]'s kernel (mode 1: one closest hit + flat shade per pixel, raytrace.cu:29-62) at 3840x2160, and the same
    # primary rays as a caller-supplied batch through trc_intersect_rays_device (the picking path, main.cpp:180-208)
    sc = ta.Scene(0)
    tr = ta.Tracer(C4_W, C4_H)
    tr.upload_scene(sc.objects_ptr, sc.size)
    tr.set_record_hits(False)
    cam = ta.camera_default(C4_W, C4_H)
    tr.raytrace1(cam); tr.synchronize()
    ms = []
    for _ in range(5):
        tr.raytrace1(cam); ms.append(tr.last_render_ms())
    n_rays = C4_W * C4_H
    out.append({"name": "cornell 3840x2160, mode 1 (primary closest hit + flat shade)", "config": "BASELINE.json configs[0]'s kernel at 4K",
                "frame_ms_min": round(min(ms), 4), "frame_ms_median": round(statistics.median(ms), 4),
                "value": round(n_rays / (min(ms) * 1e-3) / 1e9, 3), "unit": "Grays/s"})
    tr.close()

    # configs[4]: one MESH of 1 310 720 faces, 1024x1024, mode 1, every face against every ray that passed the AABB
    sc = ta.Scene(2, 8)
    w = h = 1024
    tr = ta.Tracer(w, h)
    tr.upload_scene(sc.objects_ptr, sc.size)
    cam = ta.camera_default(w, h)
    for accel, label in ((0, "exhaustive loop, bounding-sphere pre-test (default)"), (2, "exhaustive loop, every pair through the exact arithmetic"),
                         (1, "BVH in front of the same exact test (SURVEY 8f-3, not the configs[4] definition)")):
        tr.set_mesh_accel(accel)
        tr.raytrace1(cam); tr.synchronize()
        ms = []
        for _ in range(3 if accel != 2 else 2):
            tr.raytrace1(cam); ms.append(tr.last_render_ms())
        tr.reset_counters(); tr.raytrace1(cam); cnt = tr.counters()
        k_min = min(ms)
        pairs = cnt["tri_tests"]
        e = {"name": f"icosphere 1 310 720 faces, 1024x1024, mode 1: {label}", "config": "BASELINE.json configs[4]", "mesh_accel": accel,
             "frame_ms_min": round(k_min, 4), "frame_ms_median": round(statistics.median(ms), 4),
             "ray_face_pairs": pairs, "value": round(pairs / (k_min * 1e-3) / 1e12, 4), "unit": "T (ray, face) pairs/s"}
        if accel != 1:
            # The reference leaves its triangle test at Ray.inl:196 / :203 / :208 for almost every pair; the flop of a pair is
            # the exit-stage costs of SURVEY.md 8(d) (20 / 30 / 46 / 74) weighted by the exit rates the oracle counted on this
            # frame. The kernel is issue-bound: instructions per pair from the committed PMC pass, clock from its own wavefronts.
            prof, rates = mesh_profile()
            rl = {"bound": "valu issue", "kernel": "k_mesh_brute<%s>" % ("true" if accel == 0 else "false")}
            p = (prof or {}).get("by_mesh_accel", {}).get(str(accel))
            mhz = cnt.get("shader_mhz")
            if p:
                # What the kernel EXECUTES against what the SIMDs can issue: per-lane VALU instructions per pair from the
                # committed PMC pass, and instructions AND cycles of that same profiled launch (GRBM_GUI_ACTIVE / 8) for the
                # fraction -- self-consistent, whatever the clock did between this run and that one. Peak: one wave64
                # instruction per SIMD per 2.0 cycles (the guide's constant; bench.py's header on why not 2.5).
                ipp = p["valu_insts_per_64_pairs"]                                           # = per-lane instructions per pair
                clk = (mhz or 0.0) * 1e6
                rl.update({"achieved": round(pairs * ipp / (k_min * 1e-3) / 1e12, 3),
                           "peak": round(64.0 * N_SIMD * clk / VALU_ISSUE_CYCLES_GUIDE / 1e12, 3) if clk else None,
                           "unit": "T lane-instructions/s (VALU, executed)",
                           "frac": round(VALU_ISSUE_CYCLES_GUIDE / p["simd_cycles_per_valu_inst"], 4),
                           "frac_source": "VALU wave-instructions and shader cycles of the profiled launch (%s)" % prof["_file"]})
                rl["valu_insts_per_64_pairs"] = round(ipp, 3)
                rl["simd_cycles_per_valu_inst"] = round(p["simd_cycles_per_valu_inst"], 4)
                rl["valu_issue_utilisation"] = {name: round(c / p["simd_cycles_per_valu_inst"], 4) for name, c in ISSUE_CONSTANTS}
                rl["issue_constants"] = ISSUE_CONSTANT_SOURCES
                rl["wave_occupancy_mean_waves_per_simd"] = round(p.get("wave_occupancy_mean_waves_per_simd", 0.0), 2)
                rl["valu_insts_source"] = prof["_file"]
                rl["valu_insts_profile_matches_run"] = bool(p["pairs_per_frame"] == pairs and prof.get("kernel_source_digest") == kernel_source_digest())
                rl["frame_shader_cycles_in_profile"] = round(p["shader_cycles_per_dispatch"])
                if mhz:
                    rl["shader_mhz"] = round(mhz, 1)
                    rl["frame_shader_cycles_live"] = round(k_min * 1e-3 * mhz * 1e6)
            if rates:
                fpp = float(rates["flop_per_pair_weighted"])
                tf = pairs * fpp / (k_min * 1e-3) / 1e12
                e["reference_arithmetic"] = {
                    "tflops_equivalent": round(tf, 2), "flop_per_pair": round(fpp, 3), "flop_per_pair_source": rates["_file"],
                    "exit_rates_Ray.inl_196_203_208_none": [round(rates["exit_rates"][k], 8) for k in ("det", "bary_x", "bary_y", "full")],
                    "note": "how fast the REFERENCE's arithmetic gets done, not what the kernel issues: flop per pair = exit-stage costs "
                            "20/30/46/74 (SURVEY.md 8d) weighted by where Ray.inl:189-226 leaves on this frame. With mesh_accel 0 nearly "
                            "every pair is decided by a conservative bounding-sphere test of about ten instructions (same hits, bit for "
                            "bit), so this figure may exceed the FP32 peak and is no roofline fraction; the roofline object prices the "
                            "instructions the kernel executes. The frame time also holds k_mesh_rays and k_raytrace1 (< 1 %)."}
            e["roofline"] = rl
        out.append(e)
    tr.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pipeline", type=int, default=3, help="0 wavefront, 1 compacted, 2 fused, 3 work queue (default), 4 auto (= 3 on this scene)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_configs and config4_strong")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import tracer_amd as ta
    from tracer_amd import multi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # TRC_BENCH_REHEARSAL=1: every rank on cuda:0 and gloo instead of RCCL (which refuses two ranks on one device):
    # a check of the N > 1 control flow on a one-GPU box, flagged in the output line; not a measurement.
    rehearsal = os.environ.get("TRC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    class _HostReduce:
        """Rehearsal only (all ranks on one GPU): RCCL refuses two ranks on one device, so the sum goes over gloo
        through the host. Same interface as torch.distributed for multi.reduce_film."""
        ReduceOp = dist.ReduceOp
        is_available = staticmethod(lambda: True)
        is_initialized = staticmethod(lambda: True)
        get_world_size = staticmethod(lambda: world)

        @staticmethod
        def reduce(t, dst=0, op=None):
            host = t.cpu()
            dist.reduce(host, dst=dst, op=op)
            if rank == dst:
                t.copy_(host)

    reducer = _HostReduce if rehearsal else dist

    # scene + camera through the product's own host API (no oracle here)
    scene = ta.Scene(0)
    cam = ta.camera_default(W, H)
    # ONE stream for everything: torch's zero_(), the tracer's launches and the RCCL reduce are then ordered by
    # the stream itself (render_partitioned would otherwise order them with host-side waits).
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    assert work_stream.cuda_stream != 0
    tr = ta.Tracer(W, H, device=local_rank, stream=work_stream.cuda_stream)
    tr.set_pipeline(args.pipeline)
    tr.upload_scene(scene.objects_ptr, scene.size)
    film = torch.zeros(W * H * 3, dtype=torch.float32, device=dev)   # torch owns the film so RCCL can reduce it
    tr.bind_film(film.data_ptr())

    reduce_events = []

    def step():
        # zero the film, render iterations rank+1, rank+1+world, ... of ITERS x world, ONE sum-reduce to rank 0,
        # display transform there
        multi.render_partitioned(tr, cam, SEED, ITERS * world, film, rank, world, reducer,
                                 timing=reduce_events if (world > 1 and not rehearsal) else None)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    reduce_events.clear()
    tr.reset_counters()
    kernel_ms, step_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        # HIP events recorded on the launch stream around the path kernels of this step (the query waits for the
        # kernels, which the next step's film.zero_() would do anyway)
        kernel_ms.append(tr.last_kernel_ms())
        step_ms.append((time.perf_counter() - ts) * 1e3)
    fence()
    elapsed = time.perf_counter() - t0
    cnt = tr.counters()
    # the film sum-reduce by itself (N > 1): torch events around dist.reduce on this rank's stream; on the root that is
    # the wait for the slowest rank plus the transfer, so the minimum over the steps is the closest to the transfer alone
    reduce_ms = [a.elapsed_time(b) for a, b in reduce_events]

    # N > 1: what ONE GPU does alone on the same per-GPU workload (rank 0, the others wait at the barrier), so that the
    # line carries its own weak-scaling efficiency = value / (N x that)
    n1_value = None
    if world > 1:
        if rank == 0:
            k1 = max(5, min(args.steps, 20))
            for _ in range(2):
                film.zero_(); tr.pathtrace(cam, SEED, 1, ITERS)
            torch.cuda.synchronize()
            tr.reset_counters()
            t1 = time.perf_counter()
            for _ in range(k1):
                film.zero_(); tr.pathtrace(cam, SEED, 1, ITERS)
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            n1_value = tr.counters()["live_segments"] / dt1 / 1e6
        fence()

    # Untimed self-check of the multi-GPU path: the reduced film must be the one-rank film of the same ITERS x world
    # iterations up to float summation order (1e-5 * sqrt(iterations), relative to the film's scale).
    film_check = None
    if world == 1:
        film_check = check_film_against_reference(film, tr)
    if world > 1:
        step(); torch.cuda.synchronize()
        if rank == 0:
            reduced = film.clone()
            film.zero_(); tr.pathtrace(cam, SEED, 1, ITERS * world); torch.cuda.synchronize()     # all iterations on this one GPU
            scale = float(film.max().item())
            err = float((reduced - film).abs().max().item())
            tol = 1e-5 * (ITERS * world) ** 0.5 * scale
            film_check = {"max_abs_difference_to_one_rank_film": err, "tolerance_1e-5_sqrt_iters": tol, "ok": bool(err <= tol)}

    if world > 1:
        cdev = torch.device("cpu") if rehearsal else dev
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        seg = torch.tensor([cnt["live_segments"], cnt["paths"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(seg, op=dist.ReduceOp.SUM)
        total_segments = float(seg[0].item())
    else:
        total_segments = float(cnt["live_segments"])

    # configs[3]'s shape with a FIXED total of samples split over the ranks (strong scaling), one film sum-reduce
    config4 = None
    if not args.no_extra:
        sc4 = ta.Scene(0)
        cam4 = ta.camera_default(C4_W, C4_H)
        tr4 = ta.Tracer(C4_W, C4_H, device=local_rank, stream=work_stream.cuda_stream)
        tr4.upload_scene(sc4.objects_ptr, sc4.size)
        film4 = torch.zeros(C4_W * C4_H * 3, dtype=torch.float32, device=dev)
        tr4.bind_film(film4.data_ptr())
        multi.render_partitioned(tr4, cam4, SEED, C4_TOTAL_ITERS, film4, rank, world, reducer)       # warm-up
        fence()
        tr4.reset_counters()
        t4 = time.perf_counter()
        multi.render_partitioned(tr4, cam4, SEED, C4_TOTAL_ITERS, film4, rank, world, reducer)
        fence()
        dt4 = time.perf_counter() - t4
        c4 = tr4.counters()
        if world > 1:
            cdev = torch.device("cpu") if rehearsal else dev
            v = torch.tensor([dt4], dtype=torch.float64, device=cdev); dist.all_reduce(v, op=dist.ReduceOp.MAX); dt4 = float(v.item())
            s4 = torch.tensor([c4["live_segments"]], dtype=torch.float64, device=cdev); dist.all_reduce(s4, op=dist.ReduceOp.SUM)
            seg4 = float(s4.item())
        else:
            seg4 = float(c4["live_segments"])
        config4 = {"workload": f"cornell_box_{C4_W}x{C4_H}_mode2", "total_iterations": C4_TOTAL_ITERS,
                   "note": "BASELINE.json configs[3] quotes 4096 iterations; this is an eighth of them, split over the GPUs by sample index",
                   "iterations_per_gpu": multi.rank_iterations(C4_TOTAL_ITERS, rank, world)[2], "n_gpus": world, "scaling": "strong",
                   "seconds": round(dt4, 4), "value": round(seg4 / dt4 / 1e6, 1), "unit": "Mseg/s",
                   "film_reduce_bytes": 12 * C4_W * C4_H if world > 1 else 0}
        tr4.close()
        del film4

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_segments / elapsed / 1e6
        # roofline of the dominant kernel (rank 0's launches): algorithmic flop / measured duration
        k_ms = sum(kernel_ms) / len(kernel_ms)
        seg_per_launch = cnt["live_segments"] / args.steps
        paths_per_launch = cnt["paths"] / args.steps
        flop = seg_per_launch * FLOP_PER_SEGMENT + paths_per_launch * FLOP_PER_PATH
        achieved = flop / (k_ms * 1e-3) / 1e12
        kernel_name = {0: "k_bounce<false> (+k_accumulate)", 1: "k_bounce<true> (+k_accumulate)", 2: "k_fused",
                       3: "k_queue (+k_accumulate)", 4: "k_queue (+k_accumulate)"}[args.pipeline]
        out = {
            "metric": "Mpath-segments/s, Cornell box 512x512 path trace (mode 2), 300 iterations",
            "value": round(value, 2),
            "unit": "Mseg/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_min": round(min(step_ms), 4),
            "ms_per_step_median": round(statistics.median(step_ms), 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "cornell_box_512x512_mode2_300it", "width": W, "height": H,
                       "iterations_per_gpu": ITERS, "path_depth": 5, "rng": "philox4x32-10, reference reuse",
                       "pipeline": ["wavefront", "compacted", "fused", "work-queue", "work-queue"][args.pipeline],
                       "parallelism": f"sample-index x{world}, one RCCL film sum-reduce per step" if world > 1 else "single GPU",
                       "live_segments_per_step_per_gpu": int(seg_per_launch),
                       "segments_per_path": round(seg_per_launch / max(paths_per_launch, 1), 4)},
            "roofline": {
                "bound": "valu",
                "kernel": kernel_name,
                "achieved": round(achieved, 3),
                "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                "kernel_ms": round(k_ms, 4),
                "kernel_ms_min": round(min(kernel_ms), 4),
                "kernel_ms_median": round(statistics.median(kernel_ms), 4),
                "flop_per_segment": FLOP_PER_SEGMENT,
                "flop_per_path": FLOP_PER_PATH,
                "traffic": None,
                "traffic_source": None,
                "hbm_algorithmic_GBs": round(seg_per_launch * BYTES_PER_SEGMENT_REF / (k_ms * 1e-3) / 1e9, 2),
                "hbm_peak_GBs": PEAK_HBM_GBS,
            },
        }
        rl = out["roofline"]
        if args.pipeline in (3, 4):
            rl["traffic"], rl["traffic_source"] = measured_hbm_traffic("k_queue", "*_hbm.json")
        if args.pipeline == 1:
            # the per-bounce ray state is the point of this pipeline: its HBM rate against the chip's peak
            t_bytes, src = measured_hbm_traffic(None, "*_compacted_hbm_summary.json", whole_call=True)
            rl["traffic"], rl["traffic_source"] = t_bytes, src
            if t_bytes:
                rl["hbm_achieved_GBs"] = round(t_bytes / (k_ms * 1e-3) / 1e9, 1)
                rl["hbm_frac"] = round(t_bytes / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
        if args.pipeline in (2, 3, 4) and cnt.get("shader_mhz"):
            # The unfused ceiling at the clock the kernel ran at (its own wavefronts measure it), and the north_star's
            # figure of merit: VALU issue utilisation = issue cycles the kernel's VALU instructions need at the SIMD's
            # best rate / the cycles it ran. Instruction count from the committed PMC pass of the same launch
            # (flagged if that profile's segment count is not this run's), clock and duration live.
            mhz = cnt["shader_mhz"]
            rl["shader_mhz"] = round(mhz, 1)
            rl["peak_unfused_at_clock"] = {name: round(unfused_peak_tflops(mhz, c), 2) for name, c in ISSUE_CONSTANTS}
            rl["frac_unfused_at_clock"] = {name: round(achieved / unfused_peak_tflops(mhz, c), 4) for name, c in ISSUE_CONSTANTS}
            rl["issue_constants"] = ISSUE_CONSTANT_SOURCES
            per64, prof_segs, src, prof_digest = measured_valu_mix()
            if per64 and args.pipeline != 2:
                rl["valu_insts_per_64_segments"] = round(per64, 1)
                rl["valu_insts_source"] = src
                # the profile must be of this workload (segment count) and of these kernel sources (digest recorded by
                # tools/pmc_summary.py); otherwise the utilisation below mixes two kernels -- flagged, not hidden
                rl["valu_insts_profile_matches_run"] = bool(prof_segs == int(seg_per_launch) and prof_digest == kernel_source_digest())
                rl["valu_issue_utilisation"] = issue_utilisation(per64 * seg_per_launch / 64.0, k_ms, mhz)
        if world > 1:
            out["multi_gpu"] = {"film_reduce_bytes_per_step": 12 * W * H,
                                "film_reduce_ms_on_root": {"mean": round(sum(reduce_ms) / len(reduce_ms), 4), "min": round(min(reduce_ms), 4),
                                                           "max": round(max(reduce_ms), 4),
                                                           "note": "torch events around dist.reduce on rank 0's stream: the wait for the slowest rank plus the transfer"} if reduce_ms else None,
                                "render_kernel_ms_rank0": round(k_ms, 4),
                                "one_gpu_value_same_workload": round(n1_value, 2) if n1_value else None,
                                "efficiency_vs_n1": round(value / (world * n1_value), 4) if n1_value else None,
                                "efficiency_note": "weak scaling: value / (n_gpus x the rate rank 0 reaches alone on its 300 iterations, measured in this run)"}
        if film_check is not None:
            out["film_check"] = film_check
        if config4 is not None:
            out["config4_strong"] = config4
        if rehearsal:
            out["rehearsal"] = "all ranks on one GPU over gloo: control-flow check only, not a measurement"
        tr.close()
        if world == 1 and not args.no_extra:
            out["extra_configs"] = extra_configs(ta, k_ms)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    else:
        tr.close()

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
