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
followed by the display transform on rank 0 -- the only exchange the path has.

Rank 0 prints one JSON line. Only the `cpu_baseline` leg touches oracle/.
"""
import argparse
import json
import os
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
BYTES_PER_SEGMENT_REF = 108.0  # reference buffer layout, SURVEY.md 8(d)
PEAK_FP32_TFLOPS = 157.3       # MI355X FP32 vector peak (packed FMA), MI355X_MICROARCH.md
PEAK_FP32_UNFUSED_TFLOPS = 65.5  # measured (tools/micro/inst_cost.hip): v_mul_f32 / v_add_f32 issue every ~2.4
                                 # cycles per wave64 per SIMD at 8 waves -> 64 * 1024 SIMDs * 2.4 GHz / 2.4
PEAK_HBM_GBS = 8000.0
N_SIMD = 1024                    # 256 CUs x 4
VALU_ISSUE_CYCLES = 2.17         # measured (tools/micro/clock_probe.hip): shader-clock cycles one SIMD needs per
                                 # wave64 v_mul_f32 / v_add_f32, the cheapest VALU class (v_fma_f32, v_cmp: ~4)


def measured_valu_mix():
    """VALU wave-instructions per 64 live segments of k_queue on this workload, from the latest committed
    rocprofv3 PMC pass (tools/pmc_fused.sh -> profiles/*_pmc_k_queue.json: SQ_INSTS_VALU / segments * 64)."""
    import glob
    files = sorted(glob.glob(os.path.join(HERE, "profiles", "*_pmc_k_queue.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return float(d["derived"]["valu_insts_per_64_segments"]), os.path.basename(files[-1])
    except (KeyError, ValueError):
        return None, None


def measured_hbm_traffic():
    """HBM bytes of one k_queue launch of this workload from the committed rocprofv3 PMC passes
    (tools/pmc_hbm.sh -> profiles/*_hbm.json: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE)."""
    import glob
    files = sorted(glob.glob(os.path.join(HERE, "profiles", "*_hbm.json")))
    if not files:
        return None, None
    try:
        per = json.load(open(files[-1]))["per_kernel"]
        k = next(v for name, v in per.items() if "k_queue" in name)
        return int((k["hbm_read_MB_gfx950_corrected_x2"] + k["hbm_write_MB"]) * 1048576), os.path.basename(files[-1])
    except (KeyError, ValueError, StopIteration):
        return None, None


def measured_hbm_traffic_compacted():
    """HBM bytes of one 300-iteration call of the compacted wavefront pipeline (every kernel and fill of the call)
    from the committed PMC passes (PIPE=1 tools/pmc_hbm.sh -> profiles/*_compacted_hbm_summary.json)."""
    import glob
    files = sorted(glob.glob(os.path.join(HERE, "profiles", "*_compacted_hbm_summary.json")))
    if not files:
        return None, None
    try:
        per = json.load(open(files[-1]))["per_kernel"]
        mb = sum(v["hbm_read_MB_gfx950_corrected_x2"] + v["hbm_write_MB"] for k, v in per.items() if "tonemap" not in k and "copyBuffer" not in k)
        return int(mb * 1048576), os.path.basename(files[-1])
    except (KeyError, ValueError):
        return None, None


def cpu_baseline():
    """Time the CPU oracle (plain-C restatement, bit-identical to the reference host build)
    on this box's host cores over a bounded sample of the same workload."""
    from oracle.pyoracle import Oracle, available
    o = Oracle("c")
    o.set_math_mode(Oracle.MATH_LIBM)
    s = o.scene_create(0)
    cam = o.camera_default(W, H)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    out = {}
    # single thread: 8 iterations of the 512x512 workload
    t = time.perf_counter()
    r = o.pathtrace(s, W, H, cam, SEED, 1, 8, want_pbo=False, nthreads=1)
    dt1 = time.perf_counter() - t
    v1 = r["stats"]["live_segments"] / dt1 / 1e6
    # all threads: 48 iterations
    n_mt = 48
    t = time.perf_counter()
    r = o.pathtrace(s, W, H, cam, SEED, 1, n_mt, want_pbo=False, nthreads=threads)
    dtm = time.perf_counter() - t
    vm = r["stats"]["live_segments"] / dtm / 1e6
    out = {"value": round(vm, 3), "unit": "Mseg/s", "cores": threads, "kind": "port",
           "sample": f"cornell {W}x{H} mode 2, {n_mt} of {ITERS} iterations, OpenMP over pixels "
                     f"({threads} threads); single thread: 8 iterations",
           "single_thread_value": round(v1, 3)}
    if available("ref"):
        # the reference's own .inl + GLM host build (single-threaded by construction)
        R = Oracle("ref")
        sr = R.scene_create(0, data_dir=os.path.join(HERE, "tracer_amd", "data"))
        t = time.perf_counter()
        rr = R.pathtrace(sr, W, H, cam, SEED, 1, 8, want_pbo=False)
        dtr = time.perf_counter() - t
        out["reference_host_build_single_thread_value"] = round(rr["stats"]["live_segments"] / dtr / 1e6, 3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pipeline", type=int, default=3, help="0 wavefront, 1 compacted, 2 fused, 3 work queue (default), 4 auto (= 3 on this scene)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import tracer_amd as ta

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

    # scene + camera through the product's own host API (no oracle here)
    scene = ta.Scene(0)
    cam = ta.camera_default(W, H)
    # ONE stream for everything: torch's zero_(), the tracer's launches and the RCCL reduce are then ordered by
    # the stream itself. (torch's default stream has handle 0, which trc_create takes as "make a private
    # stream" -- the film sum would not wait for the render.)
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    assert work_stream.cuda_stream != 0
    tr = ta.Tracer(W, H, device=local_rank, stream=work_stream.cuda_stream)
    tr.set_pipeline(args.pipeline)
    tr.upload_scene(scene.objects_ptr, scene.size)
    film = torch.zeros(W * H * 3, dtype=torch.float32, device=dev)   # torch owns the film so RCCL can reduce it
    tr.bind_film(film.data_ptr())

    from tracer_amd import multi

    class _HostReduce:
        """Rehearsal only (all ranks on one GPU): RCCL refuses two ranks on one device, so the sum goes over gloo
        through the host. Same interface as torch.distributed for multi.reduce_film."""
        @staticmethod
        def is_available(): return True
        @staticmethod
        def is_initialized(): return True
        @staticmethod
        def get_world_size(): return world
        ReduceOp = dist.ReduceOp
        @staticmethod
        def reduce(t, dst=0, op=None):
            host = t.cpu()
            dist.reduce(host, dst=dst, op=op)
            if rank == dst:
                t.copy_(host)

    def step():
        # the product's own multi-GPU entry: zero the film, render iterations rank+1, rank+1+world, ... of
        # ITERS x world, ONE sum-reduce to rank 0, display transform there (tracer_amd/multi.py)
        multi.render_partitioned(tr, cam, SEED, ITERS * world, film, rank, world, _HostReduce if rehearsal else dist)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    tr.reset_counters()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # HIP events recorded on the launch stream around the path kernels of this step
        # (read after the fence below would only see the last step, so query per step; the
        # query waits for the kernel, which the next step's film.zero_() would do anyway)
        kernel_ms.append(tr.last_kernel_ms())
    fence()
    elapsed = time.perf_counter() - t0
    cnt = tr.counters()

    # Untimed self-check of the multi-GPU path: the reduced film holds world x ITERS samples per pixel, so its
    # mean per sample must agree with the mean of rank 0's own ITERS samples (a sum that did not wait for a
    # rank's render, or lost a rank, would not).
    film_check = None
    if world > 1:
        step(); torch.cuda.synchronize()
        mean_all = float(film.sum().item()) / (3.0 * W * H * ITERS * world) if rank == 0 else 0.0
        film.zero_(); tr.pathtrace(cam, SEED, rank + 1, ITERS, iter_stride=world); torch.cuda.synchronize()
        mean_own = float(film.sum().item()) / (3.0 * W * H * ITERS)
        if rank == 0:
            film_check = {"mean_radiance_all_ranks": round(mean_all, 6), "mean_radiance_rank0": round(mean_own, 6),
                          "ok": bool(abs(mean_all - mean_own) <= 0.01 * mean_own)}

    if world > 1:
        cdev = torch.device("cpu") if rehearsal else dev
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        seg = torch.tensor([cnt["live_segments"], cnt["paths"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(seg, op=dist.ReduceOp.SUM)
        total_segments, total_paths = float(seg[0].item()), float(seg[1].item())
    else:
        total_segments, total_paths = float(cnt["live_segments"]), float(cnt["paths"])

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_segments / elapsed / 1e6
        # roofline of the dominant kernel (rank 0's launches): algorithmic flop / measured duration
        k_ms = sum(kernel_ms) / len(kernel_ms)
        seg_per_launch = cnt["live_segments"] / args.steps
        paths_per_launch = cnt["paths"] / args.steps
        flop = seg_per_launch * FLOP_PER_SEGMENT + paths_per_launch * FLOP_PER_PATH
        achieved = flop / (k_ms * 1e-3) / 1e12
        kernel_name = {0: "k_bounce<false> (+k_raygen)", 1: "k_bounce<true> (+k_raygen, k_accumulate)", 2: "k_fused",
                       3: "k_queue (+k_accumulate)", 4: "k_queue (+k_accumulate)"}[args.pipeline]
        out = {
            "metric": "Mpath-segments/s, Cornell box 512x512 path trace (mode 2), 300 iterations",
            "value": round(value, 2),
            "unit": "Mseg/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
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
                "peak_unfused_measured": PEAK_FP32_UNFUSED_TFLOPS,
                "frac_unfused_measured": round(achieved / PEAK_FP32_UNFUSED_TFLOPS, 4),
                "peak_unfused_nominal": 39.3,      # 16 lanes/SIMD/clock
                "frac_unfused_nominal": round(achieved / 39.3, 4),
                "kernel_ms": round(k_ms, 4),
                "flop_per_segment": FLOP_PER_SEGMENT,
                "flop_per_path": FLOP_PER_PATH,
                "traffic": None,
                "traffic_source": None,
                "hbm_algorithmic_GBs": round(seg_per_launch * BYTES_PER_SEGMENT_REF / (k_ms * 1e-3) / 1e9, 2),
                "hbm_peak_GBs": PEAK_HBM_GBS,
            },
        }
        if args.pipeline in (3, 4):
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured_hbm_traffic()
        if args.pipeline == 1:
            # the per-bounce ray state is the point of this pipeline: its HBM rate against the chip's peak
            t_bytes, src = measured_hbm_traffic_compacted()
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = t_bytes, src
            if t_bytes:
                out["roofline"]["hbm_achieved_GBs"] = round(t_bytes / (k_ms * 1e-3) / 1e9, 1)
                out["roofline"]["hbm_frac"] = round(t_bytes / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
        if args.pipeline in (2, 3, 4) and cnt.get("shader_mhz"):
            # FP32 VALU utilisation (the north_star's figure of merit): issue cycles the kernel's VALU
            # instructions need at the best rate the SIMD has, over the cycles it ran -- instruction count
            # from the committed PMC pass, shader clock measured live by the kernel's own wavefronts.
            mhz = cnt["shader_mhz"]
            out["roofline"]["shader_mhz"] = round(mhz, 1)
            out["roofline"]["peak_unfused_at_clock"] = round(64 * N_SIMD * mhz * 1e6 / VALU_ISSUE_CYCLES / 1e12, 2)
            out["roofline"]["frac_unfused_at_clock"] = round(achieved / out["roofline"]["peak_unfused_at_clock"], 4)
            per64, src = measured_valu_mix()
            if per64 and args.pipeline != 2:
                insts_per_simd = per64 * seg_per_launch / 64.0 / N_SIMD
                cycles = k_ms * 1e-3 * mhz * 1e6
                out["roofline"]["valu_insts_per_64_segments"] = per64
                out["roofline"]["valu_insts_source"] = src
                out["roofline"]["valu_issue_utilisation"] = round(insts_per_simd * VALU_ISSUE_CYCLES / cycles, 4)
        if film_check is not None:
            out["film_check"] = film_check
        if rehearsal:
            out["rehearsal"] = "all ranks on one GPU over gloo: control-flow check only, not a measurement"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)

    tr.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
