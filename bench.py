#!/usr/bin/env python3
"""bench.py — tracked frames/sec of the HIP hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1 with no RANK in the environment: this process starts the N rank processes itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), relays
rank 0's JSON line and exits with the children's status; it never imports torch or touches HIP.
Launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE set) it is one rank.

A step is one pass of the hot path over one batch of synthetic input: every one of the B streams of
a GPU gets its next 1080p NV12 frame and produces one vt_result (NV12 window -> crop / resize /
normalise -> patch embed -> 12-block joint encoder -> centre head -> box decode, all inside
libvittrack_hip.so).

`value` ("timing includes the H2D copy", /root/reference/src/pipeline.rs:95-112:
the host maps the buffer, then tracks): the frames live in ordinary HOST memory and go through the
product's own ingest, vt_group_enqueue_host / vt_group_wait_next - each stream's search window is
packed into a pinned arena and crosses PCIe on a copy stream while the previous pass runs.
`device_only` (frames already resident in HBM, vt_group_enqueue_device) and `full_frame` (every
3.1 MB frame copied whole) are measured beside it in the same run. value = tracked frames of ALL
ranks / max-over-ranks wall time. Streams are independent (one frame chain each); ranks exchange
nothing per frame - the only collective is the start-up weight broadcast (RCCL), outside the timed
region, recorded under `collective`.

Also on the JSON line: "roofline" for the dominant kernel (HIP events on the library's own stream,
algorithmic FLOPs / measured time against the 2.5 PFLOP/s dense bf16 MFMA peak), "byte_kernels"
(HBM-bound kernels against 8 TB/s) and "cpu_baseline" (float32 torch-CPU forward on all host cores,
oracle/cpu_fp32.py, rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json: metric is quoted on ViT-B/16 384x192 (configs[2]); cfg2 = configs[1]
    "cfg3": ("vitb16_t192_s384", "1080p NV12, ViT-B/16 template192/search384 (OSTrack-384 shape)"),
    "cfg2": ("vitb16_t128_s256", "1080p NV12, ViT-B/16 template128/search256"),
    "cfg5": ("vitl14_t196_s392", "4K NV12, ViT-L/14 template196/search392"),
    "tiny": ("tiny_t64_s128", "640x480 NV12, tiny test model (not a BASELINE config)"),
}
PEAK_BF16_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
PEAK_HBM_GBS = 8000.0
PCIE_LINK_GBS = 63.0        # MI355X_MICROARCH.md "Host link": PCIe Gen5 x16 (spec)


def iou(a, b):
    ix = max(0, min(a[0] + a[2], b[0] + b[2]) - max(a[0], b[0]))
    iy = max(0, min(a[1] + a[3], b[1] + b[3]) - max(a[1], b[1]))
    u = a[2] * a[3] + b[2] * b[3] - ix * iy
    return ix * iy / u if u > 0 else 0.0


def usable_cores() -> int:
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def baseline_metric() -> str:
    """BASELINE.json's metric string, verbatim (it travels with the repository snapshot)"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "tracked frames/sec @1080p ViT-B/16 384×192, 1 GPU; + MFMA roofline %"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0,
                    help="independent tracked streams per GPU (0: groups x weights.recommended_streams). "
                         "BASELINE.json configs[3] taken literally (8 independent streams, ONE per GPU) is "
                         "`--gpus 8 --streams 1 --groups 1`; the line then carries config.cfg4_literal = true. The "
                         "default keeps 60 streams per GPU (weak scaling of the batched engines)")
    ap.add_argument("--groups", type=int, default=2,
                    help="engines per GPU, each with streams/groups streams on its own HIP stream "
                         "(kernels of different groups overlap on the chip)")
    ap.add_argument("--engines", default="",
                    help="engine sizes instead of --groups equal ones: '30+1', or 'auto' = "
                         "vt_plan_engines(--streams)")
    ap.add_argument("--ring", type=int, default=64, help="distinct frames of the clip (host memory; HBM for device_only)")
    ap.add_argument("--ingest", default="host", choices=("host", "device"),
                    help="what `value` times: host = frames in host memory through vt_group_enqueue_host "
                         "(H2D-inclusive, SURVEY 8(d)); device = frames resident in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the secondary ingest legs")
    ap.add_argument("--no-single-leg", action="store_true", help="skip the one-tracker-per-process leg")
    ap.add_argument("--host-steps", type=int, default=100)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--eager", action="store_true", help="no hipGraph replay")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch only: rendezvous port (0: a free one)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: ranks rendezvous over gloo on the CPU, broadcast the weight blob, plan "
                         "their streams and aggregate synthetic timings (exercises the launcher, the collective "
                         "record and the aggregation; the JSON line says dry_run)")
    return ap.parse_args(argv)


# ---- N > 1 without a launcher: start the ranks ourselves (no torch, no HIP in this process) ----------

def self_launch(n: int, argv: list, port: int = 0, timeout_s: float | None = None) -> int:
    """Start `python -m torch.distributed.run --nnodes=1 --nproc-per-node n bench.py <argv>` as a CHILD
    process (the parent never initialises the GPU: no exec, no torch import), stream its stdout
    through, and return its exit status. Rank 0 prints the single JSON line; everything the children
    write to stderr stays on stderr."""
    if port <= 0:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    try:
        out, _ = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
        sys.stdout.write(out or "")
        return 124
    sys.stdout.write(out or "")
    sys.stdout.flush()
    return proc.returncode


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus, argv, args.master_port))
    run_rank(args)


def pin_rank(local: int, world: int) -> dict:
    """CPU placement of this rank BEFORE numpy / torch / HIP load (their threads inherit the mask): the CPUs of the
    GPU's NUMA node, sliced between the ranks that share it (gstreamer-vit-tracker_amd/placement.py, loaded by path so
    that nothing else is imported first)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_vt_placement", os.path.join(ROOT, "gstreamer-vit-tracker_amd", "placement.py"))
    try:            # placement is an optimisation: nothing it meets on a machine may stop the bench
        pl = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(pl)
        return pl.apply(local, world)
    except Exception as e:
        return {"cpus": "", "n": 0, "source": f"placement failed, mask left as given: {e!r}"}


def run_rank(args):
    # dmabuf IPC only on this pool: RCCL's (and torch's) cross-process buffer sharing fails with
    # hipIpcGetMemHandle otherwise. Set before torch / HIP load, so that ranks started by an external
    # `python -m torch.distributed.run ... bench.py --gpus N` get it exactly like the self-launched ones.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    affinity = pin_rank(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))
    import numpy as np
    import torch
    import gstreamer_vit_tracker_amd as vt
    from gstreamer_vit_tracker_amd import distributed as vd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: run `python bench.py --gpus N` (it starts "
                         "its own ranks) or python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    cfg_name, wl_text = WORKLOADS[args.workload]
    if args.dry_run:
        return dry_run(args, vt, vd, world, rank, cfg_name, wl_text, affinity)
    have = torch.cuda.device_count()
    if have < world:
        raise SystemExit(f"bench.py --gpus {world}: rank {rank} sees {have} GPU(s); this run needs {world} devices "
                         "(one process per GPU)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    G = args.groups
    # default: every engine gets the batch that fills the 256 CUs in whole GEMM rounds (30 for cfg3)
    B = args.streams if args.streams > 0 else G * vt.recommended_streams(vt.model_info_for(cfg_name))
    K, W, R = args.steps, args.warmup, args.ring
    fw, fh, sq = (3840, 2160, 160) if args.workload == "cfg5" else \
        ((640, 480, 64) if args.workload == "tiny" else (1920, 1080, 64))

    # ---- weights: rank 0 generates/reads the blob, RCCL broadcast, every rank builds from HBM ----
    wpath = vt.weights.ensure_weights(cfg_name) if rank == 0 else None
    if args.engines == "auto":
        if args.streams <= 0:
            raise SystemExit("--engines auto needs --streams")
        sizes = vt.plan_engines(vt.model_info_for(cfg_name), args.streams)      # the C ABI's vt_plan_engines
    elif args.engines:
        sizes = [int(x) for x in args.engines.split("+")]
    else:
        if B % G:
            raise SystemExit("--streams must be a multiple of --groups (or give --engines)")
        sizes = [B // G] * G
    G, B = len(sizes), sum(sizes)
    off = [sum(sizes[:g]) for g in range(G + 1)]         # engine g holds streams off[g] .. off[g+1]
    eng = [g for g in range(G) for _ in range(sizes[g])]  # stream -> engine
    Bg = sizes[0]                                          # the engine whose pass is profiled below
    collective = {"backend": "none", "world_size": 1, "broadcast_bytes": 0, "broadcast_ms": 0.0}
    if world > 1:
        blob, collective = vd.timed_broadcast_weights(wpath, device=dev)
        grps = [vt.Group(n_streams=b, device=local, use_graph=not args.eager,
                         device_blob=(blob.data_ptr(), blob.numel())) for b in sizes]
        del blob
    else:
        grps = [vt.Group(wpath, n_streams=b, device=local, use_graph=not args.eager) for b in sizes]
    grp = grps[0]
    mi = grp.model_info()

    # ---- synthetic clip: R frames of a closed path; stream i runs it with a phase --------------------
    # The clip lives in host memory (pinned: the full-frame leg copies from it) and, for device_only, in HBM.
    plan = vd.plan_rank(rank, world, B, R)
    sc = vt.synth.MovingSquare(fw, fh, sq, seed=plan["clip_seed"], path="circle", period=R,
                               amp=3.8 * R / (2 * np.pi))
    host_t = torch.from_numpy(np.stack([sc.frame_nv12(t) for t in range(R)])).pin_memory()
    host = host_t.numpy()
    clip = host_t.to(dev)
    fbytes = host.shape[1]
    base = clip.data_ptr()
    phase = plan["phase"]
    frames_at = []
    for t in range(R):
        frames_at.append([vt.frame_nv12(base + ((t + phase[i]) % R) * fbytes,
                                        base + ((t + phase[i]) % R) * fbytes + fw * fh, fw, fh)
                          for i in range(B)])
    hclip = [vt.NV12Frame(host[t], fw, fh) for t in range(R)]      # the same frames as host buffers

    def host_frames_for(gi, t):
        return [hclip[(t + phase[i]) % R] for i in range(off[gi], off[gi + 1])]

    def reinit_device():
        for i in range(B):
            grps[eng[i]].init_device(i - off[eng[i]], frames_at[0][i], vt.BBox.new(*sc.gt_box(phase[i])))

    def reinit_host():
        for i in range(B):
            grps[eng[i]].init_host(i - off[eng[i]], hclip[phase[i] % R], vt.BBox.new(*sc.gt_box(phase[i])))

    def barrier():
        if world > 1:
            dist.barrier()

    def check_tracking(res, t_last, n_updates):
        """every stream really tracked every frame, and still sits on the square"""
        ious, ok = [], True
        for i in range(B):
            st = grps[eng[i]].read_state(i - off[eng[i]])
            ok = ok and st["frames_done"] == n_updates and st["success_count"] == n_updates
            ious.append(iou(res[i].bbox, sc.gt_box((t_last + phase[i]) % R)))
        return bool(ok and min(ious) > 0.5), float(min(ious))

    # ---- the two timed loops: exactly K steps after W untimed ones, barrier + synchronize on both sides ----
    def timed_device(K_, W_, fa=None):
        """frames resident in HBM (or, fa given, wherever its descriptors point): vt_group_enqueue_device,
        one host thread"""
        fa = frames_at if fa is None else fa
        for i in range(B):
            grps[eng[i]].init_device(i - off[eng[i]], fa[0][i], vt.BBox.new(*sc.gt_box(phase[i])))

        def enqueue_all(t):
            fr = fa[t % R]
            for g in range(G):
                grps[g].enqueue_device(fr[off[g]:off[g + 1]])

        def wait_all():
            r = []
            for g in range(G):
                r += grps[g].wait()
            return r

        for t in range(1, W_ + 1):
            enqueue_all(t)
        wait_all()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(W_ + 1, W_ + K_ + 1):
            enqueue_all(t)
        res = wait_all()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        ok, miou = check_tracking(res, W_ + K_, W_ + K_)
        return dt, ok, miou, enqueue_all, wait_all

    def timed_host(K_, W_, pipelined=True):
        """frames in host memory through the product's ingest; one host thread per engine (as a host
        with one capture thread per engine would), all released together by a barrier"""
        import threading
        reinit_host()
        results = [None] * G
        oks = [True] * G
        start = threading.Barrier(G + 1)
        stop = threading.Barrier(G + 1)

        def run(gi, first, n):
            ok, r = True, None
            g_ = grps[gi]
            if pipelined:     # upload of step t+1 (copy stream) overlaps the pass of step t
                g_.enqueue_host(host_frames_for(gi, first))
                for t in range(first + 1, first + n):
                    g_.enqueue_host(host_frames_for(gi, t))
                    r = g_.wait_next()
                    ok = ok and all(x.success for x in r)
                r = g_.wait_next()
                ok = ok and all(x.success for x in r)
            else:
                for t in range(first, first + n):
                    r = g_.update_host(host_frames_for(gi, t))
                    ok = ok and all(x.success for x in r)
            return ok, r

        errs = []

        def worker(gi):
            try:
                if W_ > 0:
                    run(gi, 1, W_)
                start.wait()
                oks[gi], results[gi] = run(gi, W_ + 1, K_)
                stop.wait()
            except threading.BrokenBarrierError:
                pass
            except BaseException as e:          # a failing engine must not leave the other threads at a barrier
                errs.append(e)
                start.abort()
                stop.abort()

        th = [threading.Thread(target=worker, args=(gi,)) for gi in range(G)]
        [x.start() for x in th]
        # the workers are in their warm-up; join them at the start line, then open the timed region
        while start.n_waiting < G and not errs:
            time.sleep(0.0005)
        try:
            if errs:
                raise errs[0]
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            start.wait()
            stop.wait()
            torch.cuda.synchronize()
            barrier()
            dt = time.perf_counter() - t0
        except threading.BrokenBarrierError:
            [x.join() for x in th]
            raise errs[0] if errs else RuntimeError("host ingest worker failed")
        [x.join() for x in th]
        res = [r for g in range(G) for r in results[g]]
        ok, miou = check_tracking(res, W_ + K_, W_ + K_)
        return dt, bool(ok and all(oks)), miou

    # ---- headline ------------------------------------------------------------------------------------
    if args.ingest == "host":
        dt_local, tracked_ok, miou = timed_host(K, W)
        ingest_text = ("host memory -> vt_group_enqueue_host / vt_group_wait_next: every stream's search window "
                       "(speculative: 1.75x the crop side) packed into one of two pinned arenas, one H2D copy per "
                       "engine and step on a copy stream, overlapping the previous pass (H2D-inclusive)")
    else:
        dt_local, tracked_ok, miou, _, _ = timed_device(K, W)
        ingest_text = "frames resident in HBM (vt_group_enqueue_device)"
    agg = vd.aggregate_throughput(B * K, dt_local, device=dev if world > 1 else "cpu")
    dt, total_frames, fps = agg["seconds"], agg["frames"], agg["frames_per_s"]
    per_rank = vd.gather_per_rank(B * K / dt_local, device=dev if world > 1 else "cpu")
    collective["world_size"] = dist.get_world_size() if world > 1 else 1
    redos = int(sum(g_.host_redos() for g_ in grps))

    out = {
        "metric": baseline_metric(),
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": wl_text, "vit_config": cfg_name, "frame": f"{fw}x{fh} NV12",
                   "streams_per_gpu": B, "engines_per_gpu": G, "engine_sizes": sizes, "tokens": mi.tokens_template + mi.tokens_search,
                   # BASELINE.json configs[3] taken literally: 8 independent streams, ONE per GPU, on 8 GPUs;
                   # cfg4_shape: the per-GPU shape of that configuration (one tracker on the GPU) at any N
                   "cfg4_literal": bool(B == 1 and G == 1 and world == 8), "cfg4_shape": bool(B == 1 and G == 1),
                   "cpu_affinity": affinity,
                   "ingest": ingest_text, "launch": "eager" if args.eager else "hipGraph",
                   "weights": "synthetic seeded encoder + fitted head (no reference weights exist)"},
        "per_stream_fps": fps / (world * B),
        "per_rank_fps": per_rank,
        "collective": collective,
        "tracked_ok": bool(tracked_ok), "min_iou_vs_truth": miou, "redone_passes": redos,
        "gflop_per_frame": mi.flops_per_frame / 1e9,
        "encoder_gflop_per_frame": mi.encoder_flops_per_frame / 1e9,
        "whole_frame_mfma_frac": fps / world * mi.flops_per_frame / 1e12 / PEAK_BF16_TFLOPS,
    }

    # ---- device_only: the same engines on frames already resident in HBM (today's `value` until round 3) ----
    if args.ingest == "host":
        ddt, dok, dmiou, enqueue_all, wait_all = timed_device(K, W)
        dagg = vd.aggregate_throughput(B * K, ddt, device=dev if world > 1 else "cpu")
        out["device_only"] = {
            "value": dagg["frames_per_s"], "unit": "frames/s", "ms_per_step": dagg["seconds"] / K * 1e3,
            "tracked_ok": dok, "min_iou_vs_truth": dmiou,
            "ingest": "frames resident in HBM (vt_group_enqueue_device)",
            "whole_frame_mfma_frac": dagg["frames_per_s"] / world * mi.flops_per_frame / 1e12 / PEAK_BF16_TFLOPS,
            "headline_vs_device_only": fps / dagg["frames_per_s"]}
    else:
        _, _, _, enqueue_all, wait_all = timed_device(0, 2)

    # ---- synchronous single-call latency (the literal drop-in call pattern), device-resident frames ------
    lat = []
    for t in range(W + K + 1, W + K + 101):
        a = time.perf_counter()
        enqueue_all(t)
        wait_all()
        lat.append(time.perf_counter() - a)
    lat_ms = float(np.median(lat) * 1e3)
    lat_p99 = float(np.percentile(lat, 99) * 1e3)
    # Reference-style reporting (/root/reference/src/timing_stats.rs:18-46, fed at src/pipeline.rs:104-122):
    # a ring of the last 120 probe-to-probe intervals in integer microseconds, fps = 1e6 / mean. In
    # this synchronous leg every stream completes one frame per call, so a stream's probe-to-probe
    # interval is the call-to-call time and its track time (`trk:` in the reference overlay) the call.
    iv = [int(round(x * 1e6)) for x in lat][-120:]
    ref_fps_stream = 1e6 / (sum(iv) / len(iv)) if sum(iv) else 0.0
    out["sync_update_latency_ms"] = lat_ms
    out["sync_update_latency_p99_ms"] = lat_p99
    out["reference_style"] = {   # src/timing_stats.rs semantics, synchronous calls (one frame per stream per call)
        "per_stream_fps": ref_fps_stream, "aggregate_fps": ref_fps_stream * B * world,
        "window": len(iv), "frame_latency_ms_p50": lat_ms, "frame_latency_ms_p99": lat_p99,
        "avg_track_ms": sum(iv) / len(iv) / 1e3}

    # ---- per-kernel HIP-event timing on the library's stream -> roofline of the dominant kernel ------
    prof = []
    if not args.no_profile and rank == 0:
        prof = grp.profile_device(frames_at[(W + K) % R][:Bg], iters=5)
        tot = sum(p["ms"] for p in prof)
        dom = max(prof, key=lambda p: p["ms"])
        gemm_ms = sum(p["ms"] for p in prof if p["name"].startswith("gemm"))
        gemm_fl = sum(p["flops"] for p in prof if p["name"].startswith("gemm"))
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["flops"] > 0 else 0.0
        # HBM/fabric bytes per launch of the dominant kernel: PMC passes cannot run inside this process
        # (rocprofv3 --pmc is a separate run, one counter group per pass), so the number comes from the
        # committed summary of those passes on the same kernel and shape, with its provenance
        # ... and ONLY while the kernel that is running is the kernel that was counted: the summary records the
        # build identity of the dominant kernel's translation unit (vt_build_info(): sha256 over its sources and
        # compile flags, stamped into the library by build.py); any other build gets traffic = null
        traffic, traffic_src = None, None
        build_id = vt.build_info()
        pmc_path = os.path.join(ROOT, "profiles", "r06_dominant_kernel_pmc.json")
        if os.path.exists(pmc_path):
            try:
                pm = json.load(open(pmc_path))
                if pm.get("kernel_family") == dom["name"] and pm.get("streams_per_pass") == Bg:
                    if pm.get("kernel_source_sha256") and pm.get("kernel_source_sha256") == build_id.get("k_gemm256"):
                        traffic = pm["traffic_bytes_per_launch"]
                        traffic_src = pm["provenance"]
                    else:
                        traffic_src = (f"null: profiles/r06_dominant_kernel_pmc.json was collected on kernel build "
                                       f"{str(pm.get('kernel_source_sha256'))[:12]}, this library is {str(build_id.get('k_gemm256'))[:12]}")
            except Exception:
                pass
        # the same kernel inside the timed region: it shares the chip with the other engine's kernels there, so
        # its flops of a step over its share (eager pass) of the measured step time
        in_run = None
        if dom["flops"] > 0 and tot > 0:
            in_run = G * dom["flops"] / ((dt / K) * (dom["ms"] / tot)) / 1e12 / PEAK_BF16_TFLOPS
        out["roofline"] = {
            "bound": "mfma", "kernel": dom["name"], "achieved": ach, "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
            "traffic_source": traffic_src, "kernel_build": build_id.get("k_gemm256"),
            "in_run_frac": in_run,
            "in_run_source": "the dominant kernel's flops of one timed step (all engines) / (measured ms_per_step x the kernel's "
                             "share of the eager instrumented pass) / peak: what the kernel delivers while it shares the chip "
                             "with the other engine's kernels - `frac` is the kernel alone on the chip",
            "source": "HIP events bound to every dispatch (hipExtLaunchKernelGGL start / stop events) of ONE engine's "
                      "eager instrumented pass (vt_group_profile_device, 5 passes) on the library's own stream, right "
                      "after the timed region - not the timed hipGraph replays. They read 3-4 us per launch ABOVE "
                      "the begin -> end durations rocprofv3's kernel trace gives for the same dispatches (same process, "
                      "profiles/r05_trace_cfg3_30x1_bench_line.json against r05_bench_cfg3_30x1_kernel_stats.csv: 97.3 "
                      "against 93.6 us): `achieved` is the conservative one of the two",
            "launches_per_step": dom["launches"],
            "avg_launch_us": dom["ms"] / max(dom["launches"], 1) * 1e3,
            "flops_per_launch": dom["flops"] / max(dom["launches"], 1),
            "all_gemm_tflops": gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0,
        }
        out["kernels"] = [{"name": p["name"], "launches": p["launches"], "ms": round(p["ms"], 4),
                           "share": round(p["ms"] / tot, 4),
                           "tflops": round(p["flops"] / (p["ms"] * 1e-3) / 1e12, 2) if p["ms"] > 0 else 0}
                          for p in sorted(prof, key=lambda p: -p["ms"])]
        out["eager_event_ms_per_step"] = tot

    # ---- secondary ingest legs (rank 0, N = 1) ----------------------------------------------------------
    if not args.no_host_leg and rank == 0 and world == 1:
        hs = args.host_steps
        # (a) the synchronous form of the headline ingest (vt_group_update_host): no overlap
        sdt, sok, _ = timed_host(hs, 3, pipelined=False)
        out["host_synchronous"] = {"value": B * hs / sdt, "unit": "frames/s", "tracked_ok": sok,
                                   "ingest": "vt_group_update_host: pack + H2D + pass, one call per step"}
        if args.ingest != "host":
            hdt, hok, _ = timed_host(hs, 3, pipelined=True)
            out["pcie_inclusive"] = {"value": B * hs / hdt, "unit": "frames/s", "tracked_ok": hok}
        # (b) full frames: every stream's WHOLE NV12 frame crosses PCIe every step, what a host that maps
        # the whole buffer pays (/root/reference/src/pipeline.rs:95-106). The streams of an engine sit at
        # consecutive clip positions (plan_rank), so their frames are ONE contiguous range of the pinned
        # ring (two when it wraps): one hipMemcpyAsync per engine and step, double-buffered on one shared
        # copy stream so that the upload of step t+1 runs under the pass of step t.
        es = [torch.cuda.ExternalStream(grps[g].hip_stream(), device=dev) for g in range(G)]
        cs1 = torch.cuda.Stream(device=dev)
        dbuf = [[torch.empty((sizes[g], fbytes), dtype=torch.uint8, device=dev) for _ in range(2)] for g in range(G)]
        up = [[torch.cuda.Event() for _ in range(2)] for g in range(G)]
        done = [[torch.cuda.Event() for _ in range(2)] for g in range(G)]
        contiguous = all(phase[i + 1] == phase[i] + 1 for g in range(G) for i in range(off[g], off[g + 1] - 1))
        reinit_device()
        ncopies = [0]

        def full_step(t):
            for g in range(G):
                k = t & 1
                with torch.cuda.stream(cs1):
                    cs1.wait_event(done[g][k])                           # the pass that read this buffer is over
                    if contiguous:
                        first = (t + phase[off[g]]) % R
                        n1 = min(sizes[g], R - first)
                        dbuf[g][k][:n1].copy_(host_t[first:first + n1], non_blocking=True)
                        ncopies[0] += 1
                        if n1 < sizes[g]:
                            dbuf[g][k][n1:].copy_(host_t[:sizes[g] - n1], non_blocking=True)
                            ncopies[0] += 1
                    else:
                        for j, i in enumerate(range(off[g], off[g + 1])):
                            dbuf[g][k][j].copy_(host_t[(t + phase[i]) % R], non_blocking=True)
                            ncopies[0] += 1
                    up[g][k].record(cs1)
                es[g].wait_event(up[g][k])
                base_g = dbuf[g][k].data_ptr()
                grps[g].enqueue_device([vt.frame_nv12(base_g + j * fbytes, base_g + j * fbytes + fw * fh, fw, fh)
                                        for j in range(sizes[g])])
                done[g][k].record(es[g])

        for t in range(1, 4):
            full_step(t)
        wait_all()
        torch.cuda.synchronize()
        ncopies[0] = 0
        f0 = time.perf_counter()
        for t in range(4, 4 + hs):
            full_step(t)
        fres = wait_all()
        torch.cuda.synchronize()
        fdt = time.perf_counter() - f0
        f_ok = all(r.success for r in fres) and min(iou(fres[i].bbox, sc.gt_box((3 + hs + phase[i]) % R)) for i in range(B)) > 0.5
        ffps = B * hs / fdt
        out["full_frame"] = {
            "value": ffps, "unit": "frames/s", "ms_per_step": fdt / hs * 1e3, "tracked_ok": bool(f_ok),
            "vs_headline": ffps / (fps / world),
            "h2d_GBps": ffps * fbytes / 1e9, "pcie_link_GBps_spec": PCIE_LINK_GBS,
            "h2d_copies_per_step": ncopies[0] / hs,
            "whole_frame_mfma_frac": ffps * mi.flops_per_frame / 1e12 / PEAK_BF16_TFLOPS,
            "ingest": "every stream's whole NV12 frame (%.2f MB) copied from a pinned host ring each step, %s, "
                      "double-buffered per engine on one shared copy stream" %
                      (fbytes / 1e6, "one contiguous copy per engine (two at the ring's wrap)" if contiguous
                       else "one copy per stream")}
        # what the link gives with nothing else running: the same copies, no passes
        torch.cuda.synchronize()
        nb = min(R, 32)
        pure = torch.empty((nb, fbytes), dtype=torch.uint8, device=dev)
        for _ in range(2):
            pure.copy_(host_t[:nb], non_blocking=True)
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        for _ in range(10):
            pure.copy_(host_t[:nb], non_blocking=True)
        torch.cuda.synchronize()
        out["full_frame"]["h2d_alone_GBps"] = 10 * nb * fbytes / (time.perf_counter() - p0) / 1e9
        del dbuf, pure
        # (c) zero copy: the frames stay in (registered) host memory and the pixel kernel reads the pixels it
        # samples over PCIe - vt_host_register once, then host-mapped pointers in vt_frame with the _device
        # entry points. No staging copy, no CPU packing, nothing proportional to the frame size.
        hm = vt.HostMapping(np.array(host))          # a pageable copy: torch's pinned tensor is registered already
        zframes = [[vt.frame_nv12(hm.d_ptr + ((t + phase[i]) % R) * fbytes, hm.d_ptr + ((t + phase[i]) % R) * fbytes + fw * fh,
                                  fw, fh) for i in range(B)] for t in range(R)]
        zdt, zok, zmiou, _, _ = timed_device(hs, 5, fa=zframes)
        out["zero_copy"] = {
            "value": B * hs / zdt, "unit": "frames/s", "ms_per_step": zdt / hs * 1e3, "tracked_ok": zok,
            "min_iou_vs_truth": zmiou, "vs_headline": (B * hs / zdt) / (fps / world),
            "ingest": "whole frames in host memory registered with vt_host_register; vt_group_enqueue_device on the "
                      "host-mapped pointers: the pixel kernel fetches the search windows over PCIe itself"}
        del zframes
        hm.close()
        reinit_device()

    # ---- the literal drop-in case: ONE tracker per process (/root/reference/src/pipeline.rs:55,109-120) ----
    # vt_create + vt_update_nv12 from a host pointer (what the reference's probe would call) and the _device
    # form on HBM-resident frames, synchronous calls, reference-style statistics; per-kernel times of one
    # update. A leg of its own: the headline batches 60 streams, the reference runs one.
    if not args.no_single_leg and rank == 0 and world == 1:
        ntrk = 300
        trk = vt.VitTrack(wpath)
        single = {}
        hm1 = None
        for name in ("host_pointer", "host_pointer_registered", "device_pointer"):
            if name == "host_pointer":
                trk.init(hclip[0], vt.BBox.new(*sc.gt_box(0)))
                call = lambda t: trk.update(hclip[t % R])
            elif name == "host_pointer_registered":
                # the same host-pointer calls on frames inside a range registered with vt_host_register: the library
                # takes the zero-copy route by itself (no window packing, no staging copy)
                reg = np.array(host)
                hm1 = vt.HostMapping(reg)
                rclip = [vt.NV12Frame(reg[t], fw, fh) for t in range(R)]
                trk.init(rclip[0], vt.BBox.new(*sc.gt_box(0)))
                call = lambda t: trk.update(rclip[t % R])
            else:
                trk.init_nv12_device(base, base + fw * fh, fw, fh, fw, fw, vt.BBox.new(*sc.gt_box(0)))
                call = lambda t: trk.update_nv12_device(base + (t % R) * fbytes, base + (t % R) * fbytes + fw * fh, fw, fh, fw, fw)
            for t in range(1, 21):
                call(t)
            lat1, ok1 = [], True
            for t in range(21, 21 + ntrk):
                a0 = time.perf_counter()
                r1 = call(t)
                lat1.append(time.perf_counter() - a0)
                ok1 = ok1 and r1.success and iou(r1.bbox, sc.gt_box(t % R)) > 0.5
            iv1 = [int(round(x * 1e6)) for x in lat1][-120:]
            single[name] = {"updates": ntrk, "fps": ntrk / sum(lat1), "ms_p50": float(np.median(lat1) * 1e3),
                            "ms_p99": float(np.percentile(lat1, 99) * 1e3), "tracked_ok": bool(ok1),
                            "reference_style_fps": 1e6 / (sum(iv1) / len(iv1))}
        prof1 = trk.as_group().profile_device([vt.frame_nv12(base, base + fw * fh, fw, fh)], iters=5)
        tot1 = sum(p["ms"] for p in prof1)
        single["launches_per_update"] = int(sum(p["launches"] for p in prof1))
        single["eager_event_ms_per_update"] = tot1
        single["kernels"] = [{"name": p["name"], "launches": p["launches"], "ms": round(p["ms"], 4),
                              "tflops": round(p["flops"] / (p["ms"] * 1e-3) / 1e12, 2) if p["ms"] > 0 else 0}
                             for p in sorted(prof1, key=lambda p: -p["ms"])]
        out["single_stream"] = single
        del trk
        if hm1 is not None:
            hm1.close()

    # ---- byte-bound kernels against the HBM roofline (north_star: "HBM GB/s against gfx950 peak") ------
    if not args.no_profile and rank == 0:
        bk = []

        def time_converter(cw, ch, nfr, iters):
            """the reference's whole-frame converter through the product ABI, device-resident random frames, torch events
            on the stream the calls are enqueued on; nfr == 0: vt_nv12_to_rgb8_device, else nfr frames per launch"""
            n_ = max(nfr, 1)
            src = torch.randint(0, 256, (n_, cw * ch * 3 // 2), dtype=torch.uint8, device=dev)
            dst = torch.empty((n_, cw * ch * 3), dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream(dev)
            ins, outs, lens = [src[i].data_ptr() for i in range(n_)], [dst[i].data_ptr() for i in range(n_)], [src.shape[1]] * n_

            def once():
                if nfr == 0:
                    vt._check(vt.lib().vt_nv12_to_rgb8_device(local, ins[0], lens[0], cw, ch, outs[0], st.cuda_stream))
                else:
                    vt.nv12_to_rgb8_batch_device(ins, lens, cw, ch, outs, device=local, hip_stream=st.cuda_stream)
            for _ in range(3):
                once()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(iters):
                once()
            e1.record(st)
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e3 / iters

        for (cw, ch, nfr) in ((1920, 1080, 0), (3840, 2160, 0), (1920, 1080, 30), (1920, 1080, 60)):
            us = time_converter(cw, ch, nfr, 50 if nfr == 0 else 20)
            by = cw * ch * 4.5 * max(nfr, 1)                    # 1.5 B read + 3 B written per pixel
            rec = {"kernel": "nv12_to_rgb8_kernel" if nfr == 0 else "nv12_to_rgb8_batch_kernel",
                   "frame": f"{cw}x{ch}" + (f" x {nfr} frames in one launch (vt_nv12_to_rgb8_batch_device)" if nfr else ""),
                   "us": us, "algorithmic_bytes": by, "GBps": by / us / 1e3, "frac_of_peak": by / us / 1e3 / PEAK_HBM_GBS}
            if nfr == 30:       # HBM bytes of this shape from the committed PMC passes (separate rocprofv3 runs): 1.00 x algorithmic
                try:
                    pm = json.load(open(os.path.join(ROOT, "profiles", "r06_nv12_batch_pmc.json")))
                    rec["traffic"], rec["traffic_source"] = pm["traffic_bytes_per_launch"], pm["provenance"]
                except Exception:
                    rec["traffic"] = None
            bk.append(rec)
        pre = [p for p in prof if p["name"] == "preproc_search"]
        if pre:
            crop_side = 4.0 * sq
            by = Bg * (crop_side * crop_side * 1.5 + mi.search_size ** 2 * 3 * 2)
            us = pre[0]["ms"] * 1e3
            bk.append({"kernel": "preproc_kernel", "frame": f"{Bg} streams, {int(crop_side)}-px NV12 window -> "
                       f"{mi.search_size}x{mi.search_size}x3 bf16 patch rows", "us": us, "algorithmic_bytes": by,
                       "GBps": by / us / 1e3, "frac_of_peak": by / us / 1e3 / PEAK_HBM_GBS})
        # the GEMMs that write the residual stream with K = D (proj: 152 FLOP/B, under the chip's 312 FLOP/B
        # balance) are byte-bound: operands once + the 3-byte residual pair (bf16 hi + int8 lo) read and written
        for p in prof:
            if p["name"].startswith("gemm_bf16_xresid") and p["bytes"] > 0 and p["flops"] / p["bytes"] < 312.0:
                us = p["ms"] * 1e3 / max(p["launches"], 1)
                by = p["bytes"] / max(p["launches"], 1)
                bk.append({"kernel": p["name"], "frame": f"{Bg} streams x {mi.tokens_template + mi.tokens_search} tokens, "
                           "operands once + 3-byte residual pair read and written", "us": us, "algorithmic_bytes": by,
                           "flop_per_byte": p["flops"] / p["bytes"],
                           "GBps": by / us / 1e3, "frac_of_peak": by / us / 1e3 / PEAK_HBM_GBS})
        out["byte_kernels"] = {"peak_GBps": PEAK_HBM_GBS, "kernels": bk}

    # ---- CPU baseline: float32 on all host cores, same clip, bounded sample ------
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        from oracle import cpu_fp32, vit_ref
        cores = usable_cores()
        trk = cpu_fp32.VitTrackFp32(wpath, threads=cores)
        fr0 = vit_ref.Frame.nv12(host[0], fw, fh)
        trk.init(fr0, sc.gt_box(0))
        trk.update(fr0)  # warm up the thread pool
        n, a = 0, time.perf_counter()
        ok_cpu = True
        while True:
            r = trk.update(vit_ref.Frame.nv12(host[(n + 1) % R], fw, fh))
            ok_cpu = ok_cpu and r.success and iou(r.bbox, sc.gt_box((n + 1) % R)) > 0.5
            n += 1
            if time.perf_counter() - a > 12.0 or n >= 200:
                break
        cpu_dt = time.perf_counter() - a
        # the reference's own CPU stage: whole-frame NV12->RGB on 8 threads (src/main.rs:43-46) and on all cores
        conv = {}
        for nt in sorted({min(8, cores), cores}):
            c0 = time.perf_counter()
            for i in range(20):
                vit_ref.nv12_to_rgb8(host[i % R], fw, fh, nt)
            conv[str(nt)] = (time.perf_counter() - c0) / 20 * 1e3
        out["cpu_baseline"] = {
            "value": n / cpu_dt, "unit": "frames/s", "cores": trk.threads, "kind": "port",
            "sample": f"{n} updates of the same clip: float32 torch-CPU forward on {trk.threads} threads "
                      f"(oracle/cpu_fp32.py; C pixel stages and decode), {cpu_dt:.1f} s",
            "tracked_ok": bool(ok_cpu), "cpu_model": cpu_fp32.cpu_model_name(), "nproc": os.cpu_count(),
            "nv12_full_frame_convert_ms_by_threads": conv,
        }

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def dry_run(args, vt, vd, world, rank, cfg_name, wl_text, affinity):
    """No GPU: the ranks rendezvous over gloo, broadcast the (tiny) weight blob, plan their streams and
    aggregate synthetic per-rank timings. Covers what the N > 1 leg adds around the kernels - the
    launcher, the collective record, the per-rank plan and the aggregation - on a CPU-only machine."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    wpath = vt.weights.ensure_weights(cfg_name) if rank == 0 else None
    collective = {"backend": "none", "world_size": 1, "broadcast_bytes": 0, "broadcast_ms": 0.0}
    if world > 1:
        blob, collective = vd.timed_broadcast_weights(wpath, device="cpu")
        collective["world_size"] = dist.get_world_size()
        del blob
    B = args.streams if args.streams > 0 else 2
    G = 1 if args.streams == 1 else args.groups
    plan = vd.plan_rank(rank, world, B, args.ring)
    masks = vd.gather_objects({"rank": rank, "running_on": sorted(os.sched_getaffinity(0)), **affinity})
    local_dt = 0.5 + 0.25 * rank                    # synthetic: rank r "took" 0.5 + r/4 seconds
    agg = vd.aggregate_throughput(B * args.steps, local_dt)
    per_rank = vd.gather_per_rank(B * args.steps / local_dt)
    if rank == 0:
        print(json.dumps({
            "metric": baseline_metric(), "dry_run": True, "value": agg["frames_per_s"], "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": agg["seconds"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "none (dry run: synthetic timings, no kernels ran)",
            "config": {"workload": wl_text, "vit_config": cfg_name, "streams_per_gpu": B, "engines_per_gpu": G,
                       "cfg4_literal": bool(B == 1 and G == 1 and world == 8), "cfg4_shape": bool(B == 1 and G == 1),
                       "cpu_affinity": affinity},
            "cpu_affinity_by_rank": masks,
            "per_rank_fps": per_rank, "collective": collective,
            "env": {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")},
            "global_stream_ids_rank0": plan["global_stream_ids"]}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
