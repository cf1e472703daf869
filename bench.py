#!/usr/bin/env python3
"""bench.py — tracked frames/sec of the HIP hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A step is one pass of the hot path over one batch of synthetic input: every one of the B streams
of a GPU gets its next 1080p NV12 frame (already resident in HBM) and produces one vt_result
(NV12 window -> crop/resize/normalise -> patch embed -> 12-block joint encoder -> centre head ->
box decode, all inside libvittrack_hip.so). value = tracked frames of ALL ranks / max-over-ranks
wall time. Streams are independent (one frame chain each); ranks exchange nothing per frame —
the only collective is the start-up weight broadcast (RCCL), outside the timed region.

Also on the JSON line: "roofline" for the dominant kernel (HIP events on the library's own stream,
algorithmic FLOPs / measured time against the 2.5 PFLOP/s dense bf16 MFMA peak) and
"cpu_baseline" (the CPU oracle timed on this host, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
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
        return "tracked frames/sec @1080p ViT-B/16 384\u00d7192, 1 GPU; + MFMA roofline %"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="independent tracked streams per GPU (0: groups x weights.recommended_streams)")
    ap.add_argument("--groups", type=int, default=2,
                    help="engines per GPU, each with streams/groups streams on its own HIP stream "
                         "(kernels of different groups overlap on the chip)")
    ap.add_argument("--engines", default="",
                    help="engine sizes instead of --groups equal ones: '30+1', or 'auto' = "
                         "vt_plan_engines(--streams)")
    ap.add_argument("--ring", type=int, default=64, help="distinct frames kept in HBM per clip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the PCIe-inclusive legs")
    ap.add_argument("--no-single-leg", action="store_true", help="skip the one-tracker-per-process leg")
    ap.add_argument("--host-steps", type=int, default=100)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--eager", action="store_true", help="no hipGraph replay")
    args = ap.parse_args()

    import numpy as np
    import torch
    import gstreamer_vit_tracker_amd as vt
    from gstreamer_vit_tracker_amd import distributed as vd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    cfg_name, wl_text = WORKLOADS[args.workload]
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
    if world > 1:
        blob = vd.broadcast_weights(wpath, device=dev)
        grps = [vt.Group(n_streams=b, device=local, use_graph=not args.eager,
                         device_blob=(blob.data_ptr(), blob.numel())) for b in sizes]
        del blob
    else:
        grps = [vt.Group(wpath, n_streams=b, device=local, use_graph=not args.eager) for b in sizes]
    grp = grps[0]
    mi = grp.model_info()

    # ---- synthetic clip: R frames of a closed path, resident in HBM; stream i runs it with a phase
    plan = vd.plan_rank(rank, world, B, R)
    sc = vt.synth.MovingSquare(fw, fh, sq, seed=plan["clip_seed"], path="circle", period=R,
                               amp=3.8 * R / (2 * np.pi))
    host = np.stack([sc.frame_nv12(t) for t in range(R)])
    clip = torch.from_numpy(host).to(dev)
    fbytes = host.shape[1]
    base = clip.data_ptr()
    phase = plan["phase"]
    frames_at = []
    for t in range(R):
        frames_at.append([vt.frame_nv12(base + ((t + phase[i]) % R) * fbytes,
                                        base + ((t + phase[i]) % R) * fbytes + fw * fh, fw, fh)
                          for i in range(B)])
    for i in range(B):
        grps[eng[i]].init_device(i - off[eng[i]], frames_at[0][i], vt.BBox.new(*sc.gt_box(phase[i])))

    def enqueue_all(t):
        fr = frames_at[t % R]
        for g in range(G):
            grps[g].enqueue_device(fr[off[g]:off[g + 1]])

    def wait_all():
        res = []
        for g in range(G):
            res += grps[g].wait()
        return res

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warm-up (untimed), then exactly K timed steps -------------------------------------------
    for t in range(W):
        enqueue_all(t)
    wait_all()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(W, W + K):
        enqueue_all(t)
    res = wait_all()
    torch.cuda.synchronize()
    barrier()
    dt_local = time.perf_counter() - t0
    agg = vd.aggregate_throughput(B * K, dt_local, device=dev if world > 1 else "cpu")
    dt, total_frames, fps = agg["seconds"], agg["frames"], agg["frames_per_s"]

    # sanity: every stream really tracked every frame, and still sits on the square
    t_last = W + K - 1
    ious, done, succ = [], [], []
    for i in range(B):
        st = grps[eng[i]].read_state(i - off[eng[i]])
        done.append(st["frames_done"])
        succ.append(st["success_count"])
        ious.append(iou(res[i].bbox, sc.gt_box((t_last + phase[i]) % R)))
    tracked_ok = all(d == W + K for d in done) and all(s == W + K for s in succ) and min(ious) > 0.5

    # ---- synchronous single-call latency (the literal drop-in call pattern) --------------------------
    lat = []
    for t in range(W + K, W + K + 100):
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

    out = {
        "metric": baseline_metric(),
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": wl_text, "vit_config": cfg_name, "frame": f"{fw}x{fh} NV12",
                   "streams_per_gpu": B, "engines_per_gpu": G, "engine_sizes": sizes, "tokens": mi.tokens_template + mi.tokens_search,
                   "ingest": "frames resident in HBM", "launch": "eager" if args.eager else "hipGraph",
                   "weights": "synthetic seeded encoder + fitted head (no reference weights exist)"},
        "per_stream_fps": fps / (world * B),
        "sync_update_latency_ms": lat_ms, "sync_update_latency_p99_ms": lat_p99,
        "reference_style": {   # src/timing_stats.rs semantics, synchronous calls (one frame per stream per call)
            "per_stream_fps": ref_fps_stream, "aggregate_fps": ref_fps_stream * B * world,
            "window": len(iv), "frame_latency_ms_p50": lat_ms, "frame_latency_ms_p99": lat_p99,
            "avg_track_ms": sum(iv) / len(iv) / 1e3},
        "tracked_ok": bool(tracked_ok), "min_iou_vs_truth": float(min(ious)),
        "gflop_per_frame": mi.flops_per_frame / 1e9,
        "encoder_gflop_per_frame": mi.encoder_flops_per_frame / 1e9,
        "whole_frame_mfma_frac": fps / world * mi.flops_per_frame / 1e12 / PEAK_BF16_TFLOPS,
    }

    # ---- per-kernel HIP-event timing on the library's stream -> roofline of the dominant kernel ------
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
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "r03_dominant_kernel_pmc.json")
        if os.path.exists(pmc_path):
            try:
                pm = json.load(open(pmc_path))
                if pm.get("kernel_family") == dom["name"] and pm.get("streams_per_pass") == Bg:
                    traffic = pm["traffic_bytes_per_launch"]
                    traffic_src = pm["provenance"]
            except Exception:
                pass
        out["roofline"] = {
            "bound": "mfma", "kernel": dom["name"], "achieved": ach, "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
            "traffic_source": traffic_src,
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

    full_leg = {}
    # ---- full-frame upload leg ("timing includes the 3.11 MB H2D copy per frame") ----
    # What a host that maps the whole buffer pays (/root/reference/src/pipeline.rs:95-106): every stream's
    # WHOLE NV12 frame crosses PCIe from pinned memory every step, double-buffered per engine on a copy
    # stream so that the upload of step t+1 runs under the pass of step t. Same engines, same kernels;
    # beside the window-only figure above, never `value`.
    if not args.no_host_leg and rank == 0 and world == 1:
        hs = args.host_steps
        pinned = torch.from_numpy(host).pin_memory()                       # [R][fbytes]
        es = [torch.cuda.ExternalStream(grps[g].hip_stream(), device=dev) for g in range(G)]
        # ONE copy stream for all engines: HIP multiplexes a process's streams onto four hardware queues, and
        # an upload that shares a queue with an engine's pass waits for all of it; this
        # leg runs before the library creates its own copy streams for the window-only leg below
        cs1 = torch.cuda.Stream(device=dev)
        cs = [cs1] * G
        dbuf = [[torch.empty((sizes[g], fbytes), dtype=torch.uint8, device=dev) for _ in range(2)] for g in range(G)]
        up = [[torch.cuda.Event() for _ in range(2)] for g in range(G)]
        done = [[torch.cuda.Event() for _ in range(2)] for g in range(G)]
        for i in range(B):
            grps[eng[i]].init_device(i - off[eng[i]], frames_at[0][i], vt.BBox.new(*sc.gt_box(phase[i])))

        def full_step(t):
            for g in range(G):
                k = t & 1
                with torch.cuda.stream(cs[g]):
                    cs[g].wait_event(done[g][k])                           # the pass that read this buffer is over
                    for j, i in enumerate(range(off[g], off[g + 1])):
                        dbuf[g][k][j].copy_(pinned[(t + phase[i]) % R], non_blocking=True)
                    up[g][k].record(cs[g])
                es[g].wait_event(up[g][k])
                base_g = dbuf[g][k].data_ptr()
                grps[g].enqueue_device([vt.frame_nv12(base_g + j * fbytes, base_g + j * fbytes + fw * fh, fw, fh)
                                        for j in range(sizes[g])])
                done[g][k].record(es[g])

        for t in range(1, 4):
            full_step(t)
        wait_all()
        torch.cuda.synchronize()
        f0 = time.perf_counter()
        for t in range(4, 4 + hs):
            full_step(t)
        fres = wait_all()
        torch.cuda.synchronize()
        fdt = time.perf_counter() - f0
        f_ok = all(r.success for r in fres) and min(iou(fres[i].bbox, sc.gt_box((3 + hs + phase[i]) % R)) for i in range(B)) > 0.5
        ffps = B * hs / fdt
        full_leg = ({
            "full_frame_value": ffps, "full_frame_ms_per_step": fdt / hs * 1e3, "full_frame_tracked_ok": bool(f_ok),
            "full_frame_vs_hbm_resident": ffps / (fps / world),
            "full_frame_h2d_GBps": ffps * fbytes / 1e9, "pcie_link_GBps": 63.0,
            "full_frame_ingest": "every stream's whole NV12 frame (%.2f MB) copied from pinned host memory each step, "
                                 "double-buffered per engine on a copy stream" % (fbytes / 1e6)})
        del dbuf, pinned
        for i in range(B):      # back to the HBM-resident clip for what follows
            grps[eng[i]].init_device(i - off[eng[i]], frames_at[0][i], vt.BBox.new(*sc.gt_box(phase[i])))

    # ---- PCIe-inclusive leg: the same streams fed from ORDINARY host memory (vt_group_update_host) ----
    # timing that includes the H2D copy, beside the HBM-resident `value` (never
    # instead of it). Only each stream's search window crosses PCIe (~100 KB for a 64-px target; a whole
    # 1080p NV12 frame is 3.11 MB); one host thread per engine, as a host with G capture threads would.
    if not args.no_host_leg and rank == 0 and world == 1:     # like cpu_baseline: at N = 1 only
        import threading
        hs = args.host_steps
        hclip = [vt.NV12Frame(host[t], fw, fh) for t in range(R)]     # pageable numpy memory
        hg = grps          # the same engines: a vt_group takes device and host frames alike
        oks = []

        def frames_for(gi, t):
            return [hclip[(t + phase[i]) % R] for i in range(off[gi], off[gi + 1])]

        def run_host(gi, n, rec, pipelined):
            ok = True
            if pipelined:      # upload of step t+1 (copy stream) overlaps the pass of step t
                hg[gi].enqueue_host(frames_for(gi, 1))
                for t in range(2, n + 1):
                    hg[gi].enqueue_host(frames_for(gi, t))
                    ok = ok and all(r.success for r in hg[gi].wait_next())
                ok = ok and all(r.success for r in hg[gi].wait_next())
            else:
                for t in range(1, n + 1):
                    ok = ok and all(r.success for r in hg[gi].update_host(frames_for(gi, t)))
            rec.append(ok)

        def reinit():
            for i in range(B):
                hg[eng[i]].init_host(i - off[eng[i]], hclip[phase[i] % R], vt.BBox.new(*sc.gt_box(phase[i])))

        reinit()

        legs = {}
        for name, pipelined in (("sync", False), ("pipelined", True)):
            for gi in range(G):
                run_host(gi, 3, [], pipelined)
            reinit()             # the clip positions advanced: restart where the timed loop expects
            oks = []
            h0 = time.perf_counter()
            th = [threading.Thread(target=run_host, args=(gi, hs, oks, pipelined)) for gi in range(G)]
            [x.start() for x in th]
            [x.join() for x in th]
            legs[name] = (time.perf_counter() - h0, all(oks))
            reinit()
        hdt, _ = legs["pipelined"]
        oks = [legs["pipelined"][1] and legs["sync"][1]]
        win_bytes = (4 * sq + 16) ** 2 * 1.5
        out["pcie_inclusive"] = {
            "value": B * hs / hdt, "unit": "frames/s", "steps": hs, "ms_per_step": hdt / hs * 1e3,
            "tracked_ok": bool(all(oks)), "vs_hbm_resident": (B * hs / hdt) / (fps / world),
            "ingest": "vt_group_enqueue_host / vt_group_wait_next: frames in pageable host memory, search "
                      "windows (speculative: 1.75x the crop side) packed into one of two pinned arenas, one "
                      "H2D copy per engine and step on a copy stream, overlapping the previous pass",
            "synchronous_value": B * hs / legs["sync"][0],
            "redone_passes": int(sum(g_.host_redos() for g_ in hg)),
            "approx_h2d_bytes_per_frame": win_bytes, "full_frame_bytes": fw * fh * 1.5,
        }
        out["pcie_inclusive"].update(full_leg)

    # ---- the literal drop-in case: ONE tracker per process (/root/reference/src/pipeline.rs:55,109-120) ----
    # vt_create + vt_update_nv12 from a host pointer (what the reference's probe would call) and the _device
    # form on HBM-resident frames, synchronous calls, reference-style statistics; per-kernel times of one
    # update. A leg of its own: the headline batches 60 streams, the reference runs one.
    if not args.no_single_leg and rank == 0 and world == 1:
        ntrk = 300
        trk = vt.VitTrack(wpath)
        hclip1 = [vt.NV12Frame(host[t], fw, fh) for t in range(R)]
        single = {}
        for name in ("host_pointer", "device_pointer"):
            if name == "host_pointer":
                trk.init(hclip1[0], vt.BBox.new(*sc.gt_box(0)))
                call = lambda t: trk.update(hclip1[t % R])
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

    # ---- byte-bound kernels against the HBM roofline (north_star: "HBM GB/s against gfx950 peak") ------
    if not args.no_profile and rank == 0:
        bk = []
        for (cw, ch) in ((1920, 1080), (3840, 2160)):
            us = vt.op_nv12_to_rgb8_bench(cw, ch, iters=50)
            by = cw * ch * 4.5                                  # 1.5 B read + 3 B written per pixel
            bk.append({"kernel": "nv12_to_rgb8_kernel", "frame": f"{cw}x{ch}", "us": us,
                       "algorithmic_bytes": by, "GBps": by / us / 1e3, "frac_of_peak": by / us / 1e3 / PEAK_HBM_GBS})
        pre = [p for p in prof if p["name"] == "preproc_search"]
        if pre:
            crop_side = 4.0 * sq
            by = Bg * (crop_side * crop_side * 1.5 + mi.search_size ** 2 * 3 * 2)
            us = pre[0]["ms"] * 1e3
            bk.append({"kernel": "preproc_kernel", "frame": f"{Bg} streams, {int(crop_side)}-px NV12 window -> "
                       f"{mi.search_size}x{mi.search_size}x3 bf16 patch rows", "us": us, "algorithmic_bytes": by,
                       "GBps": by / us / 1e3, "frac_of_peak": by / us / 1e3 / PEAK_HBM_GBS})
        out["byte_kernels"] = {"peak_GBps": PEAK_HBM_GBS, "kernels": bk}

    # ---- CPU baseline: the oracle on this host's cores, same clip, bounded sample --------------------
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        from oracle import vit_ref
        cores = usable_cores()
        try:   # keep BLAS from oversubscribing a cgroup-limited box
            from threadpoolctl import threadpool_limits
            threadpool_limits(limits=cores)
        except Exception:
            pass
        trk = vit_ref.VitTrackRef(wpath)
        fr0 = vit_ref.Frame.nv12(host[0], fw, fh)
        trk.init(fr0, sc.gt_box(0))
        trk.update(fr0)  # warm BLAS
        n, a = 0, time.perf_counter()
        while True:
            trk.update(vit_ref.Frame.nv12(host[(n + 1) % R], fw, fh))
            n += 1
            if time.perf_counter() - a > 12.0 or n >= 100:
                break
        cpu_dt = time.perf_counter() - a
        # the reference's own CPU stage: whole-frame NV12->RGB on 8 threads (src/main.rs:43-46)
        c0 = time.perf_counter()
        for i in range(20):
            vit_ref.nv12_to_rgb8(host[i % R], fw, fh, min(8, cores))
        conv_ms = (time.perf_counter() - c0) / 20 * 1e3
        out["cpu_baseline"] = {
            "value": n / cpu_dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} updates of the same clip by the CPU oracle (NumPy/BLAS float32 with the "
                      f"same bf16 rounding points + C pixel stages), {cpu_dt:.1f} s",
            "nv12_full_frame_convert_ms_8_threads": conv_ms,
        }

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
