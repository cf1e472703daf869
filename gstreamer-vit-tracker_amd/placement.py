"""CPU placement of a rank: one process per GPU, pinned to the CPUs of that GPU's NUMA node.

The reference gives its CPU stage a thread pool of its own (/root/reference/src/main.rs:43-46) and runs one
tracker per process (src/pipeline.rs:55). Here every rank of an 8-GPU node packs the search windows of its
streams on the CPU (vt_group_enqueue_host: one ingest thread per engine) and copies them to ITS GPU; with
8 ranks x 2 packing threads the first thing to bite is ranks wandering across sockets (SURVEY.md section 8(e):
the host side is the limiter). So, BEFORE torch / HIP load (threads the runtimes start inherit the mask):

  * the CPUs local to the rank's GPU come from sysfs alone - KFD topology node -> drm render minor ->
    /sys/class/drm/renderD<minor>/device/local_cpulist - no HIP call, nothing initialises the GPU;
  * ranks whose GPUs share a NUMA node split that node's CPUs into disjoint contiguous slices;
  * without the topology (CPU-only box, container without /sys/class/kfd) the allowed CPUs are sliced by
    local rank.

Pure functions + one `apply`; no torch, no HIP import (bench.py calls this first thing in a rank).
"""
from __future__ import annotations

import glob
import os
import re


def parse_cpulist(text: str) -> list:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def format_cpulist(cpus) -> str:
    cpus = sorted(set(cpus))
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def gpu_local_cpulists(sysfs: str = "/sys") -> list:
    """[CPUs local to GPU 0, GPU 1, ...] in KFD node order (= HIP device order without *_VISIBLE_DEVICES
    remapping), each a list of ints; [] when the topology is not readable."""
    nodes = []
    for d in glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/*")):
        try:
            props = dict(line.split()[:2] for line in open(os.path.join(d, "properties")) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:        # CPU nodes have none
                nodes.append((int(os.path.basename(d)), int(props.get("drm_render_minor", "-1"))))
        except (OSError, ValueError):
            continue
    out = []
    for _, minor in sorted(nodes):
        try:
            out.append(parse_cpulist(open(os.path.join(sysfs, f"class/drm/renderD{minor}/device/local_cpulist")).read()))
        except (OSError, ValueError):
            out.append([])
    return out


def visible_device_map(env=None) -> "list | None":
    """local device index -> KFD GPU index when ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES hold plain integers"""
    env = os.environ if env is None else env
    for key in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(key, "").strip()
        if v:
            if re.fullmatch(r"\d+(,\d+)*", v):
                return [int(x) for x in v.split(",")]
            return None          # UUIDs: order unknown
    return None


def rank_cpu_mask(local_rank: int, world: int, allowed, gpu_cpus=None, devmap=None):
    """(cpus, source) for one rank: disjoint between the ranks of a node by construction.
    allowed: CPUs this process may use now; gpu_cpus: gpu_local_cpulists(); devmap: visible_device_map()."""
    allowed = sorted(set(allowed))
    if world < 1 or not (0 <= local_rank < world) or not allowed:
        raise ValueError("bad rank / world / empty CPU set")

    def node_of(r):
        g = devmap[r] if devmap and r < len(devmap) else r
        if gpu_cpus and g < len(gpu_cpus):
            loc = [c for c in gpu_cpus[g] if c in set(allowed)]
            if loc:
                return tuple(loc)
        return None

    mine = node_of(local_rank)
    if mine is not None and all(node_of(r) is not None for r in range(world)):
        sharers = [r for r in range(world) if node_of(r) == mine]          # ranks on the same NUMA node
        k, n = sharers.index(local_rank), len(sharers)
        if len(mine) >= n:
            lo, hi = k * len(mine) // n, (k + 1) * len(mine) // n
            return list(mine[lo:hi]), f"NUMA-local CPUs of the rank's GPU, slice {k + 1}/{n} of {format_cpulist(mine)}"
    if len(allowed) >= world:
        lo, hi = local_rank * len(allowed) // world, (local_rank + 1) * len(allowed) // world
        return allowed[lo:hi], f"slice {local_rank + 1}/{world} of the allowed CPUs {format_cpulist(allowed)} (no GPU topology)"
    return allowed, "all allowed CPUs (fewer CPUs than ranks)"


def apply(local_rank: int, world: int) -> dict:
    """Pin the calling process (and every thread it starts later: ingest threads, the HIP runtime's) and
    return the record bench.py prints under config.cpu_affinity."""
    if not hasattr(os, "sched_setaffinity"):
        return {"cpus": "", "n": 0, "source": "unsupported platform"}
    allowed = os.sched_getaffinity(0)
    if world <= 1:       # a single rank keeps what it was given (the 1-GPU box: its whole cgroup share)
        return {"cpus": format_cpulist(allowed), "n": len(allowed), "source": "single rank: mask left as given"}
    cpus, src = rank_cpu_mask(local_rank, world, allowed, gpu_local_cpulists(), visible_device_map())
    try:
        os.sched_setaffinity(0, cpus)
    except OSError as e:
        return {"cpus": format_cpulist(allowed), "n": len(allowed), "source": f"sched_setaffinity failed: {e}"}
    return {"cpus": format_cpulist(cpus), "n": len(cpus), "source": src}
