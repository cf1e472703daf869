"""Re-make every committed oracle trajectory with the parameters it records (configuration, seed, target size, frame count,
occlusion): needed whenever the oracle's numerical specification changes (round 6: the 3-byte residual pair).
   python tests/golden/regen_traj.py [name.npz ...]       (default: all traj_*.npz and forced_*.npz, small configurations first)
CPU only; the cfg5 (4K, ViT-L/14) fixtures take ~50 minutes each on 8 cores."""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_traj as mt  # noqa: E402
import gstreamer_vit_tracker_amd as vt  # noqa: E402

names = sys.argv[1:] or sorted((os.path.basename(p) for p in glob.glob(os.path.join(HERE, "traj_*.npz")) +
                                glob.glob(os.path.join(HERE, "forced_*.npz"))),
                               key=lambda n: ({"cfg2": 0, "cfg3": 1, "cfg5": 2}[n.split("_")[1]], n))
for name in names:
    path = os.path.join(HERE, name)
    with np.load(path) as z:
        cfg, frames, seed, sq = str(z["config"]), int(z["frames"]), int(z["seed"]), int(z["square"])
        hide = tuple(int(v) for v in z["hide"]) if "hide" in z.files and z["hide"][1] > z["hide"][0] else None
    weights = mt.gen1_weights(cfg) if name.startswith("forced_") else vt.weights.ensure_weights(cfg)
    print(f"== {name}: {cfg}, {frames} frames, seed {seed}, square {sq}, hide {hide}", flush=True)
    mt.run(cfg, weights, frames, seed, path, square=sq, hide=hide)
