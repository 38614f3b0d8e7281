#!/usr/bin/env python3
"""VERDICT r02 item 4, the measurement: per Lloyd iteration, the fraction of 1024-pixel tiles / 256-pixel wave tiles /
32-pixel cache-line groups in which NO label changes — the upper bound of what an exact incremental (changed-pixels-only)
update with conservative distance bounds could skip — on the `--data hard` raster of bench.py (config 3, 15 features, k = 8).
Run on the GPU box:  python3 profiles/lloyd_tile_stability.py [size] > profiles/r03_lloyd_tile_stability.json
The label map after t iterations is rsseg_kmeans_fit_predict(max_iter = t) (the E-step with the centres of iteration t)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rs-image-segmentation_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from rsseg import pipeline as P  # noqa: E402
from rsseg.runtime import Context  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
ctx = Context(0, use_dist=False)
bands = bench.synth_rows(torch, dev, size, 0, size, kind="hard")
labels, meta, planes = P.config3(ctx, bands, size, size, 8, 7, 1, 3, size * size)
n_iter = int(meta["n_iter"])
del labels, bands
rows = []
prev = None
for t in range(1, n_iter + 1):
    lab, m = ctx.kmeans_fit_predict(planes, 8, max_iter=t)
    lab = lab.to(torch.uint8)
    if prev is not None:
        ch = lab != prev
        n = ch.numel()
        rows.append({"iteration": t, "changed_px": float(ch.float().mean()),
                     **{f"unchanged_tiles_{g}px": float(1.0 - ch.view(n // g, g).any(1).float().mean()) for g in (32, 256, 1024)}})
        print(rows[-1], file=sys.stderr, flush=True)
    prev = lab
out = {"raster": [size, size, 7], "data": "bench.py --data hard", "config": "c3: 15 float32 features, KMeans k=8", "n_iter": n_iter,
       "note": "fraction of aligned groups of consecutive pixels without a label change between the E-steps of iterations t-1 and t",
       "iterations": rows}
print(json.dumps(out, indent=1))
