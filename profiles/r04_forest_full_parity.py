#!/usr/bin/env python3
"""Config 5 at its real size with its real forest, EXHAUSTIVELY: RandomForestClassifier(100 trees, max_depth 16) fitted on the
19-feature stack (bench.fit_c5_forest), the 16384 x 16384 stack classified on the GPU, and EVERY pixel compared with
model.predict on the same rows (chunks of 8 M rows, 32 threads: scikit-learn's per-tree probability buffers stay below 20 GB).
The test suite does the same on 30 000 sampled pixels.  Usage: python profiles/r04_forest_full_parity.py > out.json"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rs-image-segmentation_amd"), ROOT]
import torch  # noqa: E402

import bench  # noqa: E402
from rsseg import pipeline as P  # noqa: E402
from rsseg.runtime import Context  # noqa: E402

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
t0 = time.time()


def log(msg):
    print(f"[{time.time() - t0:7.1f} s] {msg}", file=sys.stderr, flush=True)


def beat():
    while True:
        time.sleep(60)
        log("... still running")


threading.Thread(target=beat, daemon=True).start()
ctx = Context(0, use_dist=False)
fm = bench.fit_c5_forest(torch, None, ctx.device, P, 0, 1, W)
model = fm["model"]
model.n_jobs = 32
ctx.forest_load(fm["flat"])
bands = bench.synth_rows(torch, ctx.device, W, 0, H)
planes, _ = P.feature_stack19(ctx, bands, H, W)
fp = P.stack19_forest_planes(ctx, planes)
got = ctx.forest_predict(fp)
torch.cuda.synchronize()
log("product")
n = H * W
CH = 1 << 23
bad = 0
classes = {}
for lo in range(0, n, CH):
    hi = min(lo + CH, n)
    X = np.stack([p[lo:hi].cpu().numpy() for p in fp], 1)
    want = model.predict(X)
    g = got[lo:hi].cpu().numpy()
    bad += int((g != want).sum())
    u, c = np.unique(want, return_counts=True)
    for a, b in zip(u.tolist(), c.tolist()):
        classes[a] = classes.get(a, 0) + b
    log(f"rows {hi} of {n}: {bad} differing so far")
print(json.dumps({"raster": f"bench.synth_rows {H}x{W}x7", "pixels": n, "forest": "100 trees, max_depth 16, 19 features (bench.fit_c5_forest)",
                  "labels_differing_from_model_predict": bad, "class_counts": classes, "seconds": round(time.time() - t0, 1)}, indent=1))
