#!/usr/bin/env python3
"""Config-5 forest: for how many levels from the root do the 64 pixels of a wave (64 consecutive pixels of a row) sit on the SAME
node of a tree?  On those levels a walk needs no per-lane LDS gather (the node and the feature index are wave-uniform).
Measured on the real forest and real feature rows, 65 536 consecutive pixels = 1024 waves x 100 trees.
Usage (GPU box): python profiles/r04_forest_wave_coherence_sim.py > gpurun_out/r04/forest_wave_coherence_sim.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rs-image-segmentation_amd"), ROOT]
import torch  # noqa: E402

import bench  # noqa: E402
from rsseg import pipeline as P  # noqa: E402
from rsseg.runtime import Context  # noqa: E402

dev = torch.device("cuda", 0)
ctx = Context(0, use_dist=False)
W = 16384
fm = bench.fit_c5_forest(torch, None, dev, P, 0, 1, W)
model = fm["model"]
Ht = 2048
tb = bench.synth_rows(torch, dev, W, 4096, 4096 + Ht)
planes, _ = P.feature_stack19(ctx, tb, Ht, W)
fp = P.stack19_forest_planes(ctx, planes)
start = 700 * W + 3000
X = np.stack([p[start:start + 65536].cpu().numpy() for p in fp], 1)
conv = []          # per (wave, tree): levels walked together
depth_all = []
for est in model.estimators_:
    tr = est.tree_
    node = np.zeros(X.shape[0], np.int64)
    together = np.zeros(X.shape[0] // 64, np.int32)
    alive = np.ones(X.shape[0] // 64, bool)
    d = np.zeros(X.shape[0], np.int32)
    for level in range(64):
        leaf = tr.children_left[node] == -1
        if leaf.all():
            break
        same = (node.reshape(-1, 64) == node.reshape(-1, 64)[:, :1]).all(1)
        alive &= same
        together += alive & ~leaf.reshape(-1, 64).all(1)
        f = tr.feature[node]
        go_left = X[np.arange(X.shape[0]), np.where(leaf, 0, f)] <= tr.threshold[node]
        nxt = np.where(go_left, tr.children_left[node], tr.children_right[node])
        d += ~leaf
        node = np.where(leaf, node, nxt)
    conv.append(together)
    depth_all.append(d)
conv = np.stack(conv, 1)
depth_all = np.stack(depth_all, 1)
out = {"note": "levels (node visits that are not leaves) the 64 lanes of a wave walk on one common node, per (wave, tree); config-5 forest, 1024 waves x 100 trees",
       "mean_levels_walked_together": round(float(conv.mean()), 2), "mean_path_edges": round(float(depth_all.mean()), 2),
       "fraction_of_visits_that_are_wave_uniform": round(float(conv.sum() * 64) / float(depth_all.sum()), 4),
       "histogram_levels_together": {str(k): int((conv == k).sum()) for k in range(0, 17)}}
print(json.dumps(out, indent=1))
