#!/usr/bin/env python3
"""Would exact triangle-inequality pruning pay in the k-means++ sweeps?  A pixel x whose nearest chosen centre is c0 cannot be
improved by a candidate c when ||c - c0|| >= 2 sqrt(closest(x) + E) + sqrt(2E) (E bounds the float32 rounding of both computed
distances), so a GROUP of consecutive pixels that all satisfy this for all four candidates of a round could skip its 60 B/px
feature read (its contribution to every candidate's potential is its present closest distance).  This script replays
k-means++ (sklearn's greedy form, 2 + log k trials) on the bench rasters with the product's feature planes and reports, per
round, the fraction of aligned groups of 64 / 256 / 1024 consecutive pixels that are prunable as a whole.
Run on the GPU box:  python3 profiles/kpp_prune_sim.py [size] [easy|hard] > profiles/r03_kpp_prune_sim_<kind>.json"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rs-image-segmentation_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from rsseg import pipeline as P  # noqa: E402
from rsseg.runtime import Context  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
kind = sys.argv[2] if len(sys.argv) > 2 else "easy"
k = 8
dev = torch.device("cuda", 0)
ctx = Context(0, use_dist=False)
bands = bench.synth_rows(torch, dev, size, 0, size, kind=kind)
labels, meta, planes = P.config3(ctx, bands, size, size, k, 7, 1, 3, size * size)
del labels, bands
F, n = len(planes), planes[0].numel()
scale, mn, mean = (torch.tensor(np.asarray(meta[key], np.float32), device=dev) for key in ("scale", "min", "mean"))
CH = 1 << 24


def rows_of(idx):   # the scaled, centred rows of the pixels idx
    return torch.stack([(planes[f][idx] - mn[f]) * scale[f] - mean[f] for f in range(F)], 1)


def dist_to(cands, lo, hi):   # (L, hi - lo) squared distances, float32
    d = torch.zeros((cands.shape[0], hi - lo), device=dev)
    for f in range(F):
        x = (planes[f][lo:hi] - mn[f]) * scale[f] - mean[f]
        d += (x[None, :] - cands[:, f][:, None]) ** 2
    return d


rs = np.random.RandomState(42)
L = 2 + int(math.log(k))
E = F * F * 2.0 ** -20
first = int(rs.choice(n))
centres = rows_of(torch.tensor([first], device=dev))
closest = torch.empty(n, device=dev)
label = torch.zeros(n, dtype=torch.uint8, device=dev)
for lo in range(0, n, CH):
    closest[lo:lo + CH] = dist_to(centres, lo, min(n, lo + CH))[0]
rounds = []
for r in range(1, k):
    pot = closest.sum(dtype=torch.float64)
    cum = torch.cumsum(closest.to(torch.float64), 0)
    targets = torch.tensor(rs.uniform(size=L), device=dev, dtype=torch.float64) * pot
    cid = torch.searchsorted(cum, targets).clamp_(max=n - 1)
    del cum
    cands = rows_of(cid)
    D = torch.cdist(cands.double(), centres.double())            # (L, r)
    thr = D.min(0).values.float()                                 # every candidate must be that far from centre j
    prunable = 2.0 * torch.sqrt(closest + E) + math.sqrt(2 * E) <= thr[label.long()]
    row = {"round": r, "prunable_px": float(prunable.float().mean()),
           **{f"prunable_groups_{g}px": float(prunable.view(n // g, g).all(1).float().mean()) for g in (64, 256, 1024)}}
    pots = torch.zeros(L, dtype=torch.float64, device=dev)
    newd = []
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        d = torch.minimum(dist_to(cands, lo, hi), closest[lo:hi][None, :])
        # exactness check of the rule itself: a prunable pixel's minimum must be its present closest distance
        bad = (d < closest[lo:hi][None, :]) & prunable[lo:hi][None, :]
        assert not bool(bad.any()), "pruning rule violated"
        pots += d.sum(1, dtype=torch.float64)
        newd.append(d)
    w = int(torch.argmin(pots))
    for i, lo in enumerate(range(0, n, CH)):
        hi = min(n, lo + CH)
        better = newd[i][w] < closest[lo:hi]
        label[lo:hi][better] = r
        closest[lo:hi] = newd[i][w]
    del newd
    centres = torch.cat([centres, cands[w:w + 1]])
    rounds.append(row)
    print(row, file=sys.stderr, flush=True)
saved = {g: sum(x[f"prunable_groups_{g}px"] for x in rounds) for g in (64, 256, 1024)}
print(json.dumps({"raster": [size, size, 7], "data": f"bench.py --data {kind}", "config": f"c3: {F} float32 features, k-means++ k={k}, {L} trials per round",
                  "E": E, "rounds": rounds,
                  "sweeps_saved_of_%d" % (k - 1): {f"{g}px_groups": round(v, 3) for g, v in saved.items()},
                  "note": "a prunable group skips its 4F B/px feature read in that round; the rule is checked against the computed distances in every round"}, indent=1))
