#!/usr/bin/env python3
"""Independent replay of KMeans (k-means++ seeding round by round + Lloyd from the library's seeds) in plain torch on the config-3 feature planes of the bench raster, at a
size no CPU oracle can reach (default 32768^2 = 1 Gpixel; planes of 4.3 GB, byte offsets beyond 2^32): the seeds the library chose,
its iteration count and its labels against the replay.  The replay follows scikit-learn's steps (MinMax scaling and centring in
float32, distances in float64, greedy k-means++ with 2 + log k trials and RandomState(42), Lloyd until labels repeat or the centre
shift falls under tol) but accumulates in floating point, so labels may differ on near-ties: the script reports the fraction.
python3 profiles/replay_check.py [size] > profiles/r03_replay_check_<size>.json"""
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

size = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
k = 8
dev = torch.device("cuda", 0)
ctx = Context(0, use_dist=False)
bands = bench.synth_rows(torch, dev, size, 0, size)
labels, meta, planes = P.config3(ctx, bands, size, size, k, 7, 1, 3, size * size)
del bands
F, n = len(planes), planes[0].numel()
scale, mn, mean = (torch.tensor(np.asarray(meta[key], np.float32), device=dev) for key in ("scale", "min", "mean"))
CH = 1 << 25


def xs(f, lo, hi):      # MinMaxScaler (X * scale_ + min_) then X -= mean, float32, as the library's scaler_t does
    return planes[f][lo:hi] * scale[f] + mn[f] - mean[f]


def rows_of(idx):
    return torch.stack([planes[f][idx] * scale[f] + mn[f] - mean[f] for f in range(F)], 1)


def dist_to(C, lo, hi):   # (m, hi - lo) squared distances in float64, clamped at 0, cast to float32 (sklearn's euclidean_distances)
    d = torch.zeros((C.shape[0], hi - lo), device=dev, dtype=torch.float64)
    for f in range(F):
        x = xs(f, lo, hi).double()
        d += (x[None, :] - C[:, f].double()[:, None]) ** 2
    return d.clamp_(min=0).float()


# ---- the scaler itself
chk = {"min_max_scale_equal": True}
for f in range(F):
    lo_, hi_ = float(planes[f].min()), float(planes[f].max())
    s_ = np.float32(1.0) / np.float32(np.float32(hi_) - np.float32(lo_)) if hi_ > lo_ else np.float32(1.0)
    if abs(float(scale[f]) - float(s_)) > 1e-6 * abs(float(s_)):
        chk["min_max_scale_equal"] = False
# ---- k-means++, FOLLOWING the library's choices: a replay cannot reproduce the sampled indices bit for bit (the running sums
# differ in their last bits, and a target that falls next to a pixel boundary then lands on the neighbour, whose noise is
# different), so every round is checked on its own — from the library's previous seeds, the replay's candidates must contain the
# library's new seed (+-1 pixel), and no candidate may have a smaller potential than the library's seed
rs = np.random.RandomState(42)
L = 2 + int(math.log(k))
# RandomState.choice(n, p = w / w.sum()) with equal float32 weights: one random_sample() u, the index is searchsorted(cdf, u, 'right');
# for n a power of two the cdf is exact ((i + 1) / n), hence floor(u * n) — without the 12 GB of host arrays at 2^30 pixels
assert n & (n - 1) == 0, "the first-seed shortcut needs a power-of-two pixel count"
first = int(rs.random_sample() * n)
lib = [int(v) for v in meta["init_indices"]]
chk["first_seed_equal"] = first == lib[0]
C = rows_of(torch.tensor([lib[0]], device=dev))
closest = torch.empty(n, device=dev)
for lo in range(0, n, CH):
    closest[lo:lo + CH] = dist_to(C, lo, min(n, lo + CH))[0]
rounds = []
for r in range(1, k):
    cum = torch.cumsum(closest.double(), 0)
    targets = torch.tensor(rs.uniform(size=L), device=dev, dtype=torch.float64) * cum[-1]
    cid = torch.searchsorted(cum, targets).clamp_(max=n - 1)
    del cum
    ids = torch.cat([cid, torch.tensor([lib[r]], device=dev)])
    cands = rows_of(ids)
    pots = torch.zeros(L + 1, dtype=torch.float64, device=dev)
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        pots += torch.minimum(dist_to(cands, lo, hi), closest[lo:hi][None, :]).sum(1, dtype=torch.float64)
    near = [abs(int(c) - lib[r]) for c in cid.tolist()]
    j_ = int(np.argmin(near))
    others = [float(pots[i]) for i in range(L) if i != j_]
    # the two running sums drift apart by the accumulated last-bit differences of n distances (two formulas for the same
    # distance): a few pixels per 2^24; an indexing fault would be off by millions
    rounds.append({"round": r, "library_seed": lib[r], "nearest_replay_candidate_offset": near[j_],
                   "library_seed_potential": float(pots[L]), "smallest_other_candidate_potential": min(others),
                   "ok": near[j_] <= 1 + (n >> 24) and (near[j_] > 1 or float(pots[L]) <= min(others) * (1 + 1e-9))})
    print(rounds[-1], file=sys.stderr, flush=True)
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        closest[lo:hi] = torch.minimum(dist_to(cands[L:L + 1], lo, hi)[0], closest[lo:hi])
    C = torch.cat([C, cands[L:L + 1]])
del closest
chk["kmeanspp_rounds"] = rounds
chk["kmeanspp_every_round_ok"] = all(x["ok"] for x in rounds)
# ---- Lloyd
import time  # noqa: E402
t0 = time.time()
var = torch.stack([xs(f, 0, n).double().var(unbiased=False) for f in range(F)]).mean()
print("variance done", round(time.time() - t0, 1), "s", file=sys.stderr, flush=True)
tol = float(var) * 1e-4
C = C.double()
old = None
it = 0
for it in range(1, 301):
    lab = torch.empty(n, dtype=torch.uint8, device=dev)
    sums = torch.zeros(k, F, dtype=torch.float64, device=dev)
    cnt = torch.zeros(k, dtype=torch.float64, device=dev)
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        l_ = torch.argmin(dist_to(C.float(), lo, hi), 0)
        lab[lo:hi] = l_.to(torch.uint8)
        masks = [l_ == j for j in range(k)]          # masked reductions: bincount's atomics on eight bins crawl at 2^25 values
        cnt += torch.stack([m_.sum() for m_ in masks]).double()
        for f in range(F):
            x = xs(f, lo, hi).double()
            sums[:, f] += torch.stack([(x * m_).sum() for m_ in masks])
    newC = (sums / cnt.clamp(min=1)[:, None]).float().double()
    print("lloyd iteration", it, round(time.time() - t0, 1), "s", file=sys.stderr, flush=True)
    if old is not None and bool((lab == old).all()):
        break
    shift = float(((newC - C) ** 2).sum())
    C = newC
    old = lab
    if shift <= tol:
        lab = torch.empty(n, dtype=torch.uint8, device=dev)
        for lo in range(0, n, CH):
            hi = min(n, lo + CH)
            lab[lo:hi] = torch.argmin(dist_to(C.float(), lo, hi), 0).to(torch.uint8)
        break
chk["n_iter_library"], chk["n_iter_replay"] = int(meta["n_iter"]), it
diff = 0
for lo in range(0, n, CH):
    hi = min(n, lo + CH)
    diff += int((labels[lo:hi].to(torch.uint8) != lab[lo:hi]).sum())
chk["labels_differing"], chk["labels_differing_fraction"] = diff, diff / n
chk["raster"] = [size, size, 7]
chk["note"] = "library = rsseg config 3 (exact fixed-point sums); replay = plain torch floating point; differences are near-ties"
print(json.dumps(chk, indent=1))
