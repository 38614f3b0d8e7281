"""Scratch: the c5 forest alone (profiling).  usage: forest_only.py [size] [reps]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # profiles/ -> repository root
for p in (os.path.join(ROOT, "rs-image-segmentation_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
import bench as B
from rsseg import _lib as _L
if os.environ.get('RSSEG_LIB'):
    _L.LIB_PATH = os.environ['RSSEG_LIB']
from rsseg import pipeline as P
from rsseg.runtime import Context
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
ctx = Context(0, use_dist=False)
fm = B.fit_c5_forest(torch, None, dev, P, 0, 1, 16384)
flat = fm["flat"]
off = flat["tree_off"]
nn = np.diff(off)
leaf = flat["left"] == -1
val = flat["value"]
pure = leaf & (np.isclose(val.max(1), 1.0))
print(json.dumps(dict(n_trees=len(nn), nodes_total=int(off[-1]), nodes_per_tree_mean=float(nn.mean()), nodes_per_tree_max=int(nn.max()),
                      leaves=int(leaf.sum()), pure_leaves=int(pure.sum()), n_classes=int(val.shape[1]))), flush=True)
ctx.forest_load(flat)
H = W = size
bands = B.synth_rows(torch, dev, W, 0, H)
planes, _ = P.feature_stack19(ctx, bands, H, W)
fp = P.stack19_forest_planes(ctx, planes)
out = ctx.forest_predict(fp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = ctx.forest_predict(fp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps(dict(size=size, forest_ms=dt * 1e3, mpx_s=H * W / dt / 1e6)), flush=True)
# parity against sklearn on a sample
idx = np.random.default_rng(0).choice(H * W, 20000, replace=False)
ti = torch.from_numpy(idx).to(dev)
X = np.stack([p[ti].cpu().numpy() for p in fp], 1)
want = fm["model"].predict(X)
got = out[ti].cpu().numpy()
print(json.dumps(dict(parity_mismatch=int((want != got).sum()))), flush=True)
