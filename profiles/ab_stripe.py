"""A/B of two builds of the library on one box: config 3 on an H x 16384 raster (what a rank of 16384/H computes)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rs-image-segmentation_amd")); sys.path.insert(0, ROOT)
from rsseg import _lib
if len(sys.argv) > 2 and sys.argv[2] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
    for name in ("rsseg_ctx_host_syncs",):
        _lib.SIGNATURES.pop(name, None)
import torch
import bench
from rsseg.runtime import Context
from rsseg import pipeline as P
dev = torch.device("cuda:0")
ctx = Context(0)
H, W = int(sys.argv[1]), 16384
kind = sys.argv[3] if len(sys.argv) > 3 else "easy"
bands = bench.synth_rows(torch, dev, W, 0, H, kind=kind)
for i in range(3):
    labels, meta, _ = P.config3(ctx, bands, H, W, 8, 7, 1, 3, H * W)
torch.cuda.synchronize()
ts = []
for i in range(7):
    torch.cuda.synchronize(); t = time.perf_counter()
    labels, meta, _ = P.config3(ctx, bands, H, W, 8, 7, 1, 3, H * W)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
ts.sort()
print(os.path.basename(_lib.LIB_PATH), H, kind, "rows: median ms/step", round(ts[len(ts) // 2], 2), "min", round(ts[0], 2), "init", round(meta["ms_init"], 2), "lloyd", round(meta["ms_lloyd"], 2), "iters", meta["n_iter"], flush=True)
