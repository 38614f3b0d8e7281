import sys, os, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/profiles") else os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [os.path.join(ROOT, "rs-image-segmentation_amd"), ROOT]
import torch, bench
from rsseg.runtime import Context
from rsseg import pipeline as P
ctx = Context(0)
H = W = 4096
bands = bench.synth_rows(torch, ctx.device, W, 0, H)
for i in range(3): P.config2(ctx, bands, 6)
torch.cuda.synchronize()
ts = []
for i in range(9):
    torch.cuda.synchronize(); t = time.perf_counter(); lab, meta, _ = P.config2(ctx, bands, 6); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
ts.sort(); print("c2 4096 median ms", round(ts[4], 3), "min", round(ts[0], 3), "iters", meta["n_iter"], "syncs/step", ctx.host_syncs(reset=True) / 12)
