#!/usr/bin/env python3
"""K1 small-integer pass (k1_hist<3>) at 16384^2: time per plane against the number of workgroups (RSSEG_K1_GRID; 1024
threads each, 66 KB of LDS: two resident per CU), measured with the library's HIP-event timers.  r04, VERDICT r03 item 3d.
Usage (GPU box): python profiles/r04_k1_sweep.py > gpurun_out/r04/k1_sweep.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rs-image-segmentation_amd"), ROOT]
import torch  # noqa: E402

from rsseg.runtime import Context  # noqa: E402

ctx = Context(0, use_dist=False)
n = 16384 * 16384
g = torch.Generator(device="cuda")
g.manual_seed(1)
plane = torch.randint(0, 256, (n,), generator=g, device="cuda", dtype=torch.int32).to(torch.float32)
out = {"note": "ms per 16384^2 float32 plane (4 B/px = 1.07 GB) for the small-integer select pass; TB/s = 1.0737 GB / ms", "grid": {}}
ranks = [int(0.02 * (n - 1)), int(0.98 * (n - 1))]
for grid in (256, 512, 768, 1024, 1536, 2048, 4096, 8192):
    os.environ["RSSEG_K1_GRID"] = str(grid)
    for _ in range(3):
        ctx.order_stats(plane, ranks)
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(20):
        ctx.order_stats(plane, ranks)
    ms, cnt = ctx.prof_get("select")
    ctx.prof_enable(False)
    out["grid"][str(grid)] = {"ms": round(ms / cnt, 4), "TBs": round(n * 4 / (ms / cnt * 1e-3) / 1e12, 3), "launches": cnt}
print(json.dumps(out, indent=1))
