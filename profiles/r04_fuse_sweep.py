#!/usr/bin/env python3
"""The fused index + projection pass (k3_indices_project, config 3 form: 7 float32 bands in; 7 indices, 3 components and the
quantised texture band out) at 16384^2 against the number of workgroups (RSSEG_FUSE_GRID) and the workgroup -> pixel mapping (RSSEG_FUSE_MAP: 0 grid-stride, 1 contiguous chunks of 16 tiles), HIP-event timers of the library.
r04, VERDICT r03 item 3a.  Usage (GPU box): python profiles/r04_fuse_sweep.py > gpurun_out/r04/fuse_sweep.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rs-image-segmentation_amd"), ROOT]
import torch  # noqa: E402

import bench  # noqa: E402
from rsseg import pipeline as P  # noqa: E402
from rsseg.runtime import Context  # noqa: E402

ctx = Context(0, use_dist=False)
H = W = 16384
bands = bench.synth_rows(torch, ctx.device, W, 0, H)
out = {"note": "ms of the 'indices_project' family per config-3 step at 16384^2 (69 B/px = 18.5 GB), 6 steps each", "grid": {}}
for grid, mp in ((1024, 0), (2048, 0), (4096, 0), (8192, 0), (16384, 0), (65536, 0), (262144, 0),
                 (1024, 1), (2048, 1), (4096, 1), (8192, 1), (16384, 1)):
    os.environ["RSSEG_FUSE_GRID"] = str(grid)
    os.environ["RSSEG_FUSE_MAP"] = str(mp)
    P.config3(ctx, bands, H, W, 8)
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(6):
        P.config3(ctx, bands, H, W, 8)
    ms, cnt = ctx.prof_get("indices_project")
    ctx.prof_enable(False)
    out["grid"][f"{grid}_{'chunked' if mp else 'grid_stride'}"] = {"ms": round(ms / cnt, 4), "TBs": round(H * W * 69 / (ms / cnt * 1e-3) / 1e12, 3), "launches": cnt}
print(json.dumps(out, indent=1))
