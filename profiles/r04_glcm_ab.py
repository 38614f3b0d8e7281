#!/usr/bin/env python3
"""Texture kernel of the dense case (window 7, step 1, 32 levels) at 16384^2: k4_glcm_quad (2 x 2 windows per thread, r04)
against k4_glcm_pair (r02), same quantised plane, HIP-event timers of the library, and the outputs compared bit for bit.
Usage (GPU box): python profiles/r04_glcm_ab.py > gpurun_out/r04/glcm_ab.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rs-image-segmentation_amd"), ROOT]
import torch  # noqa: E402

import bench  # noqa: E402
from rsseg.runtime import Context  # noqa: E402

ctx = Context(0, use_dist=False)
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nir = bench.synth_rows(torch, ctx.device, W, 0, H, want=[3])[0]
q = (nir / 255.0 * 31).to(torch.uint8)
out = {"note": f"ms per launch of the dense texture kernel on a {H}x{W} plane (32 levels), 5 launches each", "kernels": {}}
res = {}
for kern in ("pair", "quad", "pair", "quad"):
    os.environ["RSSEG_GLCM_DENSE"] = kern
    ctx.glcm(q, H, W, 32, 7, 1)
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(5):
        maps, _ = ctx.glcm(q, H, W, 32, 7, 1)
    ms, cnt = ctx.prof_get("glcm")
    ctx.prof_enable(False)
    out["kernels"].setdefault(kern, []).append(round(ms / cnt, 3))
    res[kern] = [m.clone() for m in maps]
out["bit_identical"] = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(res["pair"], res["quad"]))
print(json.dumps(out, indent=1))
