#!/usr/bin/env python3
"""The 19-feature stack of config 5 (indices, PC0, their 7x7 context means, GLCM 21/21 contrast / homogeneity, 5x5 morphological
gradient, 5x5 local std, Sobel magnitude) against the CPU oracle at a size beyond the test suite's, on SURVEY 8d's raster:
column by column, bit-identical or the largest deviation.  Usage: python tests/at_size/parity_stack19_at_size.py 4096 > out.json"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "rs-image-segmentation_amd"), ROOT]
from oracle import ref_np as O  # noqa: E402
from rsseg import pipeline as P  # noqa: E402
from rsseg.runtime import Context  # noqa: E402

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
t0 = time.time()


def log(msg):
    print(f"[{time.time() - t0:7.1f} s] {msg}", file=sys.stderr, flush=True)


def beat():
    while True:
        time.sleep(60)
        log("... still running")


threading.Thread(target=beat, daemon=True).start()
ctx = Context(0, use_dist=False)
r = O.synthetic_raster(H, W)
bands = [np.ascontiguousarray(r[i]) for i in range(7)]
planes, _ = P.feature_stack19(ctx, [ctx.to_device(b.reshape(-1)) for b in bands], H, W)
stack = P.stack19_to_host(planes, H, W)
log("product")
_, hier = O.run_feature_extraction_stage(bands)
log("oracle")
names = ["ndwi", "mndwi", "ndvi", "evi", "ndbi", "bsi", "pc0"] + [f"ctx7_{n}" for n in ("ndwi", "mndwi", "ndvi", "evi", "ndbi", "bsi", "pc0")] + \
        ["glcm_contrast", "glcm_homogeneity", "gradient_5", "std_dev_scale_5", "sobel_mag"]
cols = {}
for c, nm in enumerate(names):
    a, b = stack[:, :, c], hier["all"][:, :, c]
    cols[nm] = {"bit_identical": bool(np.array_equal(a, b)), "max_abs_dev": float(np.abs(a - b).max())}
print(json.dumps({"raster": f"oracle.synthetic_raster {H}x{W}x7", "pixels": H * W, "dtype": str(stack.dtype), "columns": cols,
                  "seconds": round(time.time() - t0, 1)}, indent=1))
