#!/usr/bin/env python3
"""Size-independent sanity of the config-3 label map on the bench raster (eight prototypes on a 64-px checkerboard): every
true class should be taken by ONE cluster and block interiors should be pure.  python3 profiles/purity_check.py [size]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rs-image-segmentation_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from rsseg import pipeline as P  # noqa: E402
from rsseg.runtime import Context  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
ctx = Context(0, use_dist=False)
bands = bench.synth_rows(torch, dev, size, 0, size)
labels, meta, planes = P.config3(ctx, bands, size, size, 8, 7, 1, 3, size * size)
del planes, bands
lab = labels.view(size, size)
cont = torch.zeros(8, 8, dtype=torch.int64, device=dev)      # [true class][cluster], block interiors only (8 px from a block edge)
rows = 2048
for r0 in range(0, size, rows):
    y = torch.arange(r0, r0 + rows, device=dev)[:, None]
    x = torch.arange(size, device=dev)[None, :]
    true = ((y // 64) * 7 + (x // 64) * 3) % 8
    inner = ((y % 64 >= 8) & (y % 64 < 56) & (x % 64 >= 8) & (x % 64 < 56))
    idx = (true * 8 + lab[r0:r0 + rows].to(torch.int64))[inner]
    cont += torch.bincount(idx, minlength=64).view(8, 8)
cont = cont.cpu()
dominant = cont.argmax(1)
purity = (cont.max(1).values.double() / cont.sum(1).double())
print(json.dumps({"raster": [size, size, 7], "kmeans_n_iter": int(meta["n_iter"]), "cluster_of_true_class": dominant.tolist(),
                  "one_cluster_per_class": len(set(dominant.tolist())) == 8, "interior_purity_per_class": [round(float(p), 6) for p in purity],
                  "min_purity": round(float(purity.min()), 6)}))
