import sys, os, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "rs-image-segmentation_amd"), os.path.join(ROOT, "tests")]
import torch
from rsseg.runtime import Context
from rsseg import pipeline as P
from test_gpu_fuzz import random_bands
from oracle import ref_np as oracle
ctx = Context(0)
dev = lambda a: ctx.to_device(np.ascontiguousarray(a).reshape(-1))
for seed in (9, 2, 6):
    rng = np.random.default_rng(9900 + seed)
    H, W = int(rng.integers(21, 260)), int(rng.integers(21, 330))
    step = int(rng.choice([1, 1, 2, 7])); k = int(rng.integers(2, 10))
    kind, bands = random_bands(rng, H, W)
    bands = [np.clip(np.round(b), 0, 255) for b in bands]
    coarse = rng.random() < 0.3
    if coarse:
        bands = [np.clip(np.round(b / 32.0), 0, 7) for b in bands]
    d32 = [dev(b.astype(np.float32)) for b in bands]
    d8 = [dev(b.astype(np.uint8)) for b in bands]
    _, _, p32 = P.config3(ctx, d32, H, W, k, 7, step, 3)
    _, _, p8 = P.config3(ctx, d8, H, W, k, 7, step, 3)
    a, b = p8[1].cpu().numpy(), p32[1].cpu().numpy()
    bad = np.where(a.view(np.int32) != b.view(np.int32))[0]
    norm = [oracle.robust_normalize(x.astype(np.float32)) for x in bands]
    want = oracle.calculate_evi(norm[3], norm[2], norm[0]).reshape(-1)
    print(dict(seed=seed, H=H, W=W, kind=kind, coarse=coarse), "differing", bad.size, "u8==oracle", np.array_equal(a.view(np.int32), want.view(np.int32)), "f32==oracle", np.array_equal(b.view(np.int32), want.view(np.int32)))
    for i in bad[:4]:
        print("  px", i, "u8", repr(a[i]), "f32", repr(b[i]), "oracle", repr(want[i]), "bands B,R,N", bands[0].flat[i], bands[2].flat[i], bands[3].flat[i],
              "norm", norm[0].flat[i], norm[2].flat[i], norm[3].flat[i])
    lohi = [(np.percentile(x.astype(np.float32), 2), np.percentile(x.astype(np.float32), 98)) for x in bands[:4]]
    print("  lo/hi of B,G,R,N", lohi)
