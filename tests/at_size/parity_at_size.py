#!/usr/bin/env python3
"""Config 3 against the CPU oracle at a size between the test suite's (2048^2) and the benchmark's (16384^2): the raster of
SURVEY.md 8d (oracle.synthetic_raster, NumPy generator) through the product and through the whole CPU oracle path; records
plane-by-plane equality, the PCA deviations and the label agreement (A: literal CPU path with scikit-learn's float32 PCA;
C: CPU path fed the float64-exact components; D: oracle KMeans on the PRODUCT's own 15 planes — the KMeans kernels alone),
with the near-tie proof for differing labels and, where k-means++ drew another seed (D^2 sampling turns a 1e-7 perturbation of
three planes into another pixel once the raster has tens of millions of them), the agreement after the best relabelling.
Usage (GPU box; the oracle is the CHECKER here): python tests/at_size/parity_at_size.py 4096 > gpurun_out/r04/parity_4096.json
       python tests/at_size/parity_at_size.py 16384 kmeans-only > ...   D alone, on the raster bench.py times (generated on the device):
       the full benchmark size — 268 M labels, seeds and iteration count of the KMeans kernels against the oracle on the same planes."""
import json
import os
import sys
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


def _heartbeat():        # the oracle's KMeans is one long C call: keep the log moving (a silent run is taken to be hung)
    import threading

    def beat():
        while True:
            time.sleep(60)
            log("... still running")
    threading.Thread(target=beat, daemon=True).start()


_heartbeat()
ctx = Context(0, use_dist=False)
if len(sys.argv) > 2 and sys.argv[2] == "kmeans-only":
    import torch
    import bench
    dev = bench.synth_rows(torch, ctx.device, W, 0, H)
    labels, meta, planes = P.config3(ctx, dev, H, W, 8, 7, 1, 3)
    got = labels.cpu().numpy()
    gp = [p.cpu().numpy().reshape(H, W) for p in planes]
    del dev, planes, labels
    torch.cuda.empty_cache()
    log("product")
    want, info = O.kmeans_fit_planes(gp, 8)
    log("oracle kmeans (the product's planes)")
    bad = int((got != want).sum())
    print(json.dumps({"raster": f"bench.synth_rows {H}x{W}x7 (the raster bench.py times)", "pixels": H * W,
                      "D_product_vs_oracle_kmeans_on_the_products_planes": {"differing": bad, "same_seeds": [int(x) for x in meta["init_indices"]] == [int(x) for x in info["init_indices"]],
                                                                            "n_iter": [int(meta["n_iter"]), int(info["n_iter"])]},
                      "seconds": round(time.time() - t0, 1)}, indent=1))
    sys.exit(0)
r = O.synthetic_raster(H, W)
log("raster")
dev = [ctx.to_device(np.ascontiguousarray(r[i]).reshape(-1)) for i in range(7)]
labels, meta, planes = P.config3(ctx, dev, H, W, 8, 7, 1, 3)
got = labels.cpu().numpy()
gp = [p.cpu().numpy().reshape(H, W) for p in planes]
del dev, planes, labels
log("product")
norm = [O.robust_normalize(r[i]) for i in range(7)]
b, g, rd, n, s = norm[:5]
feats = [O.calculate_ndvi(n, rd), O.calculate_evi(n, rd, b), O.calculate_msavi(n, rd), O.calculate_ndwi(g, n), O.calculate_mndwi(g, s),
         O.calculate_ndbi(s, n), O.calculate_bsi(b, rd, n, s)]
log("oracle indices")
gl, _ = O.calculate_glcm_features(norm[3], 32, 7, 1)
feats += [gl[x] for x in ("contrast", "dissimilarity", "homogeneity", "energy", "correlation")]
log("oracle glcm")
names = ["ndvi", "evi", "msavi", "ndwi", "mndwi", "ndbi", "bsi", "glcm_contrast", "glcm_dissimilarity", "glcm_homogeneity", "glcm_energy", "glcm_correlation"]
rep = {"raster": f"oracle.synthetic_raster {H}x{W}x7 (SURVEY 8d)", "pixels": H * W,
       "planes_bit_identical": {nm: bool(np.array_equal(gp[i], feats[i])) for i, nm in enumerate(names)}}
X = np.stack([x.reshape(-1) for x in norm], 1).astype(np.float32)
c = np.nanmedian(X, axis=0)
q = np.transpose([np.nanpercentile(X[:, j], (25.0, 75.0)) for j in range(7)])
X = ((X - c) / (q[1] - q[0])).astype(np.float32)
X64 = X.astype(np.float64)
m = X64.mean(0)
C = (X64 - m).T @ (X64 - m) / (X64.shape[0] - 1)
w, V = np.linalg.eigh(C)
Vt = V[:, ::-1].T.copy()
Vt *= np.sign(Vt[np.arange(7), np.argmax(np.abs(Vt), axis=1)])[:, None]
truth = ((X64 - m) @ Vt[:3].T).T
del X, X64
pcs, _, _ = O.perform_pca(norm, n_components=3)
log("oracle pca")
rep["pc_max_abs_dev_from_float64"] = [float(np.abs(gp[12 + i] - truth[i].reshape(H, W)).max()) for i in range(3)]
rep["pc_max_abs_dev_from_sklearn_float32"] = [float(np.abs(gp[12 + i] - pcs[i]).max()) for i in range(3)]
cpu_sk, info_sk = O.kmeans_fit_planes(feats + list(pcs), 8)
log("oracle kmeans (sklearn pca)")
cpu_ex, info_ex = O.kmeans_fit_planes(feats + [truth[i].reshape(H, W).astype(np.float32) for i in range(3)], 8)
log("oracle kmeans (exact pca)")
Xp = np.stack([p.reshape(-1).astype(np.float64) for p in gp], 1)


def diff(a, bb):
    bad = np.nonzero(a != bb)[0]
    out = {"differing": int(bad.size)}
    if bad.size > 1000:      # another seed somewhere: how much of the PARTITION agrees (best one-to-one relabelling)
        from scipy.optimize import linear_sum_assignment
        ct = np.zeros((8, 8), np.int64)
        np.add.at(ct, (a, bb), 1)
        ri, ci = linear_sum_assignment(-ct)
        out["agreement_after_best_relabelling"] = float(ct[ri, ci].sum() / a.size)
        return out
    if bad.size:
        Xs = Xp[bad] * meta["scale"] + meta["min"] - meta["mean"]
        Cc = meta["centers"] - meta["mean"]
        d = ((Xs[:, None, :] - Cc[None, :, :]) ** 2).sum(-1)
        ar = np.arange(bad.size)
        out["max_gap"] = float(np.abs(d[ar, a[bad]] - d[ar, bb[bad]]).max())
        order = np.sort(np.argsort(d, axis=1)[:, :2], axis=1)
        pair = np.sort(np.stack([a[bad], bb[bad]], 1), axis=1)
        out["near_ties"] = int((order == pair).all(1).sum())
    return out


seeds = lambda mm: [int(x) for x in mm["init_indices"]]   # noqa: E731
rep["A_product_vs_cpu_path"] = dict(diff(got, cpu_sk), same_seeds=seeds(meta) == seeds(info_sk), n_iter=[int(meta["n_iter"]), int(info_sk["n_iter"])])
rep["C_product_vs_cpu_path_with_exact_pca"] = dict(diff(got, cpu_ex), same_seeds=seeds(meta) == seeds(info_ex), n_iter=[int(meta["n_iter"]), int(info_ex["n_iter"])])
cpu_own, info_own = O.kmeans_fit_planes(gp, 8)
log("oracle kmeans (the product's planes)")
rep["D_product_vs_oracle_kmeans_on_the_products_planes"] = dict(diff(got, cpu_own), same_seeds=seeds(meta) == seeds(info_own),
                                                                 n_iter=[int(meta["n_iter"]), int(info_own["n_iter"])])
rep["seconds"] = round(time.time() - t0, 1)
print(json.dumps(rep, indent=1))
