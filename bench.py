#!/usr/bin/env python3
"""
bench.py — Mpixel/s of feature-extract + classify on a device-resident 7-band raster (BASELINE.json).

One "step" = one pass of the hot path over the raster already resident in HBM:
  config c3 (default, BASELINE configs[2], the configuration the metric is quoted on):
      percentile normalisation of 7 bands -> 7 spectral indices -> RobustScaler + PCA(3) ->
      GLCM (7x7 window, step 1, 32 levels, 4 angles, 5 properties, bilinear back to H x W) ->
      15 float32 features -> MinMax + KMeans(k=8, k-means++, random_state=42) -> int32 label plane
  config c2 (BASELINE configs[1]): 7 indices -> KMeans(k=6) on a 4096 x 4096 raster.
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); ONE (N*H) x W scene is sharded by rows,
every rank owns H rows (weak scaling) plus the few NIR halo rows its texture windows read; histograms, PCA
sums and KMeans partials go through RCCL, and the label map equals the single-GPU result for that scene.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "rs-image-segmentation_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


SYNTH_STRIPE = 2048


def synth_rows(torch, device, W, g0, g1, want=range(7), bands=7):
    """SURVEY.md §8(d) generator, evaluated on the device: 8 spectral prototypes on a 64-px checkerboard
    + N(0, 6) noise, clipped, truncated to uint8, stored as float32 (integer-valued DN like the real
    preprocessed tile).  Returns the planes `want` for GLOBAL rows [g0, g1); every 2048-row stripe of the scene
    has its own seed and is always generated whole, so any rank reproduces any row of the scene exactly
    (needed for the texture halos of a row-sharded raster)."""
    proto = torch.tensor(np.random.default_rng(355).integers(20, 230, (8, bands)), dtype=torch.float32, device=device)
    want = list(want)
    out = {b: torch.empty((g1 - g0) * W, dtype=torch.float32, device=device) for b in want}
    x = torch.arange(W, device=device)[None, :]
    for s in range(g0 // SYNTH_STRIPE, (g1 - 1) // SYNTH_STRIPE + 1):
        s0 = s * SYNTH_STRIPE
        g = torch.Generator(device=device)
        g.manual_seed(355_000 + s)
        y = (torch.arange(SYNTH_STRIPE, device=device) + s0)[:, None]
        lab = ((y // 64) * 7 + (x // 64) * 3) % 8
        a, b_ = max(g0, s0), min(g1, s0 + SYNTH_STRIPE)
        for b in range(bands):
            noise = torch.randn(SYNTH_STRIPE, W, generator=g, device=device)  # always drawn: keeps the stream aligned
            if b in out:
                v = (proto[lab, b] + noise * 6.0)[a - s0:b_ - s0]
                out[b][(a - g0) * W:(b_ - g0) * W] = v.clamp_(0, 255).to(torch.uint8).to(torch.float32).reshape(-1)
    return [out[b] for b in want]


# algorithmic HBM bytes per pixel and launch of each kernel family
def algorithmic_bytes_per_px(family, F, glcm_step, k=8):
    return {
        "glcm": 4 + 20.0 / (glcm_step * glcm_step),   # read plane once, write 5 property maps
        "lloyd": 4 * F + 8,                            # F float32 features + label read/write
        # k passes: first centre reads F planes; round 1 also writes the closest plane; later rounds read + write it
        "kpp": (4 * F * k + 4 + 8 * max(k - 2, 0)) / max(k, 1),
        "select": 4,                                   # one radix pass over one float32 plane
        "indices": 20 + 28 + 4,                        # 5 bands in, 7 indices + the normalised NIR band out
        "gram": 28, "project": 28 + 12, "resize": 8, "box": 8, "stencil": 8, "forest": 4 * F + 8,
    }[family]


# kernel family -> name of its dominant kernel in profiles/*_pmc_traffic.json (rocprofv3 --pmc passes)
PMC_KERNEL = {"lloyd": "km_lloyd<float, 8, 16, true>", "kpp": "km_kpp<float, 4, 16, 2>", "glcm": "k4_glcm_thread<7, 3>",
              "select": "k1_hist<0, 1024, 4>", "indices": "k2_indices<true>", "gram": "k3_gram<7>", "project": "k3_project<7, true>", "resize": "k5_resize<true>"}


def pmc_traffic_bytes(family, px):
    """HBM bytes per launch of the family's dominant kernel from the committed PMC summary (FETCH_SIZE
    doubled per MI355X_MICROARCH.md + WRITE_SIZE), or None when no summary is available."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files or family not in PMC_KERNEL:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"].get(PMC_KERNEL[family])
        return None if k is None else round((k["read_B_per_px"] + k["write_B_per_px"]) * px)
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(O, tile_bands, H, W, crop, cfg, k):
    """Oracle ("port", single thread) on a crop of the same raster; returns (Mpx/s, description)."""
    from threadpoolctl import threadpool_limits
    c = min(crop, H, W)
    b = [t.reshape(H, W)[:c, :c].cpu().numpy().copy() for t in tile_bands]
    t0 = time.perf_counter()
    with threadpool_limits(limits=1):
        norm = [O.robust_normalize(x) for x in b]
        bl, g, r, n, s = norm[:5]
        feats = [O.calculate_ndvi(n, r), O.calculate_evi(n, r, bl), O.calculate_msavi(n, r), O.calculate_ndwi(g, n),
                 O.calculate_mndwi(g, s), O.calculate_ndbi(s, n), O.calculate_bsi(bl, r, n, s)]
        if cfg == "c3":
            pcs, _, _ = O.perform_pca(norm, n_components=3)
            gl, _ = O.calculate_glcm_features(norm[3], 32, 7, 1)
            feats = feats + [gl[x] for x in ("contrast", "dissimilarity", "homogeneity", "energy", "correlation")] + list(pcs)
        O.kmeans_fit_planes(feats, k)
    dt = time.perf_counter() - t0
    return (c * c / 1e6) / dt, f"{c}x{c}x7 crop of the same synthetic raster, full {cfg} path, oracle C/NumPy port, 1 thread, {dt:.1f} s"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c3", choices=["c2", "c3", "c5"])
    ap.add_argument("--size", type=int, default=0, help="tile edge (default 16384 for c3, 4096 for c2)")
    ap.add_argument("--glcm-step", type=int, default=1)
    ap.add_argument("--cpu-crop", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback for the product path)")
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)
    from rsseg import pipeline as P
    from rsseg.runtime import Context
    ctx = Context(local)

    H = W = args.size or (4096 if args.config == "c2" else 16384)
    k = 8 if args.config == "c3" else 6
    F = {"c2": 7, "c3": 15, "c5": 19}[args.config]
    n_global = H * W * world
    Hg, r0, r1 = H * world, rank * H, (rank + 1) * H       # one (world*H) x W scene, this rank owns rows [r0, r1)
    bands = synth_rows(torch, device, W, r0, r1)
    nir_ext, i0 = None, r0
    if world > 1 and args.config == "c3":
        _, _, i0, i1 = P.glcm_halo_rows(Hg, r0, r1, 7, args.glcm_step)
        top = synth_rows(torch, device, W, i0, r0, want=[3])[0] if i0 < r0 else bands[3][:0]
        bot = synth_rows(torch, device, W, r1, i1, want=[3])[0] if i1 > r1 else bands[3][:0]
        nir_ext = torch.cat([top, bands[3], bot])
    torch.cuda.synchronize()

    forest_model = None
    if args.config == "c5":
        forest_model = fit_c5_forest(ctx, P, bands, H, W, n_global)

    def step():
        if args.config == "c3" and world > 1:
            labels, meta, _ = P.config3_striped(ctx, bands, nir_ext, Hg, W, r0, r1, i0, k, 7, args.glcm_step, 3)
            return labels, meta
        if args.config == "c3":
            return run_c3(ctx, P, bands, H, W, k, args.glcm_step, n_global)
        if args.config == "c5":
            return run_c5(ctx, P, bands, H, W, n_global)
        return run_c2(ctx, P, bands, k, n_global)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    meta = labels = None
    for _ in range(args.warmup):
        labels = None  # release the previous label plane first: the next step reuses its memory instead of growing the pool
        labels, meta = step()
    ctx.prof_enable(True)
    ctx.prof_reset()
    if getattr(ctx, "_aux", None) is not None:
        ctx._aux.prof_enable(True)
        ctx._aux.prof_reset()
    barrier()
    t0 = time.perf_counter()
    verbose = os.environ.get("RSSEG_BENCH_VERBOSE")
    for _ in range(args.steps):
        ts = time.perf_counter()
        labels = None
        labels, meta = step()
        if verbose:  # per-step wall time (adds a device synchronisation per step: diagnostics only)
            torch.cuda.synchronize()
            print(f"[bench] step {(time.perf_counter() - ts) * 1e3:.2f} ms", file=sys.stderr, flush=True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    fams = {}
    for fam in ("glcm", "lloyd", "kpp", "select", "indices", "gram", "project", "resize", "forest", "box", "stencil"):
        ms, cnt = ctx.prof_get(fam)
        if getattr(ctx, "_aux", None) is not None:  # kernels issued on the second stream
            ms2, cnt2 = ctx._aux.prof_get(fam)
            ms, cnt = ms + ms2, cnt + cnt2
        if cnt:
            fams[fam] = (ms, cnt)
    comm_ms, comm_cnt = ctx.prof_get("allreduce")
    ctx.prof_enable(False)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_global / 1e6 / (dt / args.steps)
        dom = max(fams, key=lambda f: fams[f][0]) if fams else None
        roof = None
        if dom:
            ms, cnt = fams[dom]
            per_launch_s = ms / cnt / 1e3
            px = H * W if dom != "glcm" else ((H - 7) // args.glcm_step + 1) * ((W - 7) // args.glcm_step + 1)
            bpp = algorithmic_bytes_per_px(dom, F, args.glcm_step, k)
            achieved = px * bpp / per_launch_s / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 4), "traffic": pmc_traffic_bytes(dom, px) if H == 16384 else None,
                    "algorithmic_bytes": round(px * bpp),
                    "avg_launch_ms": round(ms / cnt, 4), "launches_per_step": cnt / args.steps,
                    "family_ms_per_step": {f: round(v[0] / args.steps, 3) for f, v in fams.items()}}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import ref_np as O
            if args.config == "c5":
                v, desc = cpu_baseline_c5(O, forest_model, bands, H, W, min(args.cpu_crop, 1024))
            else:
                v, desc = cpu_baseline(O, bands, H, W, args.cpu_crop, args.config, k)
            cpu = {"value": round(v, 4), "unit": "Mpixel/s", "cores": 1, "kind": "port", "sample": desc}
        out = {
            "metric": "Mpixel/s feature-extract+classify", "value": round(value, 2), "unit": "Mpixel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{H}x{W}x7 synthetic TM raster per GPU ({H * world}x{W} scene row-striped over {world} GPU(s)), robust-normalise + 7 spectral indices"
                                    + (f" + GLCM(7x7, step {args.glcm_step}, 32 levels, 4 angles) + RobustScaler/PCA(3)" if args.config == "c3" else "")
                                    + (f" -> {F} float32 features -> MinMax + KMeans(k={k}, k-means++, random_state=42)" if args.config != "c5" else
                                       " + PCA + GLCM(21/21) + 7x7 context + morphology/std/Sobel -> 19-feature stack -> RandomForest(100 trees, max_depth 16) inference")),
                       "tile": [H, W, 7], "n_features": F, "n_clusters": k if args.config != "c5" else None,
                       "kmeans_n_iter": int(meta["n_iter"]) if meta else None,
                       "parallelism": f"row-striped x{world}, RCCL all-reduce of histograms / PCA sums / KMeans partials",
                       "allreduce_per_step": comm_cnt / args.steps, "allreduce_host_ms_per_step": round(comm_ms / args.steps, 3)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _label_field(H, W, row0=0):
    y = (np.arange(H) + row0)[:, None]
    x = np.arange(W)[None, :]
    return ((y // 64) * 7 + (x // 64) * 3) % 8


def fit_c5_forest(ctx, P, bands, H, W, n_global):
    """BASELINE config 5: RandomForestClassifier(100, max_depth=16, random_state=42) fitted on the host on
    200 000 pixels of the 19-feature stack with the prototype index as label and 10 % label noise, so that
    depth 16 is reached; same object schema as the reference's rf_samples_model.pkl.  Training is outside
    the timed region (out of scope)."""
    import torch
    from sklearn.ensemble import RandomForestClassifier
    from rsseg.forest import flatten_forest
    planes, _ = P.feature_stack19(ctx, bands, H, W, n_global=n_global)
    fp = P.stack19_forest_planes(ctx, planes)
    rng = np.random.default_rng(355)
    idx = rng.choice(H * W, 200000, replace=False)
    ti = torch.from_numpy(idx).to(fp[0].device)
    X = np.stack([p[ti].cpu().numpy() for p in fp], 1)
    y = _label_field(H, W).reshape(-1)[idx].astype(np.int64)
    flip = rng.random(idx.size) < 0.1
    y[flip] = rng.integers(0, 8, int(flip.sum()))
    model = RandomForestClassifier(n_estimators=100, max_depth=16, random_state=42, n_jobs=-1).fit(X, y)
    ctx.forest_load(flatten_forest(model))
    return model


def run_c5(ctx, P, bands, H, W, n_global):
    planes, _ = P.feature_stack19(ctx, bands, H, W, n_global=n_global)
    labels = ctx.forest_predict(P.stack19_forest_planes(ctx, planes))
    return labels, None


def cpu_baseline_c5(O, model, tile_bands, H, W, crop):
    """Oracle feature stack + the same sklearn forest (n_jobs=None, as the reference calls it) on a crop."""
    from threadpoolctl import threadpool_limits
    c = min(crop, H, W)
    b = [t.reshape(H, W)[:c, :c].cpu().numpy().copy() for t in tile_bands]
    t0 = time.perf_counter()
    with threadpool_limits(limits=1):
        _, hier = O.run_feature_extraction_stage(b)
        model.n_jobs = None
        model.predict(hier["all"].reshape(-1, 19))
    dt = time.perf_counter() - t0
    return (c * c / 1e6) / dt, f"{c}x{c}x7 crop, oracle 19-feature stack + sklearn forest.predict (n_jobs=None), 1 thread, {dt:.1f} s"


def run_c2(ctx, P, bands, k, n_global):
    lohi = P.band_lohi(ctx, bands[:5], n_global)
    idx, _ = P.spectral_indices(ctx, bands, lohi)
    planes = [idx[n] for n in P.INDEX_NAMES]
    return ctx.kmeans_fit_predict(planes, k)


def run_c3(ctx, P, bands, H, W, k, glcm_step, n_global):
    labels, meta, _ = P.config3(ctx, bands, H, W, k, 7, glcm_step, 3, n_global, overlap=os.environ.get("RSSEG_OVERLAP", "0") == "1")
    return labels, meta


if __name__ == "__main__":
    main()
