#!/usr/bin/env python3
"""
bench.py — Mpixel/s of feature-extract + classify on a device-resident 7-band raster (BASELINE.json).

One "step" = one pass of the hot path over the raster already resident in HBM:
  config c3 (default, BASELINE configs[2], the configuration the metric is quoted on):
      percentile normalisation of 7 bands -> 7 spectral indices -> RobustScaler + PCA(3) ->
      GLCM (7x7 window, step 1, 32 levels, 4 angles, 5 properties, bilinear back to H x W) ->
      15 float32 features -> MinMax + KMeans(k=8, k-means++, random_state=42) -> int32 label plane
  config c2 (BASELINE configs[1]): 7 indices -> KMeans(k=6) on a 4096 x 4096 raster.
  config c5 (BASELINE configs[4]): the 19-feature stack (indices, PCA, 7x7 context means, GLCM 21/21, 5x5 morphological
      gradient, 5x5 local std, Sobel) -> RandomForest(100 trees, max_depth 16) inference -> int64 label plane.
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  ONE H x W raster is sharded by rows over the N ranks
("scaling": "strong" — BASELINE configs[3] / [4]: "16384x16384x7 ... tile-sharded across 8xMI355X"); every rank holds its
stripe plus the halo rows its window operators read; histograms, PCA sums, KMeans partials and the Sobel maximum go through
RCCL, and the label map equals the single-GPU result for that raster bit for bit (tests/test_gpu_dist.py).  --weak keeps H
rows per rank instead (an (N*H) x W scene).

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
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
N_SIMD, CLOCK_HZ = 1024, 2.4e9
N_CU = 256
# The forest walk is bound by dependent LDS gathers: a node visit is two of them (the pixel's feature, then the child
# node).  profiles/ubench/ubench.hip -> profiles/r02_ubench.json, "lds_dependent_chain", 16 waves per CU, 4 chains per wave:
# 545.5 ticks per round of 16 waves x 4 chains x 2 gathers = 4.26 cycles per wave-instruction per CU.
LDS_GATHER_CYCLES = 4.26
FOREST_VISITS_PEAK = N_CU * CLOCK_HZ / LDS_GATHER_CYCLES * 64 / 2


HARD_NOTE = ("the same eight prototypes mixed continuously (abundances following six plane waves of 350..5000-px period, noise N(0, 9)): "
             "smooth gradients and overlapping classes, so that KMeans needs tens of Lloyd iterations like the bundled real scene (47-51)")
HARD = dict(waves=((1 / 2311.0, 1 / 3119.0, 0.3), (1 / 1277.0, -1 / 1913.0, 1.7), (-1 / 811.0, 1 / 1499.0, 2.9), (1 / 523.0, 1 / 677.0, 4.1),
                   (1 / 5003.0, -1 / 4001.0, 5.3), (-1 / 347.0, -1 / 433.0, 0.9)),
            noise=9.0, sharp=1.6)


def synth_rows(torch, device, W, g0, g1, want=range(7), bands=7, kind="easy"):
    """SURVEY.md §8(d) generator, evaluated on the device: 8 spectral prototypes on a 64-px checkerboard
    + N(0, 6) noise, clipped, truncated to uint8, stored as float32 (integer-valued DN like the real
    preprocessed tile).  Returns the planes `want` for GLOBAL rows [g0, g1); every 2048-row stripe of the scene
    has its own seed and is always generated whole, so any rank reproduces any row of the scene exactly
    (needed for the halos of a row-sharded raster).  The noise comes from torch's device generator (Philox, seeded per
    stripe): the same rows on any rank of one run and on any MI355X with this torch build; a different torch / rocRAND
    build may draw different noise — the bench compares ranks and kernels within one run, the parity tests use
    oracle.synthetic_raster (NumPy default_rng, as §8(d) writes it).
    kind="hard": the same prototypes MIXED continuously — abundances that follow six plane waves of 350..5000-px period
    across the scene, noise N(0, 9) — i.e. smooth gradients and overlapping classes instead of eight separable ones:
    KMeans then needs tens of Lloyd iterations, like the bundled real scene (47-51), not 6."""
    proto = torch.tensor(np.random.default_rng(355).integers(20, 230, (8, bands)), dtype=torch.float32, device=device)
    want = list(want)
    out = {b: torch.empty((g1 - g0) * W, dtype=torch.float32, device=device) for b in want}
    x = torch.arange(W, device=device)[None, :]
    for s in range(g0 // SYNTH_STRIPE, (g1 - 1) // SYNTH_STRIPE + 1):
        s0 = s * SYNTH_STRIPE
        g = torch.Generator(device=device)
        g.manual_seed(355_000 + s)
        y = (torch.arange(SYNTH_STRIPE, device=device) + s0)[:, None]
        a, b_ = max(g0, s0), min(g1, s0 + SYNTH_STRIPE)
        if kind == "hard":
            xf, yf = x.to(torch.float32), y.to(torch.float32)
            # eight abundance fields from six plane waves; softmax-like mixing keeps every DN inside the prototypes' hull
            ph = [torch.cos(6.2831855 * (fx * xf + fy * yf) + p0) for fx, fy, p0 in HARD["waves"]]
            mix = torch.tensor(np.random.default_rng(77).normal(0, 1.0, (8, len(ph))), dtype=torch.float32, device=device)
            wts = torch.stack([sum(mix[i, j] * ph[j] for j in range(len(ph))) for i in range(8)])
            wts = torch.softmax(HARD["sharp"] * wts, dim=0)
            del ph
        else:
            lab = ((y // 64) * 7 + (x // 64) * 3) % 8
        for b in range(bands):
            noise = torch.randn(SYNTH_STRIPE, W, generator=g, device=device)  # always drawn: keeps the stream aligned
            if b in out:
                if kind == "hard":
                    v = ((wts * proto[:, b][:, None, None]).sum(0) + noise * HARD["noise"])[a - s0:b_ - s0]
                else:
                    v = (proto[lab, b] + noise * 6.0)[a - s0:b_ - s0]
                out[b][(a - g0) * W:(b_ - g0) * W] = v.clamp_(0, 255).to(torch.uint8).to(torch.float32).reshape(-1)
    return [out[b] for b in want]


# ---- roofline bookkeeping --------------------------------------------------------------------------------------------
def family_table(cfg, F, glcm_step, k, n_pca):
    """kernel family -> (bound, algorithmic HBM bytes per pixel and launch) — SURVEY.md §8(d), DESIGN.md §5.
    'lloyd' is charged the F float32 planes an update sweep reads (r04: the sweeps of the device-resident loop keep no label
    plane; the final E-step, one launch in seven here, also writes 4 B/px of int32 labels)."""
    idx_out = 7 * 4 + (4 if cfg in ("c3", "c5") else 0)   # + the normalised NIR band the texture chain reads
    return {
        "kpp": ("hbm", (4 * F * k + 4 + 8 * max(k - 2, 0)) / max(k, 1)),   # k passes: F planes in, closest plane r/w from round 2 on
        "lloyd": ("hbm", 4 * F),
        "moment": ("hbm", 4 * F),                                          # column means of the scaled matrix: F planes in
        "labels": ("hbm", 5),                                              # uint8 labels in, int32 labels out
        "select": ("hbm", 4),                                              # one radix pass over one float32 plane
        "indices": ("hbm", 20 + idx_out),
        "normalize": ("hbm", 8), "quantize": ("hbm", 5),
        "gram": ("hbm", 28), "project": ("hbm", 28 + 4 * n_pca),
        "indices_project": ("hbm", 28 + idx_out + 4 * n_pca),             # fused: 7 raw bands in; 7 indices (+ normalised NIR) + the components out
        "resize": ("hbm", 8),                                              # 4 taps from cache, one plane out
        "box": ("hbm", 8), "ctxmean": ("hbm", 8 * 7), "morph": ("hbm", 2), "filt_max": ("hbm", 1), "filt_write": ("hbm", 5),
        "glcm": ("valu", 4 + 20.0 / (glcm_step * glcm_step)),
        "forest": ("lds_gather", 4 * F + 8),
    }


# MFMA: the only dense contraction of the path is the 7x7 Gram / 7x3 projection of the PCA (3.5 flop/B).  It runs on the vector
# ALUs with exact fixed-point accumulation; no MFMA instruction is issued by any kernel of the step (rocprofv3 --pmc
# SQ_INSTS_VALU_MFMA_* = 0 for every kernel: profiles/r04_c3_pmc_sq.md), so the utilisation north_star asks to report is 0.
MFMA_UTIL = {"value": 0.0, "note": "no MFMA instruction in the step (SQ_VALU_MFMA_BUSY_CYCLES = 0 for every kernel, profiles/r04_c3_pmc_sq.md); "
                                   "the PCA Gram / projection (2 x 7 x 7 flop per 28 B) run on the VALU with exact fixed-point sums — "
                                   "DESIGN.md 5: an MFMA accumulates in floating point, and even at the f64 matrix peak the 16 x 16 x 4 tiles of a 7-band Gram (137 GFLOP padded) take 1.75 ms against 1.3 ms for the kernel that exists"}

# dominant kernel of a family in profiles/*_pmc_traffic.json (rocprofv3 --pmc passes)
PMC_KERNEL = {"lloyd": "km_lloyd<float, 8, 16, true>", "kpp": "km_kpp<float, 4, 16, 2>", "glcm": "k4_glcm_quad",
              "select": "k1_hist<3, 1024, 4>", "indices": "k2_indices<true", "gram": "k3_gram<7", "project": "k3_project<7, true", "indices_project": "k3_indices_project<7, true",
              "resize": "k5_resize<true>", "forest": "k11_forest", "ctxmean": "k6_box<7, false>"}
# static VALU instructions of the texture kernels and their measured issue cost (profiles/valu_mix.py ->
# profiles/r*_valu_issue.json, the newest): the bound of the texture kernel is VALU issue, not HBM.  Dense case (window 7, step 1):
# k4_glcm_quad, 256 windows per wave (r02 / r03: k4_glcm_pair, 128); other steps: k4_glcm_thread<7,3>, 64 windows per wave.
GLCM_VALU = {"insts_per_wave": 2954, "issue_cycles_per_wave": None}


def pmc_traffic_bytes(family, px):
    """HBM bytes per launch of the family's dominant kernel from the newest committed PMC summary (FETCH_SIZE
    doubled per MI355X_MICROARCH.md + WRITE_SIZE), or None when that kernel is not in a summary."""
    import glob
    if family not in PMC_KERNEL:
        return None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            ks = json.load(open(f))["kernels"]
        except Exception:  # noqa: BLE001
            continue
        for name, kk in ks.items():
            if name.startswith(PMC_KERNEL[family]):
                return round((kk["read_B_per_px"] + kk["write_B_per_px"]) * px)
    return None


def pattern_rate_gbs(family):
    """Measured rate (GB/s) of the family's bare access pattern on MI355X — 15 float32 planes read with 16-byte loads, plus
    one read-modify-write plane for the k-means kernels (profiles/r02_streams.json, best of the recorded runs) — or None."""
    try:
        if family == "indices_project":   # 28 B/px read over 7 planes + 41 B/px written over 11 (profiles/ubench/streams.hip write_heavy, r04):
            wh = json.load(open(os.path.join(ROOT, "profiles", "r04_streams_write_heavy.json")))   # the best launch shape recorded
            return round(max(v["TBs"] for v in wh.values() if isinstance(v, dict) and "TBs" in v and not v.get("other_mix")) * 1000.0, 1)
        if family in ("resize", "ctxmean", "box"):   # as many float32 bytes out as in: the best bare plane copy recorded (r04_streams_copy.json)
            cp = json.load(open(os.path.join(ROOT, "profiles", "r04_streams_copy.json")))
            return round(max(v["TBs"] for v in cp.values() if isinstance(v, dict)) * 1000.0, 1)
        if family == "select":            # one float32 plane read once: the small-integer pass at its best grid (r04_k1_sweep.json) is the pattern
            return round(max(r["read_only"]["1"] for r in json.load(open(os.path.join(ROOT, "profiles", "r02_streams.json")))["runs"]) * 1000.0, 1)
        runs = json.load(open(os.path.join(ROOT, "profiles", "r02_streams.json")))["runs"]
        key, ns = {"kpp": ("with_rw_plane", "15"), "lloyd": ("read_only", "15"), "moment": ("read_only", "1")}[family]
        return round(max(r[key][ns] for r in runs if key in r) * 1000.0, 1)
    except Exception:  # noqa: BLE001
        return None


def glcm_issue_cycles(glcm_step=1):
    """(weighted VALU issue cycles per wave, windows per wave, static VALU instructions per wave, kernel) of the texture
    kernel that runs at this step, from the committed microbenchmark summary; cycles None when it is absent."""
    key, kern = ("glcm_quad", "k4_glcm_quad") if glcm_step == 1 else ("glcm_thread_7_3", "k4_glcm_thread<7,3>")
    if glcm_step == 1 and os.environ.get("RSSEG_GLCM_DENSE") == "pair":
        key, kern = "glcm_pair", "k4_glcm_pair"
    import glob
    try:
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_issue.json")))[-1]   # the newest round's histogram of the code object
        e = json.load(open(f))[key]
        return e["issue_cycles_per_wave"], 64 * e["windows_per_thread"], e["valu_static"], kern
    except Exception:  # noqa: BLE001
        return None, 64, GLCM_VALU["insts_per_wave"], kern


# ---- CPU baseline (rank 0, N = 1) -------------------------------------------------------------------------------------
def cpu_baseline(tile_bands, H, W, crop, cfg, k, glcm_step, model=None):
    """the CPU counterpart harness (oracle/cpu_harness.py: NumPy glue restated + the scikit-learn
    estimators the reference calls + C restatements of the cv2 / skimage stages) on a crop of the same raster, once with
    every host core and once with one thread."""
    from oracle import cpu_harness as CH
    c = min(crop, H, W)
    b = [t.reshape(H, W)[:c, :c].cpu().numpy().copy() for t in tile_bands]
    px = c * c / 1e6
    if cfg == "c5":
        allc = CH.config5(b, model, None)
        one = CH.config5(b, model, 1)
        tot_all, tot_one = allc["features"] + allc["forest"], one["features"] + one["forest"]
        stages = {"all_cores": {s: round(v, 3) for s, v in allc.items()}, "one_thread": {s: round(v, 3) for s, v in one.items()}}
    else:
        allc = CH.config23(b, cfg, k, glcm_step, None)
        one = CH.config23(b, cfg, k, glcm_step, 1)
        keys = [s for s in allc if s != "n_iter"]
        tot_all, tot_one = sum(allc[s] for s in keys), sum(one[s] for s in keys)
        stages = {"all_cores": {s: round(allc[s], 3) for s in keys}, "one_thread": {s: round(one[s], 3) for s in keys},
                  "kmeans_n_iter": allc["n_iter"]}
    ti = CH.thread_info()
    threads = max([p["threads"] or 1 for p in ti["pools"]] + [1])
    return {"value": round(px / tot_all, 4), "unit": "Mpixel/s", "cores": threads, "kind": "port",
            "host_cores": ti["host_cores"], "threads": threads, "threadpools": ti["pools"],
            "value_all_cores": round(px / tot_all, 4), "value_1_thread": round(px / tot_one, 4), "stage_seconds": stages,
            "sample": (f"{c}x{c}x7 crop of the same synthetic raster, full {cfg} path: NumPy glue restated + scikit-learn "
                       "RobustScaler/PCA/MinMaxScaler/KMeans" + ("/RandomForest.predict" if cfg == "c5" else "")
                       + " + C/NumPy restatements of the cv2/skimage stages"
                       + f"; extrapolated linearly in pixels to the full raster; {tot_all:.1f} s all cores, {tot_one:.1f} s one thread")}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` run bare (no torchrun): start N FRESH child processes of this script, one per GPU, with
    the rendezvous environment torch.distributed.run would give them, wait for all of them, relay rank 0's single JSON
    line and return a non-zero code if any rank failed.  Nothing in this (parent) process has touched a GPU: torch is
    not even imported yet, and the children are new processes, not an exec of an initialised one."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RSSEG_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    # rank 0's stdout is drained by a thread while ALL children are watched: a rank that dies (out of memory, a GPU fault)
    # leaves the others inside a collective, so on the first non-zero exit — or when the overall limit expires — the rest are
    # terminated (exact PIDs, started here) and the failure is returned instead of a hang
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    limit = float(os.environ.get("RSSEG_BENCH_LAUNCH_TIMEOUT", "3000"))
    t_start = time.time()
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [i for i, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.time() - t_start > limit:
            failed = f"rank(s) {bad} failed" if bad else f"no result after {limit:.0f} s"
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    out0 = buf[0] if buf else ""
    codes = [p.wait() for p in procs]
    if failed:
        print(f"bench.py: {failed}; the other ranks were stopped", file=sys.stderr)
    lines = [ln for ln in (out0 or "").splitlines() if ln.strip()]
    json_lines = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in json_lines[-1:]:
            print(ln, file=sys.stderr)
    if any(codes) or not json_lines:
        print(f"bench.py: ranks exited with {codes}" + ("" if json_lines else "; rank 0 printed no JSON line"), file=sys.stderr)
        return max([abs(c) for c in codes if c] + [1])
    print(json_lines[-1], flush=True)
    return 0


def launch_selftest():
    """--launch-selftest: what a rank does to prove the launch without a GPU — rendezvous over gloo on the CPU, one
    all-reduce, rank 0 prints the line (used by tests/test_host.py)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    total = 1.0
    if os.environ.get("RSSEG_SELFTEST_DIE_RANK") == str(rank):    # a rank that dies before the rendezvous (tests the launcher's watch)
        return 7
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        total = float(t.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "sum_of_rank_ids_plus_1": total}), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c3", choices=["c2", "c3", "c5"])
    ap.add_argument("--size", type=int, default=0, help="raster edge (default 16384 for c3 / c5, 4096 for c2)")
    ap.add_argument("--glcm-step", type=int, default=1)
    ap.add_argument("--data", default="easy", choices=["easy", "hard"],
                    help="synthetic raster: 'easy' = SURVEY 8d's eight separable prototypes (KMeans converges in ~6 iterations), "
                         "'hard' = continuously mixed prototypes (tens of iterations, like the bundled real scene)")
    ap.add_argument("--cpu-crop", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive and GLCM-step-7 side measurements")
    ap.add_argument("--no-c5-extra", action="store_true", help="c3 at 16384: skip the config-5 side measurement (forest fit on the host + two forest steps)")
    ap.add_argument("--weak", action="store_true", help="N > 1: H rows per rank ((N*H) x W scene) instead of one H x W raster split N ways")
    ap.add_argument("--overlap", action="store_true", help="c3, N = 1: GLCM chain on a second HIP stream (profiles/r01_overlap_note.md)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--rccl-single", action="store_true", help="N = 1: a one-rank RCCL group with the all-reduce hook installed, so that every collective of a step (identity reductions) goes through torch.distributed / RCCL on this GPU")
    ap.add_argument("--comm", choices=["native", "torch"], default=None,
                    help="how the all-reduces are issued: 'native' = the library's own RCCL communicator (ncclAllReduce from C on the context's "
                         "stream; default with backend nccl), 'torch' = the callback into torch.distributed.all_reduce (default with gloo)")
    ap.add_argument("--launch-selftest", action="store_true", help="no GPU: every rank joins a gloo rendezvous on the CPU and rank 0 prints one line")
    args = ap.parse_args()

    # ---- N ranks: either torch.distributed.run started us (RANK / WORLD_SIZE set, WORLD_SIZE must equal --gpus), or
    # this process starts them itself.  Decided BEFORE torch is imported or a GPU is touched.
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch with --nproc-per-node {args.gpus}, "
                         "or run bench.py bare and let it start the ranks")
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.launch_selftest:
        raise SystemExit(launch_selftest())

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
    if args.rccl_single and world == 1:
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            free_port = s_.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)
    from rsseg import pipeline as P
    from rsseg.runtime import Context
    ctx = Context(local, force_comm=args.rccl_single and world == 1, comm=args.comm)

    cfg = args.config
    H = W = args.size or (4096 if cfg == "c2" else 16384)
    k = 8 if cfg == "c3" else 6
    F = {"c2": 7, "c3": 15, "c5": 19}[cfg]
    n_pca = {"c2": 0, "c3": 3, "c5": 7}[cfg]
    weak = args.weak and world > 1
    Hg = H * world if weak else H                      # rows of the ONE raster all ranks work on
    r0, r1 = (rank * H, (rank + 1) * H) if weak else P.stripe_rows(Hg, world, rank)
    n_global = Hg * W
    n_own = (r1 - r0) * W

    # ---- this rank's resident input: its stripe, plus halo rows when the raster is sharded ----
    nir_ext, i0, e0 = None, r0, r0
    if cfg == "c5" and world > 1:
        e0, e1 = P.stack19_halo_rows(Hg, r0, r1)
        bands = synth_rows(torch, device, W, e0, e1, kind=args.data)      # rows [e0, e1) of every band
    else:
        bands = synth_rows(torch, device, W, r0, r1, kind=args.data)
        if cfg == "c3" and world > 1:
            _, _, i0, i1 = P.glcm_halo_rows(Hg, r0, r1, 7, args.glcm_step)
            top = synth_rows(torch, device, W, i0, r0, want=[3], kind=args.data)[0] if i0 < r0 else bands[3][:0]
            bot = synth_rows(torch, device, W, r1, i1, want=[3], kind=args.data)[0] if i1 > r1 else bands[3][:0]
            nir_ext = torch.cat([top, bands[3], bot])
    torch.cuda.synchronize()

    forest_model = None
    if cfg == "c5":
        forest_model = fit_c5_forest(torch, dist, device, P, rank, world, W)
        ctx.forest_load(forest_model["flat"])

    def step(glcm_step=args.glcm_step, bands=bands):
        if cfg == "c3" and world > 1:
            labels, meta, _ = P.config3_striped(ctx, bands, nir_ext, Hg, W, r0, r1, i0, k, 7, glcm_step, 3)
            return labels, meta
        if cfg == "c3":
            labels, meta, _ = P.config3(ctx, bands, H, W, k, 7, glcm_step, 3, n_global, overlap=args.overlap)
            return labels, meta
        if cfg == "c5" and world > 1:
            planes, _ = P.stack19_striped(ctx, bands, Hg, W, r0, r1, e0)
            return ctx.forest_predict(P.stack19_forest_planes(ctx, planes)), None
        if cfg == "c5":
            planes, _ = P.feature_stack19(ctx, bands, H, W, n_global=n_global)
            return ctx.forest_predict(P.stack19_forest_planes(ctx, planes)), None
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
    ctx.host_syncs(reset=True)
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
    for fam in family_table(cfg, F, args.glcm_step, k, n_pca):
        ms, cnt = ctx.prof_get(fam)
        if getattr(ctx, "_aux", None) is not None:  # kernels issued on the second stream
            ms2, cnt2 = ctx._aux.prof_get(fam)
            ms, cnt = ms + ms2, cnt + cnt2
        if cnt:
            fams[fam] = (ms, cnt)
    comm_ms, comm_cnt = ctx.prof_get("allreduce")
    host_syncs = ctx.host_syncs()
    ctx.prof_enable(False)

    # ---- side measurements (rank 0, N = 1): GLCM step 7, the other synthetic raster, north_star's literal configuration,
    # and the PCIe-inclusive rates — never the bench `value` ----
    extras = {}

    def timed(fn, reps=2):
        nonlocal labels
        labels = None
        fn()                                   # warm
        torch.cuda.synchronize()
        ctx.prof_enable(True)
        ctx.prof_reset()
        ts = time.perf_counter()
        m = None
        for _ in range(reps):
            labels = None
            labels, m = fn()
        torch.cuda.synchronize()
        t = (time.perf_counter() - ts) / reps
        lms, lcnt = ctx.prof_get("lloyd")
        kms, _ = ctx.prof_get("kpp")
        ctx.prof_enable(False)
        out = {"ms_per_step": round(t * 1e3, 2), "value": round(n_global / 1e6 / t, 2)}
        if m:
            out.update({"kmeans_n_iter": int(m["n_iter"]), "ms_per_lloyd_iter": round(lms / max(lcnt, 1), 3), "ms_kpp": round(kms / reps, 2),
                        "ms_lloyd": round(lms / reps, 2)})
        return out

    if world == 1 and not args.no_extras:
        if cfg == "c3" and args.glcm_step == 1:
            extras["glcm_step7"] = dict(timed(lambda: step(7)), note="same path with step_size=7 (the reference's non-overlapping windows, SURVEY.md 8d)")
        if cfg in ("c2", "c3"):
            other = "hard" if args.data == "easy" else "easy"
            ob = synth_rows(torch, device, W, r0, r1, kind=other)
            extras[f"{other}_raster"] = dict(timed(lambda: step(bands=ob), reps=2),
                                             note=HARD_NOTE if other == "hard" else "SURVEY.md 8d raster: eight separable prototypes on a 64-px checkerboard")
            del ob
        if cfg in ("c2", "c3"):   # the same step with the bands resident as the 8-bit digital numbers they are (1 B/px inside K1 / K2 / K3)
            b8 = [b.to(torch.uint8) for b in bands]
            extras["uint8_resident"] = dict(timed(lambda: step(bands=b8)),
                                            note="same raster, same labels, the 7 bands resident as uint8 planes (rsseg_*_u8 entry points: order statistics, "
                                                 "indices and PCA read 1 byte per pixel; normalisation / scaling through 256-entry tables)")
            del b8
        if cfg == "c3" and H == 16384:
            extras["north_star_c2_16384"] = dict(timed(lambda: run_c2(ctx, P, bands, 6, n_global)),
                                                 note="BASELINE.json north_star's literal target configuration: 7 spectral indices + KMeans(k=6) on the 16384x16384x7 raster, 1 GPU (target: >= 100 Mpixel/s)")
        if cfg == "c3" and H == 16384 and not args.no_c5_extra:
            extras["c5_16384"] = c5_extra(torch, dist, device, ctx, P, bands, H, W, n_global)
        step_qb = None
        if cfg == "c3":
            def step_qb(dev_bands, qb, tex, texture_only=False):
                if texture_only:
                    return P.texture_planes(ctx, dev_bands[3], qb[3], H, W, 7, args.glcm_step)
                lab, m, _ = P.config3(ctx, dev_bands, H, W, k, 7, args.glcm_step, 3, n_global, qb=qb, glcm=tex)
                return lab, m
        extras["pcie_inclusive"] = pcie_inclusive(torch, device, ctx, bands, step, n_global, step_qb)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_global / 1e6 / (dt / args.steps)
        table = family_table(cfg, F, args.glcm_step, k, n_pca)
        kernels = []
        visits = (forest_visits_per_px(torch, P, ctx, bands, H, W, n_global, forest_model["model"])
                  if (cfg == "c5" and world == 1 and "forest" in fams) else None)
        for fam, (ms, cnt) in sorted(fams.items(), key=lambda kv: -kv[1][0]):
            bound, bpp = table[fam]
            per_launch_s = ms / cnt / 1e3
            px = n_own if fam != "glcm" else max((r1 - r0 - 7) // args.glcm_step + 1, 1) * ((W - 7) // args.glcm_step + 1)
            ent = {"name": fam, "bound": bound, "avg_ms": round(ms / cnt, 4), "launches_per_step": cnt / args.steps,
                   "ms_per_step": round(ms / args.steps, 3), "algorithmic_B_per_px": round(bpp, 2),
                   "hbm_GBs": round(px * bpp / per_launch_s / 1e9, 1), "hbm_frac": round(px * bpp / per_launch_s / 1e9 / HBM_PEAK_GBS, 4)}
            if bound == "hbm":
                ent["frac"] = ent["hbm_frac"]
                rate = pattern_rate_gbs(fam)
                if rate:   # what a kernel with nothing but this access pattern reaches (profiles/ubench/streams.hip)
                    ent["pattern_rate_GBs"] = rate
                    ent["frac_of_pattern_rate"] = round(ent["hbm_GBs"] / rate, 4)
            elif bound == "valu" and cfg != "c3":
                ent["frac"] = None      # window 21 / step 21 runs the workgroup-per-window kernel: no instruction model for it
            elif bound == "valu":
                cyc, win_per_wave, insts, kern = glcm_issue_cycles(args.glcm_step)
                waves = px / float(win_per_wave)
                if cyc is not None:
                    ent["frac"] = round(waves * cyc / (N_SIMD * CLOCK_HZ * per_launch_s), 4)
                    ent["valu_issue_cycles_per_wave"] = cyc
                else:  # nominal 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md) when the summary is missing
                    ent["frac"] = round(waves * insts * 2 / (N_SIMD * CLOCK_HZ * per_launch_s), 4)
                ent["valu_insts_per_wave"] = insts
                ent["windows_per_wave"] = win_per_wave
                ent["kernel"] = kern
            elif bound == "lds_gather" and visits:
                ent["node_visits_per_px"] = round(visits, 1)
                ent["node_visits_per_s"] = round(px * visits / per_launch_s, 0)
                ent["peak_node_visits_per_s"] = round(FOREST_VISITS_PEAK, 0)
                ent["frac"] = round(ent["node_visits_per_s"] / FOREST_VISITS_PEAK, 4)
                ent["bound_note"] = (f"a node visit = 2 dependent LDS gathers; {LDS_GATHER_CYCLES} cycles per gather wave-instruction per CU "
                                     "(profiles/r02_ubench.json) x 256 CUs x 2.4 GHz x 64 lanes / 2")
            kernels.append(ent)
        roof = None
        if kernels:
            dom = kernels[0]                      # the family with the most time per step LEADS the roofline object
            px = n_own
            if dom["bound"] == "lds_gather" and dom.get("frac") is not None:
                roof = {"bound": "lds_gather", "kernel": dom["name"], "achieved": dom["node_visits_per_s"], "peak": dom["peak_node_visits_per_s"],
                        "unit": "node visits/s", "frac": dom["frac"], "traffic": pmc_traffic_bytes(dom["name"], px) if (H == 16384 and world == 1) else None,
                        "algorithmic_bytes": round(px * dom["algorithmic_B_per_px"]), "hbm_frac": dom["hbm_frac"], "avg_launch_ms": dom["avg_ms"]}
            else:
                hbm = [e for e in kernels if e["bound"] == "hbm"]
                lead = dom if dom["bound"] == "hbm" or not hbm else hbm[0]
                roof = {"bound": lead["bound"], "kernel": lead["name"], "achieved": lead["hbm_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": lead["hbm_frac"], "traffic": pmc_traffic_bytes(lead["name"], px) if (H == 16384 and world == 1) else None,
                        "algorithmic_bytes": round(px * lead["algorithmic_B_per_px"]), "avg_launch_ms": lead["avg_ms"]}
            roof["dominant_by_time"] = dom["name"]
            # north_star: "MFMA used only for the PCA covariance/projection GEMM, with rocprof-reported ... MFMA utilisation"
            roof["mfma_util"] = MFMA_UTIL["value"]
            roof["mfma_note"] = MFMA_UTIL["note"].replace("r04_c3_pmc_sq.md", "r04_c5_pmc_sq.md") if cfg == "c5" else MFMA_UTIL["note"]
            roof["kernels"] = kernels
            roof["kernels_ms_per_step"] = round(sum(e["ms_per_step"] for e in kernels), 2)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(bands, H, W, min(args.cpu_crop, 1024) if cfg == "c5" else args.cpu_crop, cfg, k, args.glcm_step,
                               forest_model["model"] if forest_model else None)
        scene = f"{Hg}x{W}x7 synthetic TM raster" + (f", row-striped over {world} GPUs ({r1 - r0} rows per rank + halo rows)" if world > 1 else "")
        out = {
            "metric": "Mpixel/s feature-extract+classify", "value": round(value, 2), "unit": "Mpixel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (scene + (" (continuously mixed prototypes: --data hard)" if args.data == "hard" else "") + ", robust-normalise + 7 spectral indices"
                                    + (f" + GLCM(7x7, step {args.glcm_step}, 32 levels, 4 angles) + RobustScaler/PCA(3)" if cfg == "c3" else "")
                                    + (f" -> {F} float32 features -> MinMax + KMeans(k={k}, k-means++, random_state=42)" if cfg != "c5" else
                                       " + PCA + GLCM(21/21) + 7x7 context + morphology/std/Sobel -> 19-feature stack -> RandomForest(100 trees, max_depth 16) inference")),
                       "baseline_config": {"c2": "configs[1]", "c3": "configs[2]" if world == 1 else "configs[3]", "c5": "configs[4]"}[cfg],
                       "raster": [Hg, W, 7], "rows_per_rank": r1 - r0, "n_features": F, "n_clusters": k if cfg != "c5" else None,
                       "kmeans_n_iter": int(meta["n_iter"]) if meta else None,
                       "ms_per_lloyd_iter": round(fams["lloyd"][0] / fams["lloyd"][1], 3) if "lloyd" in fams else None,
                       "ms_kpp_per_step": round(fams["kpp"][0] / args.steps, 2) if "kpp" in fams else None,
                       "data_kind": args.data,
                       "parallelism": (f"one raster row-striped x{world}, {'RCCL' if args.backend == 'nccl' else args.backend} all-reduce of histograms / PCA sums / KMeans partials / Sobel max"
                                       if world > 1 else ("single GPU, every collective through a one-rank RCCL group" if args.rccl_single else "single GPU")),
                       "host_syncs_per_step": round(host_syncs / args.steps, 1),
                       "allreduce_per_step": comm_cnt / args.steps, "allreduce_host_ms_per_step": round(comm_ms / args.steps, 3),
                       "comm": ctx.comm_kind and {"native": "RCCL from C (rsseg_ctx_set_comm_rccl: ncclAllReduce on the context's stream)",
                                                  "torch": "callback into torch.distributed.all_reduce"}[ctx.comm_kind],
                       "allreduce_host_us_per_collective": round(1e3 * comm_ms / comm_cnt, 1) if comm_cnt else None,
                       **extras},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def _label_field(H, W, row0=0):
    y = (np.arange(H) + row0)[:, None]
    x = np.arange(W)[None, :]
    return ((y // 64) * 7 + (x // 64) * 3) % 8


def fit_c5_forest(torch, dist, device, P, rank, world, W):
    """BASELINE config 5: RandomForestClassifier(100, max_depth=16, random_state=42) fitted on the host on 200 000 pixels
    of the 19-feature stack of the first 2048 rows of the scene (prototype index as label, 10 % label noise, so that
    depth 16 is reached); same object schema as the reference's rf_samples_model.pkl.  Rank 0 fits and broadcasts the
    flattened forest, so every rank walks the same trees whatever the sharding.  Training is outside the timed region
    (out of scope)."""
    from rsseg.forest import flatten_forest
    from rsseg.runtime import Context
    model = flat = None
    if rank == 0:
        from sklearn.ensemble import RandomForestClassifier
        Ht = min(2048, W)
        c0 = Context(device.index, use_dist=False)
        tb = synth_rows(torch, device, W, 0, Ht)
        planes, _ = P.feature_stack19(c0, tb, Ht, W)
        fp = P.stack19_forest_planes(c0, planes)
        rng = np.random.default_rng(355)
        idx = rng.choice(Ht * W, 200000, replace=False)
        ti = torch.from_numpy(idx).to(device)
        X = np.stack([p[ti].cpu().numpy() for p in fp], 1)
        y = _label_field(Ht, W).reshape(-1)[idx].astype(np.int64)
        flip = rng.random(idx.size) < 0.1
        y[flip] = rng.integers(0, 8, int(flip.sum()))
        model = RandomForestClassifier(n_estimators=100, max_depth=16, random_state=42, n_jobs=-1).fit(X, y)
        flat = flatten_forest(model)
        c0.close()
        del tb, planes, fp
        torch.cuda.empty_cache()
    if world > 1:
        box = [flat]
        dist.broadcast_object_list(box, src=0)
        flat = box[0]
    return {"model": model, "flat": flat}


def forest_visits_per_px(torch, P, ctx, bands, H, W, n_global, model, sample=65536):
    """Average number of tree nodes a pixel visits over the whole forest (leaf included), from sklearn's own apply() on
    a sample of the feature rows the GPU walks: turns the forest kernel's time into node visits per second."""
    try:
        planes, _ = P.feature_stack19(ctx, bands, H, W, n_global=n_global)
        fp = P.stack19_forest_planes(ctx, planes)
        idx = torch.from_numpy(np.random.default_rng(1).choice(H * W, sample, replace=False)).to(fp[0].device)
        X = np.stack([p[idx].cpu().numpy() for p in fp], 1)
        tot = 0.0
        for est in model.estimators_:
            t = est.tree_
            left, right = t.children_left, t.children_right
            depth = np.zeros(t.node_count, np.int32)
            for nd in range(t.node_count):                  # parents precede children in sklearn's node order
                if left[nd] != -1:
                    depth[left[nd]] = depth[nd] + 1
                    depth[right[nd]] = depth[nd] + 1
            tot += float((depth[t.apply(X)] + 1).mean())
        return tot
    except Exception:  # noqa: BLE001
        return None


def c5_extra(torch, dist, device, ctx, P, bands, H, W, n_global):
    """BASELINE configs[4] on one GPU as a side figure of the default line: the 19-feature stack + RandomForest(100 trees,
    max_depth 16) inference on the same 16384^2 raster.  The forest is fitted on the host OUTSIDE the timed region (as in
    `--config c5`); two timed steps."""
    try:
        t_fit = time.perf_counter()
        fm = fit_c5_forest(torch, dist, device, P, 0, 1, W)
        t_fit = time.perf_counter() - t_fit
        ctx.forest_load(fm["flat"])

        def st():
            planes, _ = P.feature_stack19(ctx, bands, H, W, n_global=n_global)
            return ctx.forest_predict(P.stack19_forest_planes(ctx, planes))

        lab = st()
        del lab
        torch.cuda.synchronize()
        ctx.prof_enable(True)
        ctx.prof_reset()
        ctx.host_syncs(reset=True)
        reps = 2
        t0 = time.perf_counter()
        for _ in range(reps):
            lab = st()
            del lab
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
        fms, fcnt = ctx.prof_get("forest")
        syncs = ctx.host_syncs()
        ctx.prof_enable(False)
        visits = forest_visits_per_px(torch, P, ctx, bands, H, W, n_global, fm["model"])
        out = {"ms_per_step": round(t * 1e3, 2), "value": round(n_global / 1e6 / t, 2), "forest_ms": round(fms / max(fcnt, 1), 2),
               "host_syncs_per_step": round(syncs / reps, 1), "forest_fit_s_outside_timing": round(t_fit, 1),
               "note": "BASELINE configs[4] on one GPU: 19-feature stack (indices, PCA, 7x7 context, GLCM 21/21, gradient, local std, Sobel) + "
                       "RandomForest(100 trees, max_depth 16) inference, int64 labels; same raster"}
        if visits:
            per_s = n_global * visits / (fms / max(fcnt, 1) / 1e3)
            out["forest"] = {"bound": "lds_gather", "node_visits_per_px": round(visits, 1), "node_visits_per_s": round(per_s, 0),
                             "peak_node_visits_per_s": round(FOREST_VISITS_PEAK, 0), "frac": round(per_s / FOREST_VISITS_PEAK, 4)}
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def pcie_inclusive(torch, device, ctx, bands, step, n_global, step_qb=None):
    """The second figure SURVEY.md §8d asks for: the bands start in pinned HOST memory and the label map ends there.
    Never the bench `value`.  Four measurements, each for float32 bands (int32 / int64 labels back) and for the 8-bit
    digital numbers the TM tiles are (Context.upload_band: one byte per pixel over PCIe and in HBM, uint8 class ids back):
      serial            upload -> step -> download, nothing overlapped.  The device planes are touched and one untimed step
                        runs on them first, so `compute_ms` is the step itself (r03 had allocator growth / first touch inside)
      pipelined         one pass, one raster: the bands cross PCIe one after the other on a copy stream, THE NIR BAND FIRST, and each
                        band's order statistics (K1) are taken as soon as it has landed; the texture chain (re-normalise, quantise,
                        GLCM, five upsamples: 18 ms) depends on the NIR band alone and runs while the other six bands are still in
                        flight (pipeline.texture_planes); the index / PCA pass and KMeans need every band and start when the last
                        select is done.  Same arithmetic, same labels as the serial pass (compared).
      double_buffered   a stream of rasters, what a caller with many tiles does: raster i+1 is uploaded and the labels of
                        raster i-1 are downloaded (separate copy streams, PCIe is full duplex) while raster i is computed;
                        steady-state time per raster over three rasters."""
    try:
        out = {}
        for kind in ("float32_bands", "uint8_bands"):
            u8 = kind == "uint8_bands"
            src = [b.to(torch.uint8) for b in bands] if u8 else bands
            host = [torch.empty(b.numel(), dtype=b.dtype, pin_memory=True) for b in src]
            for h, b in zip(host, src):
                h.copy_(b)
            del src
            sets = [[torch.empty(h.numel(), dtype=h.dtype, device=device) for h in host] for _ in range(2)]
            for st in sets:                          # touch the fresh planes, then one untimed step on each set
                for d, h in zip(st, host):
                    d.copy_(h, non_blocking=True)
                lab0, _ = step(bands=st)
            fin = (lambda t: t.to(torch.uint8)) if u8 else (lambda t: t)
            lab0 = fin(lab0)
            lab_host = [torch.empty(lab0.numel(), dtype=lab0.dtype, pin_memory=True) for _ in range(2)]
            del lab0
            torch.cuda.synchronize()
            dev = sets[0]
            res = {}
            # ---- serial ----
            t0 = time.perf_counter()
            for h, d in zip(host, dev):
                d.copy_(h, non_blocking=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            labels, _ = step(bands=dev)
            labels = fin(labels)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            lab_host[0].copy_(labels, non_blocking=True)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            gb = sum(h.numel() * h.element_size() for h in host) / 1e9
            res["serial"] = {"value": round(n_global / 1e6 / (t3 - t0), 2), "pass_ms": round((t3 - t0) * 1e3, 1), "h2d_ms": round((t1 - t0) * 1e3, 1),
                             "h2d_GBs": round(gb / (t1 - t0), 1), "compute_ms": round((t2 - t1) * 1e3, 1), "d2h_labels_ms": round((t3 - t2) * 1e3, 1)}
            del labels
            main = torch.cuda.current_stream()
            up, down = torch.cuda.Stream(), torch.cuda.Stream()
            # ---- pipelined single pass (config 3 only) ----
            if step_qb is not None:
                from rsseg import pipeline as P
                for rep in range(2):       # the first pass warms the allocator for THIS order of requests (a cold pass stalls both
                    # streams in hipMalloc for 7-45 ms: r04_pipe_probe.txt); the second is the steady state of a caller with many rasters
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    order = [3] + [i for i in range(len(host)) if i != 3]      # the NIR band first: the texture chain needs nothing else
                    evs = {}
                    with torch.cuda.stream(up):
                        for i in order:
                            dev[i].copy_(host[i], non_blocking=True)
                            evs[i] = torch.cuda.Event()
                            evs[i].record(up)
                    qb = [None] * len(host)
                    tex = None
                    marks = []                                                 # host time at which each band's select is back
                    for i in order:
                        main.wait_event(evs[i])
                        qb[i] = P.band_quantile_bundle(ctx, dev[i], n_global)   # waits for this band only; the later bands keep flowing
                        if i == 3:        # quantise + GLCM + five upsamples of the NIR band run while the other six bands are in flight
                            tex = step_qb(dev, qb, None, texture_only=True)
                        marks.append(round((time.perf_counter() - t0) * 1e3, 1))
                    t1 = time.perf_counter()
                    labels, _ = step_qb(dev, qb, tex)
                    labels = fin(labels)
                    lab_host[1].copy_(labels, non_blocking=True)
                    torch.cuda.synchronize()
                    t3 = time.perf_counter()
                    same = bool(torch.equal(lab_host[0], lab_host[1]))         # the serial pass's labels
                    cold = res["pipelined"]["pass_ms"] if rep else None
                    res["pipelined"] = {"value": round(n_global / 1e6 / (t3 - t0), 2), "pass_ms": round((t3 - t0) * 1e3, 1),
                                        "upload_selects_and_texture_ms": round((t1 - t0) * 1e3, 1), "rest_of_step_and_download_ms": round((t3 - t1) * 1e3, 1),
                                        "band_order": order, "select_back_ms": marks, "first_cold_pass_ms": cold, "labels_equal_the_serial_pass": same}
                    del labels, tex
            # ---- double-buffered stream of rasters ----
            torch.cuda.synchronize()
            n_r = 4
            done_up = [None] * (n_r + 1)
            done_cmp = [None] * (n_r + 1)
            with torch.cuda.stream(up):
                for h, d in zip(host, sets[0]):
                    d.copy_(h, non_blocking=True)
                done_up[0] = torch.cuda.Event()
                done_up[0].record(up)
            marks = []
            keep = []
            for i in range(n_r):
                cur, nxt = sets[i & 1], sets[(i + 1) & 1]
                if i + 1 < n_r:
                    with torch.cuda.stream(up):                   # raster i+1 goes up while raster i is computed
                        if done_cmp[i - 1] is not None:
                            up.wait_event(done_cmp[i - 1])        # its buffers were read by raster i-1's step
                        for h, d in zip(host, nxt):
                            d.copy_(h, non_blocking=True)
                        done_up[i + 1] = torch.cuda.Event()
                        done_up[i + 1].record(up)
                main.wait_event(done_up[i])
                labels, _ = step(bands=cur)
                labels = fin(labels)
                done_cmp[i] = torch.cuda.Event()
                done_cmp[i].record(main)
                with torch.cuda.stream(down):                     # labels of raster i come down beside raster i+1's compute
                    down.wait_event(done_cmp[i])
                    lab_host[i & 1].copy_(labels, non_blocking=True)
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(down)
                    marks.append(e)
                keep.append(labels)                               # alive until its download has run
                if len(keep) > 2:
                    marks[len(keep) - 3].synchronize()
                    keep[len(keep) - 3] = None
            torch.cuda.synchronize()
            per = marks[0].elapsed_time(marks[-1]) / (n_r - 1)
            res["double_buffered"] = {"value": round(n_global / 1e6 / (per * 1e-3), 2), "ms_per_raster": round(per, 1), "rasters": n_r - 1,
                                      "bound": "upload" if res["serial"]["h2d_ms"] > res["serial"]["compute_ms"] else "compute"}
            out[kind] = res
            del host, sets, lab_host, keep
            torch.cuda.empty_cache()
        f = out["float32_bands"]
        flat = dict(f["serial"], unit="Mpixel/s", note="serial: 7 float32 bands pinned host -> HBM, one step, labels -> pinned host; planes pre-touched, one untimed step first")
        flat["pipelined"] = f.get("pipelined")
        flat["double_buffered"] = f["double_buffered"]
        flat["uint8_bands"] = dict(out["uint8_bands"]["serial"], pipelined=out["uint8_bands"].get("pipelined"),
                                   double_buffered=out["uint8_bands"]["double_buffered"],
                                   note="the same with the bands as uint8 (they stay uint8 in HBM: the kernels read 8-bit planes) and uint8 class ids back")
        return flat
    except Exception as e:  # noqa: BLE001
        import traceback
        return {"error": repr(e), "trace": traceback.format_exc()[-800:]}


def run_c2(ctx, P, bands, k, n_global):
    lohi = P.band_lohi(ctx, bands[:5], n_global)
    idx, _ = P.spectral_indices(ctx, bands, lohi)
    planes = [idx[n] for n in P.INDEX_NAMES]
    return ctx.kmeans_fit_predict(planes, k)


if __name__ == "__main__":
    main()
