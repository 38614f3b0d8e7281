"""GPU suite: the sharded path.  Two ranks (gloo rendezvous, both on cuda:0) each hold a row stripe of
the bundled scene; percentiles, PCA and KMeans run through the library's all-reduce hook.  Every result
must equal the single-rank result bit for bit — the reductions are exact fixed-point sums, so sharding
cannot change them."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _wait_all(procs, timeout=240):
    """Wait for every rank; as soon as one fails (or the deadline passes) the others are killed, so a crashed rank
    cannot leave its peers blocked in a collective."""
    import time
    t0 = time.time()
    try:
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):
                raise AssertionError(f"rank exit codes {codes}")
            if all(c == 0 for c in codes):
                return
            if time.time() - t0 > timeout:
                raise AssertionError(f"ranks still running after {timeout} s: {codes}")
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_single_rank(ctx, oracle, golden_dir, tmp_path, world):
    from rsseg import pipeline as P
    scene = np.load(os.path.join(golden_dir, "scene_aa.npz"))
    bands = np.stack(oracle.stage1_preprocess(scene["dn"]))[:, :301]  # 301 rows: uneven stripes, ragged tiles
    k = 6
    np.savez(tmp_path / "input.npz", bands=bands, k=k)
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), "gpu_shard", str(r), str(world), port, str(tmp_path)])
             for r in range(world)]
    _wait_all(procs)
    # single rank, same code path
    H, W = bands.shape[1:]
    dev = [ctx.to_device(bands[i].reshape(-1)) for i in range(7)]
    lohi = P.band_lohi(ctx, dev)
    idx, norms = P.spectral_indices(ctx, dev, lohi, want_norm=(True,) * 5)
    norm_all = list(norms) + [ctx.normalize(dev[i], float(lohi[i, 0]), float(lohi[i, 1])) for i in range(5, 7)]
    pcs, ratio, _ = P.pca(ctx, norm_all, 3, True)
    planes = [idx[n] for n in P.INDEX_NAMES] + list(pcs)
    labels, meta = ctx.kmeans_fit_predict(planes, k)
    want = labels.cpu().numpy()
    got = []
    for r in range(world):
        o = np.load(tmp_path / f"out_{r}.npz")
        assert np.array_equal(o["lohi"], lohi)
        assert np.array_equal(o["ratio"], ratio)
        assert int(o["n_iter"]) == meta["n_iter"]
        assert np.array_equal(o["init"], meta["init_indices"])
        assert np.array_equal(o["centers"], meta["centers"])
        assert np.array_equal(o["pc0"], pcs[0].cpu().numpy()[int(o["r0"]) * W:int(o["r1"]) * W])
        got.append(o["labels"])
    assert np.array_equal(np.concatenate(got), want)


@pytest.mark.parametrize("world", [2, 3])
def test_striped_config3_equals_single_gpu(ctx, oracle, tmp_path, world):
    """ONE raster sharded by rows, texture halos included: GLCM planes and labels of every stripe equal the rows
    of the single-GPU config-3 result bit for bit."""
    from rsseg import pipeline as P
    H, W = 150, 128
    bands = oracle.synthetic_raster(H, W)
    k = 8
    np.savez(tmp_path / "input.npz", bands=bands, k=k)
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), "gpu_striped_c3", str(r), str(world), port, str(tmp_path)])
             for r in range(world)]
    _wait_all(procs)
    labels, meta, planes = P.config3(ctx, [ctx.to_device(bands[i].reshape(-1)) for i in range(7)], H, W, k, 7, 1, 3)
    want = labels.cpu().numpy()
    g0, g4 = planes[7].cpu().numpy(), planes[11].cpu().numpy()
    got = []
    for r in range(world):
        o = np.load(tmp_path / f"out_{r}.npz")
        a, b = int(o["r0"]) * W, int(o["r1"]) * W
        assert np.array_equal(o["glcm0"], g0[a:b]) and np.array_equal(o["glcm4"], g4[a:b])
        assert int(o["n_iter"]) == meta["n_iter"]
        got.append(o["labels"])
    assert np.array_equal(np.concatenate(got), want)


def test_comm_arguments_are_checked(ctx):
    """rsseg_ctx_set_comm: more ranks than RSSEG_MAX_RANKS (16), a rank outside the world, and a failing hook are errors
    (the last one surfaces as RSSEG_ERR_COMM from the call that needed the collective), never a silent single-rank run."""
    from rsseg.runtime import Context, RssegError
    c = Context(0, use_dist=False)
    try:
        with pytest.raises(ValueError, match="rank/world"):
            c.install_comm_hook(0, 17, lambda *a: None)
        with pytest.raises(ValueError, match="rank/world"):
            c.install_comm_hook(3, 2, lambda *a: None)

        def broken(buf, offset, count, dtype, op):
            raise RuntimeError("link down")
        c.install_comm_hook(0, 2, broken)
        with pytest.raises(RssegError):
            c.order_stats(c.to_device(np.arange(1000, dtype=np.float32)), [10])
        # a hook that fails in the MIDDLE of KMeans (stream-ordered collectives between enqueued kernels): the call reports
        # it, and the context keeps working afterwards
        calls = {"n": 0}

        def flaky(buf, offset, count, dtype, op):
            calls["n"] += 1
            if calls["n"] == 9:
                raise RuntimeError("link down")
        c.install_comm_hook(0, 1, flaky)     # one rank: identity reductions, the hook is called all the same
        rng = np.random.default_rng(3)
        planes = [c.to_device(rng.random(50000).astype(np.float32)) for _ in range(4)]
        with pytest.raises(RssegError):
            c.kmeans_fit_predict(planes, 5)
        assert calls["n"] == 9
        labels, meta = c.kmeans_fit_predict(planes, 5)          # the hook works again: a complete fit
        ref, meta0 = ctx.kmeans_fit_predict([ctx.to_device(p.cpu().numpy()) for p in planes], 5)
        assert meta["n_iter"] == meta0["n_iter"] and np.array_equal(labels.cpu().numpy(), ref.cpu().numpy())
    finally:
        c.close()


def test_every_collective_through_a_one_rank_rccl_group(ctx, oracle, tmp_path):
    """The RCCL branch of make_allreduce_hook on the hardware a one-GPU box has: a one-rank `nccl` group, the hook
    installed all the same (Context(force_comm=True)), so that each collective of config 3 and of the 19-feature stack
    is an identity reduction that goes through torch.distributed / RCCL, stream-ordered, without a host wait in the
    hook.  Results equal those of a context without a hook bit for bit; the worker also checks the hook's ordering
    against work queued on a side stream."""
    from sklearn.ensemble import RandomForestClassifier
    from rsseg import pipeline as P
    from rsseg.forest import flatten_forest
    H, W = 230, 200
    bands = oracle.synthetic_raster(H, W)
    dev = [ctx.to_device(bands[i].reshape(-1)) for i in range(7)]
    planes, _ = P.feature_stack19(ctx, dev, H, W)
    X = np.stack([p.cpu().numpy() for p in P.stack19_forest_planes(ctx, planes)], 1)
    rng = np.random.default_rng(6)
    sel = rng.choice(H * W, 3000, replace=False)
    y = ((np.arange(H)[:, None] // 16 + np.arange(W)[None, :] // 16) % 4).reshape(-1)[sel]
    forest = flatten_forest(RandomForestClassifier(n_estimators=8, max_depth=8, random_state=2).fit(X[sel], y))
    np.savez(tmp_path / "input.npz", bands=bands, k=6, **{f"forest_{k}": np.asarray(v) for k, v in forest.items()})
    p = subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), "rccl_single", "0", "1", _free_port(), str(tmp_path)])
    _wait_all([p])
    assert int(open(tmp_path / "ok_rccl").read()) >= 20


@pytest.mark.parametrize("H,W,world", [(190, 136, 2), (190, 136, 3), (173, 150, 3)])
def test_striped_stack19_and_forest_equal_single_gpu(ctx, oracle, tmp_path, H, W, world):
    """BASELINE config 5 sharded: every rank holds its stripe plus the halo rows of stack19_halo_rows (3 rows for the 7x7
    context mean, 2 for the 5x5 operators, 1 for Sobel, the GLCM windows its bilinear taps reach).  The 19 planes and
    the forest labels of every stripe equal the rows of the single-GPU result bit for bit (W = 150: rows that are not
    16-byte aligned)."""
    from sklearn.ensemble import RandomForestClassifier
    from rsseg import pipeline as P
    from rsseg.forest import flatten_forest
    bands = oracle.synthetic_raster(H, W)
    dev = [ctx.to_device(bands[i].reshape(-1)) for i in range(7)]
    planes, _ = P.feature_stack19(ctx, dev, H, W)
    fp = P.stack19_forest_planes(ctx, planes)
    X = np.stack([p.cpu().numpy() for p in fp], 1)
    rng = np.random.default_rng(5)
    sel = rng.choice(H * W, 3000, replace=False)
    y = ((np.arange(H)[:, None] // 16 + np.arange(W)[None, :] // 16) % 5).reshape(-1)[sel]
    model = RandomForestClassifier(n_estimators=12, max_depth=9, random_state=1).fit(X[sel], y)
    forest = flatten_forest(model)
    ctx.forest_load(forest)
    want = ctx.forest_predict(fp).cpu().numpy()
    assert np.array_equal(want, model.predict(X))
    np.savez(tmp_path / "input.npz", bands=bands, **{f"forest_{k}": np.asarray(v) for k, v in forest.items()})
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), "gpu_striped_c5", str(r), str(world), port, str(tmp_path)])
             for r in range(world)]
    _wait_all(procs)
    host_planes = [p.cpu().numpy() for p in planes]
    got = []
    for r in range(world):
        o = np.load(tmp_path / f"out_{r}.npz")
        a, b = int(o["r0"]) * W, int(o["r1"]) * W
        for i, hp in enumerate(host_planes):
            assert np.array_equal(o[f"p{i}"], hp[a:b], equal_nan=True), (r, i)
        got.append(o["labels"])
    assert np.array_equal(np.concatenate(got), want)


def test_window_ops_rows_form_equals_full_plane(ctx):
    """Rows form of the window operators: any row range of a plane, computed from the range plus its halo rows only,
    equals the same rows of the full-plane result; missing halo rows are refused."""
    from rsseg import _lib as L
    rng = np.random.default_rng(17)
    H, W = 97, 300
    x = rng.random((H, W)).astype(np.float32)
    q = rng.integers(0, 256, (H, W)).astype(np.uint8)
    dx, dq = ctx.to_device(x.reshape(-1)), ctx.to_device(q.reshape(-1))
    full = dict(box7=ctx.box_mean(dx, H, W, 7, L.BORDER_REFLECT), box5=ctx.box_mean(dx, H, W, 5, L.BORDER_REFLECT101, square=True),
                std5=ctx.local_std(dx, H, W, 5), var7=ctx.local_var(dx, H, W, 7), grad5=ctx.morph_gradient(dq, H, W, 5),
                open7=ctx.morph(dq, H, W, 7, L.MORPH_OPEN), close3=ctx.morph(dq, H, W, 3, L.MORPH_CLOSE), ero5=ctx.morph(dq, H, W, 5, L.MORPH_ERODE))
    halo = dict(box7=3, box5=2, std5=2, var7=3, grad5=2, open7=6, close3=2, ero5=2)
    for (a, b) in [(0, 40), (33, 64), (64, 97), (10, 11), (0, 97)]:
        for name, R in halo.items():
            s0, s1 = max(a - R, 0), min(b + R, H)
            edges = (1 if s0 == 0 else 0) | (2 if s1 == H else 0)
            src = (dq if name[0] in "goce" else dx)[s0 * W:s1 * W].clone()
            rows = (a - s0, b - s0)
            Hs = s1 - s0
            got = {"box7": lambda: ctx.box_mean(src, Hs, W, 7, L.BORDER_REFLECT, rows=rows, edges=edges),
                   "box5": lambda: ctx.box_mean(src, Hs, W, 5, L.BORDER_REFLECT101, square=True, rows=rows, edges=edges),
                   "std5": lambda: ctx.local_std(src, Hs, W, 5, rows=rows, edges=edges),
                   "var7": lambda: ctx.local_var(src, Hs, W, 7, rows=rows, edges=edges),
                   "grad5": lambda: ctx.morph_gradient(src, Hs, W, 5, rows=rows, edges=edges),
                   "open7": lambda: ctx.morph(src, Hs, W, 7, L.MORPH_OPEN, rows=rows, edges=edges),
                   "close3": lambda: ctx.morph(src, Hs, W, 3, L.MORPH_CLOSE, rows=rows, edges=edges),
                   "ero5": lambda: ctx.morph(src, Hs, W, 5, L.MORPH_ERODE, rows=rows, edges=edges)}[name]()
            assert np.array_equal(got.cpu().numpy(), full[name].cpu().numpy()[a * W:b * W]), (name, a, b)
    with pytest.raises(ValueError):   # an interior stripe without its halo rows
        ctx.box_mean(dx[10 * W:20 * W].clone(), 10, W, 7, L.BORDER_REFLECT, rows=(0, 10), edges=0)
    # the multi-plane launch equals the single-plane launches
    ps = [ctx.to_device(rng.random(H * W).astype(np.float32)) for _ in range(7)]
    multi = ctx.box_mean_multi(ps, H, W, 7, L.BORDER_REFLECT)
    for p, m in zip(ps, multi):
        assert np.array_equal(m.cpu().numpy(), ctx.box_mean(p, H, W, 7, L.BORDER_REFLECT).cpu().numpy())


class _ThreadWorld:
    """N ranks as N threads of this process on one GPU: the library's all-reduce hook of every rank meets at a barrier,
    rank 0 reduces the N device buffers, every rank copies the result back.  Lets the suite run the 8-way split the
    8-GPU bench uses without 8 processes on the card."""

    def __init__(self, world):
        import threading
        self.world = world
        self.bar = threading.Barrier(world, timeout=120)
        self.slots = [None] * world
        self.result = None
        self.calls = 0

    def hook(self, rank):
        import torch
        from rsseg import _lib as L
        views = {L.F32: torch.float32, L.F64: torch.float64, L.I64: torch.int64}

        def fn(buf, offset, count, dtype, op):
            esz = 4 if dtype == L.F32 else 8
            t = buf[offset:offset + count * esz].view(views[dtype])
            torch.cuda.synchronize()
            self.slots[rank] = t
            self.bar.wait()
            if rank == 0:
                st = torch.stack(self.slots)
                self.result = st.sum(0) if op == L.SUM else (st.amin(0) if op == L.MIN else st.amax(0))
                self.calls += 1
                torch.cuda.synchronize()
            self.bar.wait()
            t.copy_(self.result)
            torch.cuda.synchronize()
            self.bar.wait()

        return fn

    def run(self, target):
        """target(rank) in one thread per rank; re-raises the first failure (the barrier is aborted so nobody hangs)."""
        import threading
        errs = []

        def wrap(r):
            try:
                target(r)
            except BaseException as e:  # noqa: BLE001
                errs.append((r, e))
                self.bar.abort()

        th = [threading.Thread(target=wrap, args=(r,)) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join(300)
        assert not any(t.is_alive() for t in th), "a rank thread did not finish"
        if errs:
            raise AssertionError(f"rank {errs[0][0]} failed: {errs[0][1]!r}")


@pytest.mark.parametrize("H,W,world", [(203, 136, 8), (203, 136, 5), (9, 40, 8), (331, 136, 16)])
def test_many_stripes_in_threads_equal_single_gpu(ctx, oracle, H, W, world):
    """The 8-way row split of the multi-GPU bench, an uneven 5-way one, and one-row stripes (9 rows over 8 ranks: halos
    wider than the stripes), every rank a thread with its own context:
    config 3 (texture halos) and the 19-feature stack of config 5 (3 / 2 / 1-row and window-aligned halos), labels and
    planes of every stripe against the rows of the single-context result, bit for bit."""
    import torch
    from rsseg import pipeline as P
    from rsseg.runtime import Context
    k = 8 if H > 50 else 4
    with19 = H >= 42          # the 19-feature stack needs two 21-row texture windows; the 9-row raster (one-row stripes) runs config 3 only
    bands = oracle.synthetic_raster(H, W)
    dev = [ctx.to_device(bands[i].reshape(-1)) for i in range(7)]
    labels, meta, planes = P.config3(ctx, dev, H, W, k, 7, 1, 3)
    want_labels = labels.cpu().numpy()
    want_planes = [p.cpu().numpy() for p in planes]
    want19 = []
    if with19:
        s19, _ = P.feature_stack19(ctx, dev, H, W)
        want19 = [p.cpu().numpy() for p in s19]
    tw = _ThreadWorld(world)
    out = [None] * world

    def rank_main(r):
        c = Context(0, use_dist=False)
        c.install_comm_hook(r, world, tw.hook(r))
        r0, r1 = P.stripe_rows(H, world, r)
        j0, j1, i0, i1 = P.glcm_halo_rows(H, r0, r1, 7, 1)
        d = [c.to_device(bands[i, r0:r1].reshape(-1)) for i in range(7)]
        nir_ext = c.to_device(bands[3, i0:i1].reshape(-1))
        lab, m, pl = P.config3_striped(c, d, nir_ext, H, W, r0, r1, i0, k)
        p19 = []
        if with19:
            e0, e1 = P.stack19_halo_rows(H, r0, r1)
            ext = [c.to_device(bands[i, e0:e1].reshape(-1)) for i in range(7)]
            p19, _ = P.stack19_striped(c, ext, H, W, r0, r1, e0)
        torch.cuda.synchronize()
        out[r] = (r0, r1, lab.cpu().numpy(), m["n_iter"], [p.cpu().numpy() for p in pl], [p.cpu().numpy() for p in p19])
        c.close()

    tw.run(rank_main)
    assert tw.calls > 10
    for r in range(world):
        r0, r1, lab, n_iter, pl, p19 = out[r]
        a, b = r0 * W, r1 * W
        assert n_iter == meta["n_iter"], r
        assert np.array_equal(lab, want_labels[a:b]), r
        for i, p in enumerate(pl):
            assert np.array_equal(p, want_planes[i][a:b], equal_nan=True), (r, i)
        for i, p in enumerate(p19):
            assert np.array_equal(p, want19[i][a:b], equal_nan=True), (r, i)


@pytest.mark.parametrize("F,k,dt,splits", [(40, 5, np.float64, (0, 7001, 7001, 20011)),      # feature-blocked kernels; rank 1 holds NO pixels
                                            (15, 8, np.float32, (0, 16384 * 3 + 5, 16384 * 3 + 9, 70001)),   # a 4-pixel rank
                                            (2, 11, np.float32, (0, 30, 64, 97))])           # duplicate-heavy (seed 22 of the relocation test): clusters run empty
def test_kmeans_entry_point_sharded_over_thread_ranks(ctx, oracle, F, k, dt, splits):
    """rsseg_kmeans_fit_predict itself on consecutive slices of the pixel list, one context per slice (threads of this
    process): the device-resident state, the stream-ordered collectives between the control kernels, a rank without
    pixels, a rank smaller than one chunk, the feature-blocked kernels (F = 40) and the empty-cluster hand-over to the
    host path, against the single-context run and the oracle: seeds, iteration count, relocations and labels."""
    from rsseg.runtime import Context
    rng = np.random.default_rng(F * 100 + k)
    n = splits[-1]
    if F == 2:   # the data of test_kmeans_empty_cluster_relocation[22]: 97 points, 2 features, 11 clusters, 2 relocations
        rng = np.random.default_rng(22)
        n_, F_, k_ = int(rng.integers(20, 120)), int(rng.integers(1, 4)), int(rng.integers(6, 14))
        assert (n_, F_, k_) == (n, F, k)
        X = rng.random((n, F)).astype(np.float32)
        X = (np.round(X * rng.integers(2, 6)) / 4.0).astype(dt)
    else:
        cent = rng.random((k + 2, F))
        X = (cent[rng.integers(0, k + 2, n)] + rng.normal(0, 0.07, (n, F))).astype(dt)
    planes = [np.ascontiguousarray(X[:, f]) for f in range(F)]
    want, info = oracle.kmeans_fit_planes(planes, k)
    if F == 2:
        assert info["relocated"] > 0
    labels, meta = ctx.kmeans_fit_predict([ctx.to_device(p) for p in planes], k)
    assert np.array_equal(labels.cpu().numpy(), want) and meta["n_iter"] == info["n_iter"]
    world = len(splits) - 1
    tw = _ThreadWorld(world)
    out = [None] * world

    def rank_main(r):
        c = Context(0, use_dist=False)
        c.install_comm_hook(r, world, tw.hook(r))
        a, b = splits[r], splits[r + 1]
        d = [c.to_device(p[a:b].copy()) for p in planes]
        lab, m = c.kmeans_fit_predict(d, k)
        out[r] = (lab.cpu().numpy(), m["n_iter"], m["relocated"], m["init_indices"].copy())
        c.close()

    tw.run(rank_main)
    for r in range(world):
        lab, n_iter, reloc, seeds = out[r]
        assert n_iter == info["n_iter"] and reloc == info["relocated"], r
        assert np.array_equal(seeds, info["init_indices"]), r
        assert np.array_equal(lab, want[splits[r]:splits[r + 1]]), r


def test_full_size_eight_stripes_in_threads_equal_single_gpu(ctx):
    """BASELINE configs[3] at its real size: the 16384 x 16384 x 7 raster split into 8 stripes of 2048 rows (what
    `bench.py --gpus 8` gives every GPU), each stripe a thread with its own context on this one GPU.  Labels, iteration
    count and the texture / component planes of every stripe equal the rows of the single-context run bit for bit —
    stripe offsets beyond 2^31 bytes, 2048 k-means chunks per rank, 8-rank prefix logic of the k-means++ sampling."""
    import torch
    from rsseg import pipeline as P
    from rsseg.runtime import Context
    H = W = 16384
    world, k = 8, 8
    yy = torch.arange(H, device=ctx.device, dtype=torch.int32)[:, None]
    xx = torch.arange(W, device=ctx.device, dtype=torch.int32)[None, :]
    bands = []
    for b in range(7):
        v = (yy * (3 + b) + xx * (5 + 2 * b) + (yy >> 5) * (xx >> 6) * 7 + ((yy * xx) >> 9) + b * 31) % 256
        bands.append(v.to(torch.float32).reshape(-1).contiguous())
        del v
    labels, meta, planes = P.config3(ctx, bands, H, W, k, 7, 1, 3, H * W)
    torch.cuda.synchronize()
    keep = [planes[7], planes[11], planes[14]]
    del planes
    tw = _ThreadWorld(world)
    ok = [None] * world

    def rank_main(r):
        c = Context(0, use_dist=False)
        c.install_comm_hook(r, world, tw.hook(r))
        r0, r1 = P.stripe_rows(H, world, r)
        j0, j1, i0, i1 = P.glcm_halo_rows(H, r0, r1, 7, 1)
        d = [b[r0 * W:r1 * W] for b in bands]
        nir_ext = bands[3][i0 * W:i1 * W]
        lab, m, pl = P.config3_striped(c, d, nir_ext, H, W, r0, r1, i0, k)
        torch.cuda.synchronize()
        a, e = r0 * W, r1 * W
        ok[r] = (m["n_iter"] == meta["n_iter"], bool(torch.equal(lab, labels[a:e])),
                 [bool(torch.equal(pl[i], kp[a:e])) for i, kp in zip((7, 11, 14), keep)])
        del lab, pl
        c.close()

    tw.run(rank_main)
    for r in range(world):
        assert ok[r] is not None and ok[r][0] and ok[r][1] and all(ok[r][2]), (r, ok[r])
