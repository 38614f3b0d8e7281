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
