"""CPU suite, part 2: host logic of the product (no GPU compute): the C-ABI library loads and exports
every symbol of include/rsseg.h, the percentile / median scalar arithmetic equals NumPy's, the
MT19937 + cumulative-probability restatement equals numpy.random.RandomState, the TIFF reader reads
what it wrote."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from rsseg import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "rsseg.h")).read()
    declared = set(re.findall(r"\b(rsseg_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"rsseg_allreduce_fn"}
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in rsseg.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.rsseg_version().startswith(b"rsseg-hip")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rs-image-segmentation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in txt.replace("oracle/", "ORACLE_DOC/").replace("oracle.c", "ORACLE_DOC").replace("oracle)", "ORACLE_DOC"), \
                    f"{f} references the oracle"


@pytest.mark.parametrize("n", [1, 2, 3, 10, 101, 9216, 360000])
@pytest.mark.parametrize("q", [2, 98, 25, 50, 0, 100])
def test_percentile_plan_equals_numpy(n, q):
    from rsseg.quantiles import percentile_plan
    rng = np.random.default_rng(n * 1000 + q)
    a = rng.integers(0, 255, n).astype(np.float32) + (rng.random(n) < 0.3) * rng.random(n).astype(np.float32)
    a = a.astype(np.float32)
    s = np.sort(a)
    ranks, fin = percentile_plan(n, q, np.float32, True)
    got = fin(s[ranks])
    want = np.percentile(a, q)
    assert got.dtype == want.dtype == np.float32
    assert got == want


@pytest.mark.parametrize("n", [2, 7, 100, 9216, 360001])
def test_robust_scaler_plans_equal_numpy(n):
    from rsseg.quantiles import median_plan, percentile_plan
    rng = np.random.default_rng(n)
    a = rng.random(n).astype(np.float32)
    s = np.sort(a)
    r, fin = median_plan(n, np.float32)
    assert fin(s[r]) == np.nanmedian(a)
    r, fin = percentile_plan(n, (25.0, 75.0), np.float32, False)
    got = fin(s[r])
    want = np.nanpercentile(a, (25.0, 75.0))
    assert got.dtype == want.dtype and np.array_equal(got, want)


@pytest.mark.parametrize("n", [1, 2, 3, 10, 101, 9216, 360000, 33554432])
def test_lerp_rows_equals_plan_finish(n):
    """The few-call finish of several planes (pipeline.band_quantile_bundles) gives, row by row, what np.percentile /
    np.nanpercentile give on each plane: every quantile, scalar q (float32 virtual index) and tuple q (float64)."""
    from rsseg.quantiles import lerp_rows, percentile_plan
    rng = np.random.default_rng(n)
    m = min(n, 4000)
    planes = [np.sort((rng.integers(0, 255, m) + (rng.random(m) < 0.5) * rng.random(m)).astype(np.float32)) for _ in range(5)]
    for q, scalar in [(2, True), (98, True), (50, True), (0, True), (100, True), (37.5, True), ((25.0, 75.0), False), ((10.0, 50.0, 99.9), False)]:
        ranks, fin = percentile_plan(n, q, np.float32, scalar)
        # the order statistics a select would fetch: any sorted values will do for ranks beyond the sample
        vals = np.stack([pl[np.minimum(np.asarray(ranks) * m // max(n, 1), m - 1)] for pl in planes])
        vals = np.sort(vals.reshape(len(planes), 2, -1), axis=1).reshape(len(planes), -1)   # prev <= next, column by column
        got = lerp_rows(fin, vals)
        for i in range(len(planes)):
            want = fin(vals[i])
            assert np.asarray(got[i]).dtype == np.asarray(want).dtype
            assert np.array_equal(got[i], want), (n, q, i)
        if n == m:   # and against NumPy itself on the full sample
            exact = np.stack([pl[ranks] for pl in planes])
            for i, pl in enumerate(planes):
                assert np.array_equal(lerp_rows(fin, exact)[i], np.percentile(pl, q))


@pytest.mark.parametrize("n,k,dt", [(5, 2, np.float32), (9216, 6, np.float32), (360000, 8, np.float32), (1000003, 7, np.float64),
                                    (12345677, 5, np.float32), (4194304, 8, np.float64), (2999999, 3, np.float32)])
def test_kmeans_draws_equal_randomstate(n, k, dt):
    from rsseg.runtime import host_kmeans_draws
    cid, u = host_kmeans_draws(n, k, dt, 42)
    rs = np.random.RandomState(42)
    w = np.ones(n, dtype=dt)
    assert cid == int(rs.choice(n, p=w / w.sum()))
    L = 2 + int(np.log(k))
    ur = np.concatenate([rs.uniform(size=L) for _ in range(k - 1)])
    assert np.array_equal(u, ur)


def test_kmeans_draws_other_seeds():
    from rsseg.runtime import host_kmeans_draws
    for seed in (0, 1, 7, 12345, 2 ** 31):
        for n in (1000, 77777):
            cid, _ = host_kmeans_draws(n, 4, np.float32, seed)
            rs = np.random.RandomState(seed)
            w = np.ones(n, dtype=np.float32)
            assert cid == int(rs.choice(n, p=w / w.sum())), (seed, n)


def test_geotiff_tags_roundtrip(tmp_path):
    """transform / EPSG / nodata survive a write-read cycle; tag layout is valid TIFF (sorted tags, even offsets)."""
    import struct
    from rsseg.tiff import read_tiff, read_tiff_georef, write_tiff
    a = (np.arange(3 * 5 * 7) % 251).astype(np.uint8).reshape(3, 5, 7)
    p = str(tmp_path / "g.tif")
    tr = (30.0, 0.0, 500000.0, 0.0, -30.0, 4100000.0)
    write_tiff(p, a, transform=tr, epsg=32650, nodata=0)
    assert np.array_equal(read_tiff(p), a)
    g = read_tiff_georef(p)
    assert g["transform"] == tr and g["epsg"] == 32650 and g["nodata"] == 0.0
    buf = open(p, "rb").read()
    (off,) = struct.unpack_from("<I", buf, 4)
    (n,) = struct.unpack_from("<H", buf, off)
    tags = [struct.unpack_from("<H", buf, off + 2 + 12 * i)[0] for i in range(n)]
    assert tags == sorted(tags) and {33550, 33922, 34735, 42113} <= set(tags)
    rot = (10.0, 2.0, 100.0, -1.5, -10.0, 900.0)   # rotated grid -> ModelTransformation
    write_tiff(p, a.astype(np.float64), transform=rot, epsg=4326, nodata=-9999.5)
    g = read_tiff_georef(p)
    assert g["transform"] == rot and g["epsg"] == 4326 and g["nodata"] == -9999.5
    write_tiff(p, a)
    assert read_tiff_georef(p) == {"transform": None, "epsg": None, "nodata": None}


def test_stage_geotiff_outputs(tmp_path):
    from rsseg import stages
    from rsseg.tiff import read_tiff, read_tiff_georef
    rng = np.random.default_rng(0)
    hier = {"level_1": rng.random((6, 9, 14)), "level_2": rng.random((6, 9, 5)), "all": rng.random((6, 9, 19))}
    tr = (30.0, 0.0, 1000.0, 0.0, -30.0, 2000.0)
    paths = stages.save_feature_outputs(str(tmp_path), {"ndvi": np.zeros((6, 9), np.float32)}, hier, 6, 9, transform=tr, crs="EPSG:32650")
    back = read_tiff(paths["tif"])
    assert back.shape == (19, 6, 9) and back.dtype == np.float64 and np.array_equal(np.moveaxis(back, 0, -1), hier["all"])
    assert read_tiff_georef(paths["tif"])["epsg"] == 32650
    out = stages.save_class_map_tif(np.arange(54).reshape(6, 9) % 4, str(tmp_path / "cls.tif"), transform=tr, crs=32650)
    assert read_tiff(out).dtype == np.uint8 and read_tiff_georef(out)["nodata"] == 0.0


def test_tiff_roundtrip_and_bundled_layout(tmp_path, golden_dir):
    from rsseg.tiff import read_tiff, write_tiff
    dn = np.load(os.path.join(golden_dir, "scene_aa.npz"))["dn"]
    p = str(tmp_path / "a.tif")
    write_tiff(p, dn)
    back = read_tiff(p)
    assert back.dtype == np.uint8 and np.array_equal(back, dn)
    f = np.random.default_rng(0).random((3, 17, 23)).astype(np.float32)
    write_tiff(p, f)
    assert np.array_equal(read_tiff(p), f)
    d = np.random.default_rng(1).random((19, 8, 9))
    write_tiff(p, d)
    assert np.array_equal(read_tiff(p), d)
    with pytest.raises(ValueError):
        open(p, "wb").write(b"not a tiff")
        read_tiff(p)


def test_tiff_lzw_tiles_bigtiff_against_libtiff(tmp_path):
    """The writer's LZW / tiled / BigTIFF forms (the reference asks rasterio for compress='lzw', tiled 256 x 256:
    scripts/2_feature_extraction.py:239-258) against an independent implementation: Pillow's libtiff reads what we
    write and we read what libtiff writes, bit for bit; the LZW codec round-trips incompressible and highly
    compressible data through table resets."""
    from PIL import Image
    from rsseg.tiff import lzw_decode, lzw_encode, read_tiff, read_tiff_georef, write_tiff
    rng = np.random.default_rng(5)
    for data in (b"", b"a", bytes(rng.integers(0, 256, 70001, dtype=np.uint8)), bytes(200000), bytes(rng.integers(0, 3, 90000, dtype=np.uint8))):
        enc = lzw_encode(data)
        assert lzw_decode(enc, len(data)) == data
    cls = (rng.integers(0, 5, (300, 517)) * (rng.random((300, 517)) < 0.7)).astype(np.uint8)
    cls[50:200, 100:400] = 3
    tr = (30.0, 0.0, 500000.0, 0.0, -30.0, 4100000.0)
    for kw in (dict(compress="lzw"), dict(compress="lzw", tiled=False), dict(tiled=True), dict(bigtiff=True), dict(compress="lzw", bigtiff=True)):
        p = str(tmp_path / "c.tif")
        write_tiff(p, cls, transform=tr, epsg=32650, nodata=0, **kw)
        assert np.array_equal(read_tiff(p)[0], cls), kw
        assert read_tiff_georef(p) == {"transform": tr, "epsg": 32650, "nodata": 0.0}
        with Image.open(p) as im:                     # libtiff's decoder on our file
            assert np.array_equal(np.asarray(im), cls), kw
    f32 = rng.random((130, 300)).astype(np.float32)
    p = str(tmp_path / "f.tif")
    write_tiff(p, f32, compress="lzw")
    with Image.open(p) as im:
        assert np.array_equal(np.asarray(im), f32)
    # libtiff's encoder -> our decoder (strips and tiles)
    q = str(tmp_path / "pil.tif")
    Image.fromarray(cls).save(q, compression="tiff_lzw")
    assert np.array_equal(read_tiff(q)[0], cls)
    Image.fromarray(f32).save(q, compression="tiff_lzw", tile=(256, 256)) if False else Image.fromarray(f32).save(q, compression="tiff_lzw")
    assert np.array_equal(read_tiff(q)[0], f32)
    # multi-band float64 stack, tiled LZW in a BigTIFF container: the layout of all_hierarchical_features.tif at full size
    stack = rng.random((19, 40, 300))
    stack[3] = 0.0
    write_tiff(p, stack, compress="lzw", bigtiff=True, transform=tr, epsg=4087, geographic=False)
    assert np.array_equal(read_tiff(p), stack)
    assert read_tiff_georef(p)["epsg"] == 4087
    # one band: no ExtraSamples tag; several bands: B - 1 entries
    import struct
    write_tiff(p, cls)
    buf = open(p, "rb").read()
    (off,) = struct.unpack_from("<I", buf, 4)
    (n,) = struct.unpack_from("<H", buf, off)
    assert 338 not in [struct.unpack_from("<H", buf, off + 2 + 12 * i)[0] for i in range(n)]


def test_context_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rsseg.runtime import Context, RssegError
    with pytest.raises(RssegError):
        Context(0)


def test_feature_dict_plumbing_matches_reference_key_rule(golden_dir, tmp_path):
    """load_features / normalize_features_structure: same top-level keys, order and shapes as
    the reference function produced for the same nested dict (tests/golden/nfs_golden.json)."""
    import json
    import pickle
    from modules.features.extract import load_features, normalize_features_structure
    z = np.zeros((4, 5), np.float32)
    nested = {"hierarchical_features": {"level_1": np.zeros((4, 5, 3)), "all": np.zeros((4, 5, 6))},
              "all_extracted_features_dict": {"NDVI": z, "pca_result": [z, z + 1], "glcm_features": {"contrast": z},
                                              "variance_ratio": np.zeros(3), "scalar": 1.5},
              "dimensions": (4, 5), "geo_transform": None, "crs": None}
    want = json.load(open(os.path.join(golden_dir, "nfs_golden.json")))
    p = str(tmp_path / "f.pkl")
    pickle.dump(nested, open(p, "wb"))
    got = normalize_features_structure(load_features(p))
    assert list(got.keys()) == want["keys"]
    assert got["height"] == want["height"] and got["width"] == want["width"]
    assert {k: list(v.shape) for k, v in got.items() if isinstance(v, np.ndarray)} == want["shapes"]
    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    np.save(str(tmp_path / "a.npy"), a)
    d = normalize_features_structure(load_features(str(tmp_path / "a.npy")))
    assert d["height"] == 3 and d["width"] == 4 and "all_features_feature_2" in d
    with pytest.raises(FileNotFoundError):
        load_features(str(tmp_path / "missing.pkl"))
    with pytest.raises(ValueError):
        open(str(tmp_path / "x.txt"), "w").write("x")
        load_features(str(tmp_path / "x.txt"))


def test_stripe_rows_cover_the_raster_and_refuse_empty_stripes():
    from rsseg import pipeline as P
    for H, world in ((16384, 8), (203, 5), (9, 8), (8, 8)):
        rows = [P.stripe_rows(H, world, r) for r in range(world)]
        assert rows[0][0] == 0 and rows[-1][1] == H
        assert all(a[1] == b[0] for a, b in zip(rows, rows[1:])) and all(r1 > r0 for r0, r1 in rows)
    with pytest.raises(ValueError):
        P.stripe_rows(7, 8, 0)
    with pytest.raises(ValueError):
        P.stripe_rows(100, 4, 4)


def test_run_classification_stage_has_the_reference_signature():
    """scripts/3_classification.py:267: (feature_file_path, method='rule_based', output_dir="segmentation_outputs",
    use_hierarchical_all=True); everything else this implementation adds is keyword-only."""
    import inspect
    from rsseg import stages
    sig = inspect.signature(stages.run_classification_stage)
    pos = [p for p in sig.parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD]
    assert [(p.name, p.default) for p in pos] == [("feature_file_path", inspect.Parameter.empty), ("method", "rule_based"),
                                                  ("output_dir", "segmentation_outputs"), ("use_hierarchical_all", True)]
    assert {n for n, p in sig.parameters.items() if p.kind == p.KEYWORD_ONLY} == {"n_clusters", "classifier", "feature_keys", "ctx",
                                                                                       "labeled_roi_file", "strict_reference"}
    assert sig.parameters["labeled_roi_file"].default == "labeled_roi.tif"      # scripts/3:403
    assert sig.parameters["n_clusters"].default == 7          # scripts/3:390
    sig2 = inspect.signature(stages.run_feature_extraction_stage)
    assert list(sig2.parameters)[:3] == ["bands_data", "preprocessing", "texture_band_index"]


def test_classification_stage_error_paths_need_no_gpu(tmp_path, capsys):
    """The reference prints and returns on unusable input (scripts/3:283-311, 489-490); FileNotFoundError is caught there too."""
    from rsseg import stages
    assert stages.run_classification_stage(str(tmp_path / "missing.pkl"), "kmeans", str(tmp_path / "o")) is None
    import pickle
    with open(tmp_path / "empty.pkl", "wb") as f:
        pickle.dump({"dimensions": (4, 4)}, f)
    assert stages.run_classification_stage(str(tmp_path / "empty.pkl"), "kmeans", str(tmp_path / "o")) is None
    with open(tmp_path / "ok.pkl", "wb") as f:
        pickle.dump({"all_extracted_features_dict": {"ndvi": np.zeros((4, 4), np.float32)}, "dimensions": (4, 4)}, f)
    assert stages.run_classification_stage(str(tmp_path / "ok.pkl"), "no_such_method", str(tmp_path / "o")) is None
    assert "不支持的分割方法" in capsys.readouterr().out


def _bench(*argv, env=None):
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([_sys.executable, os.path.join(root, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=600)


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 3` run bare — the form of the driver's command — starts 3 fresh ranks with the rendezvous
    environment, waits for them and relays rank 0's single JSON line (here: the CPU self-test of the launch, a gloo
    all-reduce of rank + 1 over 3 ranks = 6)."""
    import json
    r = _bench("--gpus", "3", "--launch-selftest")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["sum_of_rank_ids_plus_1"] == 6.0


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = _bench("--gpus", "2", "--launch-selftest", env={"WORLD_SIZE": "4", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
    r = _bench("--gpus", "1", "--launch-selftest", env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0
    r = _bench("--gpus", "1", "--launch-selftest")           # one rank, no rendezvous
    assert r.returncode == 0 and '"n_gpus": 1' in r.stdout


def test_bench_reports_a_failed_rank():
    """A rank that dies (here: no GPU in the CPU container, every rank of the real bench exits) makes the launcher exit
    non-zero instead of printing a line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a host without a GPU")
    r = _bench("--gpus", "2", "--steps", "1", "--no-cpu-baseline")
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_committed_bench_lines_carry_the_contract_fields():
    """The JSON lines bench.py printed on the MI355X (profiles/r04_bench_*.json): every field of the driver's contract and the
    two extra objects (roofline with live HIP-event timing of the dominant kernel and PMC traffic, cpu_baseline on a bounded
    sample), and the arithmetic that ties them together."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r04_bench_c[235]*.json")))
    assert len(files) >= 4
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
            assert k in d, (f, k)
        assert d["metric"].startswith("Mpixel/s") and d["unit"] == "Mpixel/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
        assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
        h, w = d["config"]["raster"][:2]
        assert abs(d["value"] - h * w / 1e6 / (d["ms_per_step"] / 1e3)) / d["value"] < 2e-3, f     # value = pixels / step time
        r = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernels"):
            assert k in r, (f, k)
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3 and 0 < r["frac"] < 1
        assert r["kernels_ms_per_step"] <= d["ms_per_step"] * 1.01                              # the kernels fit inside the step
        if os.path.basename(f) == "r04_bench_c3.json":      # the default line at the benchmark size (PMC passes exist for 16384^2 only)
            assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
            assert r["traffic"] is not None and 0.9 < r["traffic"] / r["algorithmic_bytes"] < 1.1  # measured HBM bytes = algorithmic bytes
            c = d["cpu_baseline"]
            assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "Mpixel/s" and "sample" in c and c["value"] > 0
            for side in ("glcm_step7", "hard_raster", "uint8_resident", "north_star_c2_16384", "c5_16384", "pcie_inclusive"):
                assert side in d["config"], side
            p = d["config"]["pcie_inclusive"]
            assert abs(p["compute_ms"] - d["ms_per_step"]) / d["ms_per_step"] < 0.05               # the serial pass times the step itself
            assert p["double_buffered"]["value"] > p["value"] and p["uint8_bands"]["double_buffered"]["value"] > p["uint8_bands"]["value"]


def test_bench_launcher_stops_the_other_ranks_when_one_dies():
    """One rank exits before the rendezvous, the others wait in it for ever: the launcher notices the dead child, stops the
    rest and returns non-zero within seconds (it used to block in rank 0's communicate())."""
    import time
    t0 = time.time()
    r = _bench("--gpus", "3", "--launch-selftest", env={"RSSEG_SELFTEST_DIE_RANK": "2"})
    assert r.returncode != 0 and "failed" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 120
    r = _bench("--gpus", "2", "--launch-selftest", env={"RSSEG_SELFTEST_DIE_RANK": "0"})      # rank 0 itself
    assert r.returncode != 0


# ---- the star-import surface: `from modules.features.indices import *` (scripts/2:20) and
# `from modules.features.extract import *` (scripts/3:25) are how the reference's scripts reach the path ----
OUT_OF_SCOPE_LIBRARY_NAMES = {"plt", "cv2"}     # matplotlib / OpenCV modules the reference leaks through import *: not re-exported
PLOT_NAMES = {"visualize_hierarchical_features", "create_classification_map", "visualize_combined_indices"}   # resolve, draw nothing


@pytest.mark.parametrize("script, mirror", [("scripts/2_feature_extraction.py", "modules.features.indices"),
                                            ("scripts/3_classification.py", "modules.features.extract")])
def test_star_import_resolves_every_name_the_reference_scripts_use(golden_dir, script, mirror):
    """tests/golden/star_import_names.json (oracle/gen_names.py: ast over the reference, identifiers only) lists what each
    script resolves through its star-import and what it imports by name.  Every one of them must come out of
    `from <mirror> import *`, except the two library modules above; the plot names resolve to functions that draw nothing."""
    import json
    spec = json.load(open(os.path.join(golden_dir, "star_import_names.json")))[script]
    assert spec["star_import_of"] == mirror and not spec["free_names_the_module_does_not_define"]
    ns = {}
    exec(f"from {mirror} import *", ns)                                   # noqa: S102 — what the scripts do
    wanted = set(spec["resolved_through_star_import"]) | set(spec["imported_by_name"])
    assert len(wanted) >= 15
    missing = sorted(n for n in wanted - OUT_OF_SCOPE_LIBRARY_NAMES if n not in ns)
    assert not missing, f"`from {mirror} import *` does not provide {missing}"
    for n in wanted & OUT_OF_SCOPE_LIBRARY_NAMES:
        assert spec["resolved_through_star_import"][n]["kind"] == "import"   # a library module, not a function of the path
    mod = __import__(mirror, fromlist=["*"])
    for n in sorted(wanted - OUT_OF_SCOPE_LIBRARY_NAMES):
        if spec["resolved_through_star_import"].get(n, {}).get("kind") == "function" or n in spec["imported_by_name"]:
            assert callable(ns[n]) and ns[n].__module__ == mirror, n
    assert set(mod.PLOTTING_NAMES) == {n for n in mod.__all__ if n in PLOT_NAMES | {"visualize_selected_features"}}
    assert (wanted & PLOT_NAMES) <= set(mod.PLOTTING_NAMES)
    for n in mod.PLOTTING_NAMES:          # resolve, print a line, draw and write nothing
        import inspect
        args = [None] * sum(p.default is inspect.Parameter.empty for p in inspect.signature(ns[n]).parameters.values())
        assert ns[n](*args) is None
    from modules.utils.set_chinese_font import set_chinese_font       # scripts/2:22, scripts/3:16
    assert set_chinese_font() is None


def test_names_fixture_is_what_the_generator_makes_today(golden_dir, tmp_path):
    """Build container only (skipped where /root/reference does not exist, e.g. on the GPU box): oracle/gen_names.py re-run on the
    reference gives the committed fixture."""
    import json
    import subprocess
    import sys as _sys
    if not os.path.isdir("/root/reference/scripts"):
        pytest.skip("the reference is not present here")
    src = open(os.path.join(ROOT, "oracle", "gen_names.py")).read().replace('OUT = os.path.join(REPO, "tests", "golden", "star_import_names.json")',
                                                                            f'OUT = {str(tmp_path / "names.json")!r}')
    gen = tmp_path / "gen_names_tmp.py"
    gen.write_text(src)
    subprocess.check_call([_sys.executable, str(gen)], stdout=subprocess.DEVNULL)
    assert json.load(open(tmp_path / "names.json")) == json.load(open(os.path.join(golden_dir, "star_import_names.json")))


def test_mirror_signatures_equal_the_reference_for_every_mirrored_function(golden_dir):
    """Positional order and defaults of every function both modules define, against the signature strings the fixture
    holds (names and default literals only)."""
    import inspect
    import json
    spec = json.load(open(os.path.join(golden_dir, "star_import_names.json")))
    checked = 0
    for mirror in ("modules.features.indices", "modules.features.extract"):
        mod = __import__(mirror, fromlist=["*"])
        for name, want in spec[mirror]["signatures"].items():
            fn = getattr(mod, name, None)
            if fn is None:
                continue          # a reference function the stages never call (Gabor, fusion helpers, ...): not mirrored
            got = [[p.name, None if p.default is inspect.Parameter.empty else repr(p.default)] for p in inspect.signature(fn).parameters.values()]
            norm = lambda sig: [[a, None if d is None else d.replace(" ", "").replace("math.pi", "np.pi")] for a, d in sig]   # noqa: E731
            g, w = norm(got), norm(want)
            for (ga, gd), (wa, wd) in zip(g, w):
                assert ga == wa, f"{mirror}.{name}: parameter {ga!r} != {wa!r}"
                if wd is not None and "np.pi" not in wd:
                    assert gd is not None and eval(gd) == eval(wd), f"{mirror}.{name}({wa}): default {gd} != {wd}"   # noqa: S307
            assert len(g) == len(w), f"{mirror}.{name}: {len(g)} parameters, the reference has {len(w)}"
            checked += 1
    assert checked >= 30


def test_save_classification_as_geotiff_follows_the_reference(tmp_path, capsys):
    """extract.py:778-833: empty result / incomplete metadata / shape mismatch -> message and no file; dtype by value range;
    nodata 0; LZW tiles; the file is read back by libtiff (Pillow)."""
    from modules.features.extract import save_classification_as_geotiff
    from rsseg.tiff import read_tiff, read_tiff_georef
    meta = {"transform": (30.0, 0.0, 500000.0, 0.0, -30.0, 4100000.0), "crs": "EPSG:32650", "width": 300, "height": 260}
    p = str(tmp_path / "c.tif")
    assert save_classification_as_geotiff(np.zeros((0, 0)), meta, p) is None and not os.path.exists(p)
    assert "分类结果为空" in capsys.readouterr().out
    assert save_classification_as_geotiff(np.ones((260, 300), np.uint8), {k: v for k, v in meta.items() if k != "crs"}, p) is None
    assert "元数据不完整" in capsys.readouterr().out and not os.path.exists(p)
    assert save_classification_as_geotiff(np.ones((10, 10), np.uint8), meta, p) is None and not os.path.exists(p)
    assert "does not match" in capsys.readouterr().out
    rng = np.random.default_rng(5)
    for hi, dt, src in ((7, np.uint8, np.int32), (300, np.uint16, np.int64), (70000, np.int32, np.int64), (5, np.uint8, np.float64)):
        a = rng.integers(0, hi + 1, (260, 300)).astype(src)
        a[0, 0] = hi
        save_classification_as_geotiff(a, meta, p)
        assert "已保存" in capsys.readouterr().out
        back = read_tiff(p)
        assert back.dtype == dt and back.shape == (1, 260, 300) and np.array_equal(back[0], a.astype(dt))
        g = read_tiff_georef(p)
        assert g["nodata"] == 0 and g["epsg"] == 32650 and g["transform"] == meta["transform"]
        from PIL import Image
        assert np.array_equal(np.asarray(Image.open(p)), a.astype(dt))
        os.remove(p)
    neg = np.full((260, 300), -3, np.int64)
    save_classification_as_geotiff(neg, meta, p)
    assert read_tiff(p).dtype == np.int32


@pytest.mark.parametrize("seed", range(12))
def test_tiff_and_lzw_randomised_round_trips(tmp_path, seed):
    """Random shapes (down to 1 x 1, widths and heights around the 256-pixel tile), dtypes, band counts, strip / tile / LZW /
    BigTIFF forms: what write_tiff writes, read_tiff reads back bit for bit, and for one-band 8-bit / float32 rasters Pillow's
    libtiff reads the same values; the LZW codec round-trips random byte strings of random alphabets and run lengths (table
    resets, the 4094-entry boundary)."""
    from PIL import Image
    from rsseg.tiff import lzw_decode, lzw_encode, read_tiff, write_tiff
    rng = np.random.default_rng(7000 + seed)
    for _ in range(6):
        n = int(rng.choice([0, 1, 2, 255, 256, 4093, 4094, 4095, 4096, 50000, 300000]))
        alphabet = int(rng.choice([1, 2, 3, 16, 256]))
        data = rng.integers(0, alphabet, n, dtype=np.uint8)
        if rng.random() < 0.5 and n > 10:                # long runs
            data = np.repeat(data[: max(n // 50, 1)], 50)[:n]
        data = bytes(data)
        assert lzw_decode(lzw_encode(data), len(data)) == data, (seed, n, alphabet)
    for _ in range(5):
        H = int(rng.choice([1, 2, 7, 255, 256, 257, 300, 513]))
        W = int(rng.choice([1, 3, 255, 256, 257, 400, 515]))
        B = int(rng.choice([1, 1, 2, 3, 7, 19]))
        dt = rng.choice(["u1", "u2", "i2", "i4", "f4", "f8"])
        if dt[0] == "f":
            a = rng.random((B, H, W)).astype(dt)
        else:
            hi = {"u1": 256, "u2": 65536, "i2": 32768, "i4": 2 ** 31}[str(dt)]
            a = rng.integers(-hi if dt[0] == "i" else 0, hi, (B, H, W)).astype(dt)
        if rng.random() < 0.4:
            a[:, : H // 2] = a.flat[0]                    # compressible halves
        kw = dict(compress=rng.choice([None, "lzw"]), tiled=[None, True, False][int(rng.integers(0, 3))], bigtiff=[None, True][int(rng.integers(0, 2))])
        p = str(tmp_path / "r.tif")
        write_tiff(p, a if B > 1 or rng.random() < 0.5 else a[0], **kw)
        back = read_tiff(p)
        assert back.dtype == a.dtype and np.array_equal(back, a), (seed, H, W, B, dt, kw)
        if B == 1 and dt in ("u1", "f4") and not kw["bigtiff"]:
            with Image.open(p) as im:
                assert np.array_equal(np.asarray(im), a[0]), (seed, H, W, dt, kw)
