"""Worker for the multi-rank tests (spawned by test_dist.py / test_gpu_dist.py).
argv: mode rank world port outdir"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rs-image-segmentation_amd"), ROOT):
    sys.path.insert(0, p)


def main():
    mode, rank, world, port, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    if mode == "rccl_single":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from rsseg import _lib as L
    from rsseg.runtime import make_allreduce_hook
    if mode == "hook_cpu":
        # the hook itself, on a CPU buffer: the protocol the library relies on (exact int64 limb sums,
        # MAX of negated minima, zero-padded one-owner broadcasts)
        buf = torch.zeros(4096, dtype=torch.uint8)
        hook = make_allreduce_hook(buf, None)
        i64 = buf[0:80].view(torch.int64)
        i64[:] = torch.arange(10) * (rank + 1) + (1 << 40) * rank
        assert hook(None, 0, 10, L.I64, L.SUM) == 0
        want = sum(np.arange(10) * (r + 1) + (1 << 40) * r for r in range(world))
        assert np.array_equal(i64.numpy(), want)
        f64 = buf[128:128 + 32].view(torch.float64)
        f64[:] = torch.tensor([-(rank + 0.5), rank + 0.25, 0.0, 1e300], dtype=torch.float64)
        assert hook(None, 128, 4, L.F64, L.MAX) == 0
        assert f64.tolist() == [-0.5, world - 1 + 0.25, 0.0, 1e300], f64.tolist()
        f32 = buf[256:256 + 8].view(torch.float32)
        f32[:] = torch.tensor([float(rank), 7.0])
        assert hook(None, 256, 2, L.F32, L.MIN) == 0
        assert f32.tolist() == [0.0, 7.0]
        own = buf[512:512 + 24].view(torch.float64)
        own[:] = torch.tensor([0.1, 0.2, 0.3], dtype=torch.float64) if rank == world - 1 else torch.zeros(3, dtype=torch.float64)
        assert hook(None, 512, 3, L.F64, L.SUM) == 0
        assert own.tolist() == [0.1, 0.2, 0.3]  # x + 0 is exact: a one-owner broadcast
        open(os.path.join(outdir, f"ok_{rank}"), "w").write("ok")
    elif mode == "gpu_shard":
        # both ranks share cuda:0 (gloo moves the small reduction buffers through the host); each holds
        # a row stripe of the scene; results must equal the single-rank run bit for bit
        from rsseg import pipeline as P
        from rsseg.runtime import Context
        data = np.load(os.path.join(outdir, "input.npz"))
        bands = data["bands"]
        H, W = bands.shape[1:]
        rows = [(H * r) // world for r in range(world + 1)]
        r0, r1 = rows[rank], rows[rank + 1]
        ctx = Context(0, use_dist=True)
        assert ctx.world == world
        dev = [ctx.to_device(bands[i, r0:r1].reshape(-1)) for i in range(bands.shape[0])]
        n_global = H * W
        lohi = P.band_lohi(ctx, dev, n_global)
        idx, norms = P.spectral_indices(ctx, dev, lohi, want_norm=(True,) * 5)
        norm_all = list(norms) + [ctx.normalize(dev[i], float(lohi[i, 0]), float(lohi[i, 1])) for i in range(5, len(dev))]
        pcs, ratio, model = P.pca(ctx, norm_all, 3, True, n_global)
        planes = [idx[n] for n in P.INDEX_NAMES] + list(pcs)
        labels, meta = ctx.kmeans_fit_predict(planes, int(data["k"]))
        np.savez(os.path.join(outdir, f"out_{rank}.npz"), labels=labels.cpu().numpy(), lohi=lohi, ratio=ratio,
                 pc0=pcs[0].cpu().numpy(), n_iter=meta["n_iter"], centers=meta["centers"], init=meta["init_indices"], r0=r0, r1=r1)
        ctx.close()
    elif mode == "gpu_striped_c3":
        from rsseg import pipeline as P
        from rsseg.runtime import Context
        data = np.load(os.path.join(outdir, "input.npz"))
        bands = data["bands"]
        H, W = bands.shape[1:]
        r0, r1 = P.stripe_rows(H, world, rank)
        j0, j1, i0, i1 = P.glcm_halo_rows(H, r0, r1, 7, 1)
        ctx = Context(0, use_dist=True)
        dev = [ctx.to_device(bands[i, r0:r1].reshape(-1)) for i in range(bands.shape[0])]
        nir_ext = ctx.to_device(bands[3, i0:i1].reshape(-1))
        labels, meta, planes = P.config3_striped(ctx, dev, nir_ext, H, W, r0, r1, i0, int(data["k"]))
        np.savez(os.path.join(outdir, f"out_{rank}.npz"), labels=labels.cpu().numpy(), n_iter=meta["n_iter"], r0=r0, r1=r1,
                 glcm0=planes[7].cpu().numpy(), glcm4=planes[11].cpu().numpy())
        ctx.close()
    elif mode == "gpu_striped_c5":
        # BASELINE config 5 on one raster sharded by rows: 19-feature stack with halos + forest labels
        from rsseg import pipeline as P
        from rsseg.runtime import Context
        data = np.load(os.path.join(outdir, "input.npz"))
        bands = data["bands"]
        H, W = bands.shape[1:]
        r0, r1 = P.stripe_rows(H, world, rank)
        e0, e1 = P.stack19_halo_rows(H, r0, r1)
        ctx = Context(0, use_dist=True)
        ext = [ctx.to_device(bands[i, e0:e1].reshape(-1)) for i in range(bands.shape[0])]
        planes, _ = P.stack19_striped(ctx, ext, H, W, r0, r1, e0)
        forest = {k[7:]: data[k] for k in data.files if k.startswith("forest_")}
        forest["n_features"] = int(forest["n_features"])
        ctx.forest_load(forest)
        labels = ctx.forest_predict(P.stack19_forest_planes(ctx, planes))
        np.savez(os.path.join(outdir, f"out_{rank}.npz"), labels=labels.cpu().numpy(), r0=r0, r1=r1,
                 **{f"p{i}": p.cpu().numpy() for i, p in enumerate(planes)})
        ctx.close()
    elif mode == "rccl_single":
        # ONE rank, backend nccl (= RCCL), the hook installed all the same: every collective of the step is an identity
        # reduction that really goes through torch.distributed / RCCL on this GPU, stream-ordered (no host wait in the
        # hook).  (1) the hook alone: ordered after earlier work on the context's stream, visible to later work there;
        # (2) config 3 and the 19-feature stack + forest through it, bit-identical to a context without a hook.
        from rsseg import pipeline as P
        from rsseg.runtime import Context
        side = torch.cuda.Stream()
        buf = torch.zeros(1 << 22, dtype=torch.uint8, device="cuda")
        hook = make_allreduce_hook(buf, None, side)
        big = torch.zeros(1 << 26, dtype=torch.int64, device="cuda")
        with torch.cuda.stream(side):
            big.add_(3)                                         # ~1 ms of work in front of the writer
            buf[:80].view(torch.int64).copy_(big[:10] * 5 + torch.arange(10, device="cuda"))
        assert hook(None, 0, 10, L.I64, L.SUM) == 0             # no host wait in between
        with torch.cuda.stream(side):
            got = buf[:80].view(torch.int64) + 1
        side.synchronize()
        assert got.tolist() == [16 + i for i in range(10)], got.tolist()
        # (1b) the same ordering test for RCCL driven by the library itself (rsseg_ctx_set_comm_rccl: ncclAllReduce on the
        # context's stream), on a context that lives on the side stream
        nctx = Context(0, use_dist=True, force_comm=True, comm="native", stream=side)
        assert nctx.comm_kind == "native" and nctx.world == 1
        nbuf = nctx._comm_buf
        with torch.cuda.stream(side):
            big.add_(4)
            nbuf[256:336].view(torch.int64).copy_(big[:10] * 3 + torch.arange(10, device="cuda"))
        nctx.allreduce(256, 10, L.I64, L.SUM)                   # enqueued behind the writer, no host wait
        with torch.cuda.stream(side):
            got = nbuf[256:336].view(torch.int64) * 2
        side.synchronize()
        assert got.tolist() == [2 * (21 + i) for i in range(10)], got.tolist()
        for dt, view, op, vals in ((L.F64, torch.float64, L.MAX, [-1.5, 2.25, 1e300]), (L.F64, torch.float64, L.MIN, [3.0, -0.0, 7.5]),
                                   (L.F32, torch.float32, L.SUM, [0.1, 0.2, 0.3])):
            esz = 4 if dt == L.F32 else 8
            with torch.cuda.stream(side):
                nbuf[:3 * esz].view(view).copy_(torch.tensor(vals, dtype=view))
            nctx.allreduce(0, 3, dt, op)
            side.synchronize()
            assert nbuf[:3 * esz].view(view).tolist() == torch.tensor(vals, dtype=view).tolist()
        try:
            nctx.allreduce(nbuf.numel() - 8, 2, L.I64, L.SUM)   # past the end of the buffer
            raise AssertionError("a range beyond the communication buffer was accepted")
        except ValueError:
            pass
        # argument checks of the native provider (no communicator is created by any of these)
        import ctypes as C
        bad = Context(0, use_dist=False)
        uid = (C.c_char * 128)()
        assert bad.lib.rsseg_rccl_unique_id(None, C.cast(uid, C.c_void_p)) == 0          # NULL path: librccl.so.1 as the process has it
        assert bad.lib.rsseg_ctx_set_comm_rccl(bad.h, 0, 1, None, None, None, 0) == -1                           # no unique id
        assert bad.lib.rsseg_ctx_set_comm_rccl(bad.h, 3, 2, C.cast(uid, C.c_void_p), None, None, 0) == -1        # rank outside the world
        assert bad.lib.rsseg_ctx_set_comm_rccl(bad.h, 0, 1, C.cast(uid, C.c_void_p), None, C.c_void_p(nbuf.data_ptr()), 4096) == -1   # buffer < 1 MiB
        assert bad.lib.rsseg_ctx_set_comm_rccl(bad.h, 0, 1, C.cast(uid, C.c_void_p), None, None, 0) == 0         # library-owned 4 MiB buffer
        bad.allreduce(0, 4, L.I64, L.SUM)
        assert bad.lib.rsseg_ctx_set_comm(bad.h, 0, 1, L.ALLREDUCE_FN(0), None, None, 0) == 0                     # releases the communicator
        bad.allreduce(0, 4, L.I64, L.SUM)                                                                        # no provider: the identity
        bad.close()
        nctx.close()
        data = np.load(os.path.join(outdir, "input.npz"))
        bands = data["bands"]
        H, W = bands.shape[1:]
        res = []
        for force, kind in ((False, None), (True, "torch"), (True, "native")):
            ctx = Context(0, use_dist=True, force_comm=force, comm=kind)
            assert ctx.world == 1 and ctx.comm_kind == kind
            ctx.prof_enable(True)
            dev = [ctx.to_device(bands[i].reshape(-1)) for i in range(bands.shape[0])]
            labels, meta, planes = P.config3(ctx, dev, H, W, int(data["k"]))
            st, _ = P.feature_stack19(ctx, dev, H, W)
            forest = {k[7:]: data[k] for k in data.files if k.startswith("forest_")}
            forest["n_features"] = int(forest["n_features"])
            ctx.forest_load(forest)
            fl = ctx.forest_predict(P.stack19_forest_planes(ctx, st))
            _, calls = ctx.prof_get("allreduce")
            res.append(dict(labels=labels.cpu().numpy(), centers=meta["centers"], n_iter=meta["n_iter"], init=meta["init_indices"],
                            planes=[p.cpu().numpy() for p in planes], st=[p.cpu().numpy() for p in st], fl=fl.cpu().numpy(), calls=calls))
            ctx.close()
        a = res[0]
        assert a["calls"] == 0
        for b in res[1:]:      # through torch.distributed's RCCL, then through the library's own communicator
            assert b["calls"] >= 20 and b["calls"] == res[1]["calls"], (a["calls"], b["calls"])
            assert np.array_equal(a["labels"], b["labels"]) and np.array_equal(a["centers"], b["centers"])
            assert a["n_iter"] == b["n_iter"] and np.array_equal(a["init"], b["init"])
            for x, y in zip(a["planes"] + a["st"], b["planes"] + b["st"]):
                assert np.array_equal(x, y, equal_nan=True)
            assert np.array_equal(a["fl"], b["fl"])
        open(os.path.join(outdir, "ok_rccl"), "w").write(str(b["calls"]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
