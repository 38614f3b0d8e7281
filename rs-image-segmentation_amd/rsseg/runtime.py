"""
Host runtime over the C ABI: one Context per process / GPU.

torch-ROCm is used for three things only: device allocations (tensors as buffers, `.data_ptr()`
handed to the library), the current HIP stream, and `torch.distributed` (backend "nccl" = RCCL over
xGMI) behind the library's all-reduce hook.  All arithmetic on rasters happens in librsseg_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L


class RssegError(RuntimeError):
    pass


class RssegUnsupported(RssegError):
    """RSSEG_ERR_UNSUPPORTED: a valid input beyond a capacity of the kernels (more than 64 features / classes, ...).
    Never swallowed by the mirrors: the reference would have produced a result, so an all-zero map would be a lie."""


def _torch():
    import torch
    return torch


def make_allreduce_hook(buf, group=None, stream=None):
    """The Python side of rsseg_allreduce_fn: reduces `count` elements of `dtype` at byte `offset` of the
    communication buffer `buf` (a uint8 tensor; on the GPU in production, so the collective is RCCL over
    xGMI; a CPU tensor with the gloo backend in the CPU tests) in place across the ranks of `group`.

    Stream contract (include/rsseg.h): the collective is ordered after the work already enqueued on the context's
    stream and its result is visible to work enqueued there afterwards.  torch.distributed gives exactly that for the
    CURRENT stream (ProcessGroupNCCL makes its communication stream wait for the current stream, and a synchronous
    collective makes the current stream wait for the communication stream — no host synchronisation), so the hook
    runs the collective with the context's stream current and does not block the host: the library pays one
    hipStreamSynchronize per collective, after its copy-back.  Backends other than RCCL (gloo rehearsals on a CUDA
    buffer) are followed by a stream synchronisation, since their stream semantics are not relied on."""
    torch = _torch()
    import torch.distributed as dist
    views = {L.F32: torch.float32, L.F64: torch.float64, L.I64: torch.int64}
    ops = {L.SUM: dist.ReduceOp.SUM, L.MIN: dist.ReduceOp.MIN, L.MAX: dist.ReduceOp.MAX}
    is_nccl = buf.is_cuda and dist.get_backend(group) == "nccl"

    def hook(_user, offset, count, dtype, op):
        try:
            esz = 4 if dtype == L.F32 else 8
            t = buf[offset:offset + count * esz].view(views[dtype])
            if buf.is_cuda and stream is not None:
                with torch.cuda.stream(stream):
                    dist.all_reduce(t, op=ops[op], group=group)
                    if not is_nccl:
                        stream.synchronize()
            else:
                dist.all_reduce(t, op=ops[op], group=group)
                if buf.is_cuda and not is_nccl:
                    torch.cuda.current_stream().synchronize()
            return 0
        except Exception as e:  # noqa: BLE001 — must not propagate through the C frame
            print(f"[rsseg] all-reduce hook failed: {e!r}", flush=True)
            return 1

    return hook


class Context:
    """Owns an rsseg_ctx.  `group` (optional) is a torch.distributed process group: when its world
    size is > 1 the library's reductions go through RCCL (or gloo in CPU tests of the hook)."""

    def __init__(self, device: int = 0, group=None, use_dist: Optional[bool] = None, stream=None, force_comm: bool = False,
                 comm: Optional[str] = None):
        """stream: a torch.cuda.Stream for this context (default: torch's current stream).
        force_comm: install the all-reduce path even when the group has ONE rank (identity reductions), so that every
        collective of a step runs through the backend — the way to exercise the RCCL path on a one-GPU box.
        comm: 'native' — the library drives RCCL itself (rsseg_ctx_set_comm_rccl: its own communicator, ncclAllReduce on the
        context's stream, no Python in the loop; torch.distributed only carries the 128-byte unique id once); 'torch' — the
        callback into torch.distributed.all_reduce.  Default ($RSSEG_COMM overrides): 'native' when the group's backend is
        nccl (= RCCL), 'torch' otherwise (gloo rehearsals)."""
        torch = _torch()
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise RssegError("no GPU visible to torch: the rsseg product path needs an MI355X (no CPU fallback)")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        self.torch_stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        # torch's default stream is the legacy null stream (handle 0).  NULL would make the library create a private
        # non-blocking stream that does NOT order after torch kernels still writing an input plane, so the legacy
        # default stream is named explicitly (hipStreamLegacy == (hipStream_t)1).
        handle = self.torch_stream.cuda_stream or 1
        rc = self.lib.rsseg_ctx_create(device, C.c_void_p(handle), C.byref(h))
        if rc != 0:
            raise RssegError(f"rsseg_ctx_create failed ({rc}): {self.lib.rsseg_last_error(None).decode()}")
        self.h = h
        self.rank, self.world = 0, 1
        self._comm_buf = None
        self._hook = None
        import torch.distributed as dist
        if use_dist is None:
            use_dist = dist.is_available() and dist.is_initialized()
        if force_comm and not use_dist:
            raise RssegError("force_comm needs an initialised torch.distributed process group")
        self.comm_kind = None
        if use_dist and (dist.get_world_size(group) > 1 or force_comm):
            kind = comm or os.environ.get("RSSEG_COMM") or ("native" if dist.get_backend(group) == "nccl" else "torch")
            if kind == "native-only":
                kind = "native"
            if kind not in ("native", "torch"):
                raise ValueError(f"comm={kind!r}: 'native' or 'torch'")
            if kind == "native":
                try:
                    self._install_comm_rccl(group)
                except (RssegError, OSError) as e:
                    # a failure every rank meets alike (librccl not loadable, a symbol missing, ncclCommInitRank refusing the
                    # topology): fall back to the callback provider LOUDLY rather than lose the run; RSSEG_COMM=native-only forbids it
                    if os.environ.get("RSSEG_COMM") == "native-only" or comm == "native":
                        raise
                    print(f"[rsseg] native RCCL provider unavailable ({e}); using the torch.distributed callback", flush=True)
                    kind = "torch"
                    self._install_comm(group)
            else:
                self._install_comm(group)
            self.comm_kind = kind

    # ---- communication hook ----------------------------------------------------------------
    def _install_comm(self, group):
        torch = _torch()
        import torch.distributed as dist
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._comm_buf = torch.zeros(1 << 22, dtype=torch.uint8, device=self.device)
        self._hook = L.ALLREDUCE_FN(make_allreduce_hook(self._comm_buf, group, self.torch_stream))
        self._chk(self.lib.rsseg_ctx_set_comm(self.h, self.rank, self.world, self._hook, None,
                                              C.c_void_p(self._comm_buf.data_ptr()), self._comm_buf.numel()))

    def _install_comm_rccl(self, group):
        """RCCL driven from C: rank 0 makes the ncclUniqueId, torch.distributed hands its 128 bytes to the other ranks (the
        only use of the process group), every rank joins the library's own communicator."""
        torch = _torch()
        import torch.distributed as dist
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        cpath = path.encode() if os.path.exists(path) else None
        box = [None]
        if self.rank == 0:
            raw = (C.c_char * 128)()
            rc = self.lib.rsseg_rccl_unique_id(cpath, C.cast(raw, C.c_void_p))
            # a failure here travels to every rank in place of the id, so that all of them take the same decision
            box[0] = bytes(raw.raw) if rc == 0 else f"rsseg_rccl_unique_id failed ({rc}): {self.lib.rsseg_last_error(None).decode()}"
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if not isinstance(box[0], bytes):
            raise RssegError(str(box[0]))
        uid = (C.c_char * 128).from_buffer_copy(box[0])
        self._comm_buf = torch.zeros(1 << 22, dtype=torch.uint8, device=self.device)
        torch.cuda.synchronize(self.device)
        self._chk(self.lib.rsseg_ctx_set_comm_rccl(self.h, self.rank, self.world, C.cast(uid, C.c_void_p), cpath,
                                                   C.c_void_p(self._comm_buf.data_ptr()), self._comm_buf.numel()))

    def allreduce(self, offset: int, count: int, dtype: int, op: int):
        """One in-place reduction of `count` elements at byte `offset` of the communication buffer (self._comm_buf) over the
        installed path, enqueued on this context's stream (rsseg_ctx_allreduce)."""
        self._chk(self.lib.rsseg_ctx_allreduce(self.h, offset, count, dtype, op))

    def install_comm_hook(self, rank: int, world: int, hook):
        """A caller-supplied all-reduce instead of torch.distributed: hook(buf, offset, count, dtype, op) -> None reduces
        `count` elements of `dtype` (rsseg._lib F32 / F64 / I64) at byte `offset` of the uint8 device tensor `buf` in place
        across the caller's ranks, ordered after the work on this context's stream (rsseg_allreduce_fn's contract).  Used
        by the test-suite to run many ranks as threads of one process on one GPU."""
        torch = _torch()
        self.rank, self.world = int(rank), int(world)
        self._comm_buf = torch.zeros(1 << 22, dtype=torch.uint8, device=self.device)
        buf = self._comm_buf

        def cb(_user, offset, count, dtype, op):
            try:
                hook(buf, offset, count, dtype, op)
                return 0
            except Exception as e:  # noqa: BLE001 — must not propagate through the C frame
                print(f"[rsseg] all-reduce hook failed: {e!r}", flush=True)
                return 1

        self._hook = L.ALLREDUCE_FN(cb)
        self._chk(self.lib.rsseg_ctx_set_comm(self.h, self.rank, self.world, self._hook, None,
                                              C.c_void_p(self._comm_buf.data_ptr()), self._comm_buf.numel()))

    def _needs_keep(self) -> bool:
        """A temporary dropped by Python goes back to torch's caching allocator, which reuses it in the order of the stream
        it was ALLOCATED on (torch's current stream).  When that is this context's stream — the default — queued kernels
        that still read the block run before anything that could overwrite it, so nothing has to be kept alive and the
        pipelines free their temporaries as they go (keeping everything until end_async() raised the peak by
        several planes).  Only a context on a side stream (aux()), or one that has handed work to such a stream, must hold on to them until sync()."""
        if not getattr(self, "_async", False):
            return False
        if getattr(self, "_aux", None) is not None:
            return True     # a second stream allocates from the same pool: a block freed here could be rewritten there too early
        return self.torch_stream != _torch().cuda.current_stream(self.device)

    def set_async(self, on: bool = True):
        """Asynchronous entry points.  On a side stream, buffers handed out by empty() are kept alive until sync(): the
        caching allocator would otherwise recycle a temporary the moment Python drops it, while kernels that
        read it are still queued on this context's stream (_needs_keep)."""
        self._chk(self.lib.rsseg_ctx_set_async(self.h, int(on)))
        self._async = bool(on)
        self._keep = []

    def sync(self):
        self._chk(self.lib.rsseg_ctx_sync(self.h))
        self._keep = []

    def end_async(self):
        """Back to synchronous entry points.  The buffers kept alive for queued kernels are released to torch's caching
        allocator, which recycles them in stream order: safe when this context's stream is the stream they were allocated
        on (torch's current stream, the default); a context on a side stream (aux()) must use sync() instead."""
        self._chk(self.lib.rsseg_ctx_set_async(self.h, 0))
        self._async = False
        if self.torch_stream != _torch().cuda.current_stream(self.device):
            self._chk(self.lib.rsseg_ctx_sync(self.h))
        self._keep = []

    def aux(self):
        """A second context on its own HIP stream (same device, asynchronous entry points): lets a
        VALU-bound kernel run beside the HBM-bound passes issued through this context."""
        if getattr(self, "_aux", None) is None:
            torch = _torch()
            a = Context(self.device.index, use_dist=False, stream=torch.cuda.Stream(self.device))
            a.set_async(True)
            self._aux = a
        return self._aux

    def close(self):
        if getattr(self, "_aux", None) is not None:
            self._aux.close()
            self._aux = None
        if getattr(self, "h", None):
            self.lib.rsseg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _chk(self, rc: int):
        if rc != 0:
            msg = self.lib.rsseg_last_error(self.h).decode(errors="replace")
            if rc == -1:
                raise ValueError(msg)
            if rc == -3:
                raise MemoryError(msg)
            if rc == -5:
                raise RssegUnsupported(msg)
            raise RssegError(f"rsseg error {rc}: {msg}")

    # ---- buffers --------------------------------------------------------------------------------
    def to_device(self, a: np.ndarray, dtype=None):
        torch = _torch()
        a = np.ascontiguousarray(a, dtype=dtype)
        return torch.from_numpy(a).to(self.device)

    def upload_f32(self, a: np.ndarray):
        """A band as a flat float32 device plane — the reference's `.astype(np.float32)` (extract.py:34).  An 8-bit raster
        crosses PCIe as 1 byte per pixel and is widened on the device (exact: every uint8 is a float32)."""
        torch = _torch()
        a = np.ascontiguousarray(a).reshape(-1)
        if a.dtype == np.uint8:
            # widened by the library's own kernel, on THIS context's stream (where the kernels that read the plane are enqueued)
            with torch.cuda.stream(self.torch_stream):
                q = torch.from_numpy(a).to(self.device)
            return self.widen_u8(q)
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    def widen_u8(self, q):
        """uint8 plane -> float32 plane (exact), rsseg_u8_to_f32."""
        torch = _torch()
        out = self.empty(q.numel(), torch.float32)
        if self._needs_keep():
            self._keep.append(q)
        self._chk(self.lib.rsseg_u8_to_f32(self.h, C.c_void_p(q.data_ptr()), q.numel(), C.c_void_p(out.data_ptr())))
        return out

    def upload_band(self, a: np.ndarray):
        """A band as a flat device plane in the narrowest form the kernels take: an 8-bit raster STAYS uint8 (order
        statistics, spectral indices and PCA read 1 byte per pixel; values identical to the float32 path on the widened
        band), anything else becomes float32 like the reference's `.astype(np.float32)` (scripts/2:156)."""
        torch = _torch()
        a = np.ascontiguousarray(a).reshape(-1)
        if a.dtype == np.uint8:
            with torch.cuda.stream(self.torch_stream):
                return torch.from_numpy(a).to(self.device)
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

    @staticmethod
    def _is_u8(t) -> bool:
        return t.dtype == _torch().uint8

    def empty(self, n, dtype):
        torch = _torch()
        t = torch.empty(int(n), dtype=dtype, device=self.device)
        if self._needs_keep():
            self._keep.append(t)
        return t

    @staticmethod
    def _pp(tensors: Sequence) -> "C.Array":
        arr = (C.c_void_p * len(tensors))()
        for i, t in enumerate(tensors):
            arr[i] = None if t is None else t.data_ptr()
        return arr

    # ---- extrema of produced planes -------------------------------------------------------------
    def collect_minmax(self, on: bool = True):
        """While on, spectral_indices / resize_bilinear(_rows) / pca_fit_transform tag every plane they return with
        its (min, max) — attribute `_rsseg_minmax`, NaN counted as 0 — and kmeans_fit_predict skips its MinMaxScaler
        pass when all its planes carry the tag.  The tag describes the plane as produced: do not modify a tagged plane."""
        self._chk(self.lib.rsseg_ctx_collect_minmax(self.h, int(on)))
        self._collect = bool(on)

    def _tag_minmax(self, tensors: Sequence):
        if not getattr(self, "_collect", False):
            return
        for i, t in enumerate(tensors):
            if t is None:
                continue
            mn, mx = C.c_double(0), C.c_double(0)
            self._chk(self.lib.rsseg_ctx_last_minmax(self.h, i, C.byref(mn), C.byref(mx)))
            t._rsseg_minmax = (mn.value, mx.value)

    # ---- profiling ---------------------------------------------------------------------------
    def prof_enable(self, on=True):
        self._chk(self.lib.rsseg_prof_enable(self.h, int(on)))

    def prof_reset(self):
        self._chk(self.lib.rsseg_prof_reset(self.h))

    def host_syncs(self, reset: bool = False) -> int:
        """How often the library made the host wait for this context's stream since the last reset."""
        n = C.c_int64(0)
        self._chk(self.lib.rsseg_ctx_host_syncs(self.h, int(reset), C.byref(n)))
        return n.value

    def prof_get(self, name: str) -> Tuple[float, int]:
        ms, cnt = C.c_double(0), C.c_int64(0)
        self._chk(self.lib.rsseg_prof_get(self.h, name.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    # ---- K1 ------------------------------------------------------------------------------------
    def order_stats(self, plane, ranks: Sequence[int]) -> Tuple[np.ndarray, int]:
        if self._is_u8(plane):
            v, nn = self.order_stats_multi([plane], [list(ranks)])
            return v[0], int(nn[0])
        r = (C.c_int64 * len(ranks))(*[int(x) for x in ranks])
        out = (C.c_float * len(ranks))()
        nn = C.c_int64(0)
        self._chk(self.lib.rsseg_order_stats_f32(self.h, C.c_void_p(plane.data_ptr()), plane.numel(), r, len(ranks), out,
                                                 C.byref(nn)))
        return np.array(out[:], dtype=np.float32), nn.value

    def order_stats_multi(self, planes: Sequence, ranks: Sequence[Sequence[int]]) -> Tuple[np.ndarray, np.ndarray]:
        """Order statistics of several planes of equal length in one call (<= 8 planes, the same number of ranks for
        each): (values [P, R] float32, n_nan [P])."""
        P, R = len(planes), len(ranks[0])
        if any(len(r) != R for r in ranks) or any(p.numel() != planes[0].numel() for p in planes):
            raise ValueError("order_stats_multi: planes / rank lists of unequal length")
        flat = (C.c_int64 * (P * R))(*[int(x) for r in ranks for x in r])
        out = (C.c_float * (P * R))()
        nn = (C.c_int64 * P)()
        if any(self._is_u8(p) for p in planes):
            if not all(self._is_u8(p) for p in planes):
                raise ValueError("order_stats_multi: uint8 and float32 planes cannot be mixed in one call")
            self._chk(self.lib.rsseg_order_stats_multi_u8(self.h, self._pp(planes), P, planes[0].numel(), flat, R, out, nn))
        else:
            self._chk(self.lib.rsseg_order_stats_multi_f32(self.h, self._pp(planes), P, planes[0].numel(), flat, R, out, nn))
        return np.array(out[:], dtype=np.float32).reshape(P, R), np.array(nn[:], dtype=np.int64)

    # ---- K2 ------------------------------------------------------------------------------------
    def normalize(self, plane, lo: float, hi: float, out=None):
        torch = _torch()
        if self._is_u8(plane):   # only the striped / NaN side paths normalise a raw band on its own: widen it first
            plane = self.widen_u8(plane)
        out = self.empty(plane.numel(), plane.dtype) if out is None else out
        self._chk(self.lib.rsseg_normalize_f32(self.h, C.c_void_p(plane.data_ptr()), plane.numel(), C.c_float(lo),
                                               C.c_float(hi), C.c_void_p(out.data_ptr())))
        return out

    def spectral_indices(self, bands5: Sequence, lohi: Optional[np.ndarray], want_norm: Sequence[bool] = (False,) * 5,
                         want: Sequence[bool] = (True,) * 7, evi_coef: Optional[Sequence[float]] = None):
        torch = _torch()
        n = bands5[0].numel()
        outs = [self.empty(n, torch.float32) if w else None for w in want]
        norms = [self.empty(n, torch.float32) if w else None for w in want_norm]
        lh = None
        if lohi is not None:
            lh = (C.c_float * 10)(*[float(v) for v in np.asarray(lohi, np.float32).reshape(-1)])
        ec = None if evi_coef is None else (C.c_float * 4)(*[float(np.float32(v)) for v in evi_coef])   # {L, C1, C2, G}
        u8 = [self._is_u8(b) for b in bands5]
        if any(u8) and not all(u8):
            raise ValueError("spectral_indices: uint8 and float32 bands cannot be mixed")
        fn = self.lib.rsseg_spectral_indices_evi_u8 if all(u8) else self.lib.rsseg_spectral_indices_evi_f32
        self._chk(fn(self.h, self._pp(bands5), n, lh, self._pp(outs), self._pp(norms), ec))
        self._tag_minmax(outs)
        return outs, norms

    def quantize_u8(self, plane, mult: float):
        torch = _torch()
        q = self.empty(plane.numel(), torch.uint8)
        self._chk(self.lib.rsseg_quantize_u8(self.h, C.c_void_p(plane.data_ptr()), plane.numel(), C.c_float(mult),
                                             C.c_void_p(q.data_ptr())))
        return q

    def normalize_quantize_u8(self, plane, lo: float, hi: float, mult: float):
        """trunc(robust_normalize(plane; lo, hi) * mult) as uint8 in one pass."""
        torch = _torch()
        q = self.empty(plane.numel(), torch.uint8)
        self._chk(self.lib.rsseg_normalize_quantize_u8(self.h, C.c_void_p(plane.data_ptr()), plane.numel(), C.c_float(lo), C.c_float(hi),
                                                       C.c_float(mult), C.c_void_p(q.data_ptr())))
        return q

    def u8_to_unit(self, q):
        torch = _torch()
        out = self.empty(q.numel(), torch.float32)
        self._chk(self.lib.rsseg_u8_to_unit_f32(self.h, C.c_void_p(q.data_ptr()), q.numel(), C.c_void_p(out.data_ptr())))
        return out

    # ---- K3 ------------------------------------------------------------------------------------
    def pca_fit_transform(self, bands: Sequence, center: Optional[np.ndarray], scale: Optional[np.ndarray],
                          n_components: int, lohi: Optional[np.ndarray] = None, fit: Optional[Tuple[int, int]] = None):
        """lohi (nb x 2 float32, optional): the bands are RAW and are robust-normalised with these percentiles on the fly.
        fit=(offset, count) (optional): fit on that pixel range of the planes only, project all of them (stripe + halo rows)."""
        torch = _torch()
        nb, n = len(bands), bands[0].numel()
        outs = [self.empty(n, torch.float32) for _ in range(n_components)]
        comp = np.zeros((n_components, nb), np.float32)
        ratio = np.zeros(n_components, np.float32)
        mean = np.zeros(nb, np.float32)
        ev = np.zeros(n_components, np.float32)
        fp = C.POINTER(C.c_float)
        cptr = None if center is None else np.ascontiguousarray(center, np.float32).ctypes.data_as(fp)
        sc64 = None if scale is None else np.ascontiguousarray(scale, np.float64)
        sptr = None if sc64 is None else sc64.ctypes.data_as(C.POINTER(C.c_double))
        lh = None
        if lohi is not None:
            lh = np.ascontiguousarray(lohi, np.float32).reshape(-1)
            if lh.size != 2 * nb:
                raise ValueError("pca_fit_transform: lohi must hold (lo, hi) for every band")
        u8 = [self._is_u8(b) for b in bands]
        if any(u8) and not all(u8):
            raise ValueError("pca_fit_transform: uint8 and float32 bands cannot be mixed")
        if all(u8):
            f0, fn_ = (0, n) if fit is None else (int(fit[0]), int(fit[1]))
            self._chk(self.lib.rsseg_pca_fit_transform_ext_u8(self.h, self._pp(bands), nb, n, f0, fn_,
                                                              None if lh is None else lh.ctypes.data_as(fp), cptr, sptr, n_components,
                                                              self._pp(outs), comp.ctypes.data_as(fp), ratio.ctypes.data_as(fp),
                                                              mean.ctypes.data_as(fp), ev.ctypes.data_as(fp)))
        elif fit is not None:
            self._chk(self.lib.rsseg_pca_fit_transform_ext_f32(self.h, self._pp(bands), nb, n, int(fit[0]), int(fit[1]),
                                                               None if lh is None else lh.ctypes.data_as(fp), cptr, sptr, n_components,
                                                               self._pp(outs), comp.ctypes.data_as(fp), ratio.ctypes.data_as(fp),
                                                               mean.ctypes.data_as(fp), ev.ctypes.data_as(fp)))
        elif lohi is None:
            self._chk(self.lib.rsseg_pca_fit_transform_f32(self.h, self._pp(bands), nb, n, cptr, sptr, n_components,
                                                           self._pp(outs), comp.ctypes.data_as(fp), ratio.ctypes.data_as(fp),
                                                           mean.ctypes.data_as(fp), ev.ctypes.data_as(fp)))
        else:
            self._chk(self.lib.rsseg_pca_fit_transform_raw_f32(self.h, self._pp(bands), nb, n, lh.ctypes.data_as(fp), cptr, sptr, n_components,
                                                               self._pp(outs), comp.ctypes.data_as(fp), ratio.ctypes.data_as(fp),
                                                               mean.ctypes.data_as(fp), ev.ctypes.data_as(fp)))
        self._tag_minmax(outs)
        return outs, comp, ratio, mean, ev

    def indices_pca(self, bands: Sequence, lohi: np.ndarray, center: Optional[np.ndarray], scale: Optional[np.ndarray], n_components: int,
                    want_norm: Sequence[bool] = (False,) * 5, fit: Optional[Tuple[int, int]] = None, evi_coef: Optional[Sequence[float]] = None,
                    quantize: Optional[Tuple[float, float, float]] = None):
        """The seven spectral indices and the PCA of the same RAW bands with one pass less than spectral_indices +
        pca_fit_transform (rsseg_indices_pca_*): returns (index planes [7], normalised bands [5] (None where not wanted),
        component planes, components, ratio, mean, explained_variance) — the same bits as the two separate calls.
        quantize=(lo2, hi2, mult): additionally the uint8 plane normalize_quantize_u8(normalised NIR, lo2, hi2, mult) — the
        texture chain's input — left in `self.last_quantized`."""
        torch = _torch()
        nb, n = len(bands), bands[0].numel()
        idx = [self.empty(n, torch.float32) for _ in range(7)]
        norms = [self.empty(n, torch.float32) if w else None for w in want_norm]
        pcs = [self.empty(n, torch.float32) for _ in range(n_components)]
        comp = np.zeros((n_components, nb), np.float32)
        ratio = np.zeros(n_components, np.float32)
        mean = np.zeros(nb, np.float32)
        ev = np.zeros(n_components, np.float32)
        fp = C.POINTER(C.c_float)
        lh = np.ascontiguousarray(lohi, np.float32).reshape(-1)
        if lh.size != 2 * nb:
            raise ValueError("indices_pca: lohi must hold (lo, hi) for every band")
        cptr = None if center is None else np.ascontiguousarray(center, np.float32).ctypes.data_as(fp)
        sc64 = None if scale is None else np.ascontiguousarray(scale, np.float64)
        sptr = None if sc64 is None else sc64.ctypes.data_as(C.POINTER(C.c_double))
        ec = None if evi_coef is None else (C.c_float * 4)(*[float(np.float32(v)) for v in evi_coef])
        u8 = [self._is_u8(b) for b in bands]
        if any(u8) and not all(u8):
            raise ValueError("indices_pca: uint8 and float32 bands cannot be mixed")
        f0, fn_ = (0, n) if fit is None else (int(fit[0]), int(fit[1]))
        fn = self.lib.rsseg_indices_pca_u8 if all(u8) else self.lib.rsseg_indices_pca_f32
        q = self.empty(n, torch.uint8) if quantize is not None else None
        qlo, qhi, qm = (0.0, 1.0, 1.0) if quantize is None else [float(v) for v in quantize]
        self._chk(fn(self.h, self._pp(bands), nb, n, f0, fn_, lh.ctypes.data_as(fp), cptr, sptr, n_components, ec, self._pp(idx), self._pp(norms),
                     self._pp(pcs), None if q is None else C.c_void_p(q.data_ptr()), C.c_float(qlo), C.c_float(qhi), C.c_float(qm),
                     comp.ctypes.data_as(fp), ratio.ctypes.data_as(fp), mean.ctypes.data_as(fp), ev.ctypes.data_as(fp)))
        self._tag_minmax(idx + pcs)
        self.last_quantized = q      # the uint8 texture-chain input of this call (None unless `quantize` was given)
        return idx, norms, pcs, comp, ratio, mean, ev

    # ---- K4..K8 --------------------------------------------------------------------------------
    def glcm(self, q, H: int, W: int, levels: int, win: int, step: int):
        torch = _torch()
        oh, ow = (H - win) // step + 1, (W - win) // step + 1
        outs = [self.empty(oh * ow, torch.float32) for _ in range(5)]
        self._chk(self.lib.rsseg_glcm_u8(self.h, C.c_void_p(q.data_ptr()), H, W, levels, win, step, self._pp(outs)))
        return outs, (oh, ow)

    def resize_bilinear(self, src, sh: int, sw: int, dh: int, dw: int):
        torch = _torch()
        dst = self.empty(dh * dw, torch.float32)
        self._chk(self.lib.rsseg_resize_bilinear_f32(self.h, C.c_void_p(src.data_ptr()), sh, sw, C.c_void_p(dst.data_ptr()),
                                                     dh, dw))
        self._tag_minmax([dst])
        return dst

    def resize_bilinear_rows(self, src, sh_local: int, sw: int, src_row0: int, sh: int, dh_local: int, dw: int, dst_row0: int, dh: int):
        torch = _torch()
        dst = self.empty(dh_local * dw, torch.float32)
        self._chk(self.lib.rsseg_resize_bilinear_rows_f32(self.h, C.c_void_p(src.data_ptr()), sh_local, sw, src_row0, sh,
                                                          C.c_void_p(dst.data_ptr()), dh_local, dw, dst_row0, dh))
        self._tag_minmax([dst])
        return dst

    # The window operators take `rows=(y0, y1)` and `edges` for row-sharded rasters: the plane holds H rows, rows
    # [y0, y1) are produced (compact output); edges bit 0 / 1 = row 0 / row H - 1 is a true image edge (include/rsseg.h).
    @staticmethod
    def _rows(H, rows):
        return (0, H) if rows is None else (int(rows[0]), int(rows[1]))

    def resize_bilinear_multi(self, srcs: Sequence, sh_local: int, sw: int, src_row0: int, sh: int, dh_local: int, dw: int, dst_row0: int, dh: int):
        """cv2.resize(INTER_LINEAR) of up to 8 maps of one shape in one launch (rows form; full maps: src_row0 = dst_row0 = 0)."""
        torch = _torch()
        dsts = [self.empty(dh_local * dw, torch.float32) for _ in srcs]
        self._chk(self.lib.rsseg_resize_bilinear_rows_multi_f32(self.h, self._pp(srcs), len(srcs), sh_local, sw, src_row0, sh, self._pp(dsts), dh_local,
                                                                dw, dst_row0, dh))
        self._tag_minmax(dsts)
        return dsts

    def box_mean(self, plane, H: int, W: int, k: int, border: int, square: bool = False, rows=None, edges: int = 3):
        return self.box_mean_multi([plane], H, W, k, border, square, rows, edges)[0]

    def box_mean_multi(self, planes: Sequence, H: int, W: int, k: int, border: int, square: bool = False, rows=None, edges: int = 3):
        """k x k box mean of up to 8 planes of equal shape in one launch (add_spatial_context, indices.py:760-776)."""
        torch = _torch()
        y0, y1 = self._rows(H, rows)
        outs = []
        for g in range(0, len(planes), 8):
            grp = list(planes[g:g + 8])
            o = [self.empty((y1 - y0) * W, torch.float32) for _ in grp]
            self._chk(self.lib.rsseg_box_mean_rows_f32(self.h, self._pp(grp), len(grp), H, W, y0, y1, edges, k, border, int(square), self._pp(o)))
            outs += o
        return outs

    def local_std(self, plane, H: int, W: int, k: int, rows=None, edges: int = 3, variance: bool = False):
        torch = _torch()
        y0, y1 = self._rows(H, rows)
        out = self.empty((y1 - y0) * W, torch.float32)
        self._chk(self.lib.rsseg_local_std_rows_f32(self.h, C.c_void_p(plane.data_ptr()), H, W, y0, y1, edges, k, int(variance),
                                                    C.c_void_p(out.data_ptr())))
        return out

    def local_var(self, plane, H: int, W: int, k: int, rows=None, edges: int = 3):
        return self.local_std(plane, H, W, k, rows, edges, variance=True)

    def morph(self, q, H: int, W: int, k: int, op: int, rows=None, edges: int = 3):
        """op: L.MORPH_ERODE / DILATE / OPEN / CLOSE / GRADIENT on a uint8 plane; uint8 result."""
        torch = _torch()
        y0, y1 = self._rows(H, rows)
        out = self.empty((y1 - y0) * W, torch.uint8)
        self._chk(self.lib.rsseg_morph_rows_u8(self.h, C.c_void_p(q.data_ptr()), H, W, y0, y1, edges, k, op, C.c_void_p(out.data_ptr())))
        return out

    def morph_gradient(self, q, H: int, W: int, k: int, rows=None, edges: int = 3):
        return self.morph(q, H, W, k, L.MORPH_GRADIENT, rows, edges)

    def laplacian_norm(self, q, H: int, W: int, rows=None, edges: int = 3):
        torch = _torch()
        y0, y1 = self._rows(H, rows)
        out = self.empty((y1 - y0) * W, torch.float32)
        self._chk(self.lib.rsseg_laplacian_norm_rows_u8(self.h, C.c_void_p(q.data_ptr()), H, W, y0, y1, edges, C.c_void_p(out.data_ptr())))
        return out

    def sobel_mag(self, q, H: int, W: int, rows=None, edges: int = 3):
        torch = _torch()
        y0, y1 = self._rows(H, rows)
        out = self.empty((y1 - y0) * W, torch.float32)
        self._chk(self.lib.rsseg_sobel_mag_rows_u8(self.h, C.c_void_p(q.data_ptr()), H, W, y0, y1, edges, C.c_void_p(out.data_ptr())))
        return out

    # ---- K13: LBP, rank entropy, fixed-point Gaussian ---------------------------------------------
    def lbp_uniform(self, q, H: int, W: int, n_points: int = 24, radius: float = 3):
        torch = _torch()
        out = self.empty(H * W, torch.uint8)
        self._chk(self.lib.rsseg_lbp_uniform_u8(self.h, C.c_void_p(q.data_ptr()), H, W, n_points, C.c_double(radius), C.c_void_p(out.data_ptr())))
        return out

    def rank_entropy(self, q, H: int, W: int, radius: int):
        torch = _torch()
        out = self.empty(H * W, torch.float64)
        self._chk(self.lib.rsseg_rank_entropy_u8(self.h, C.c_void_p(q.data_ptr()), H, W, radius, C.c_void_p(out.data_ptr())))
        return out

    def gaussian_blur_u8(self, q, H: int, W: int, ksize: int):
        torch = _torch()
        out = self.empty(H * W, torch.uint8)
        self._chk(self.lib.rsseg_gaussian_blur_u8(self.h, C.c_void_p(q.data_ptr()), H, W, ksize, C.c_void_p(out.data_ptr())))
        return out

    # ---- K12: rule-based classification -------------------------------------------------------
    def threshold_band(self, plane, lo: float = float("-inf"), hi: float = float("inf"), nan_as_zero: bool = True):
        """uint8 mask: 1 where lo < x < hi.  nan_as_zero: NaN counts as 0 (threshold_segmentation, extract.py:354-356);
        False: a NaN pixel is outside every interval (the plain comparisons of extract_bareland_by_rule, extract.py:486-497)."""
        torch = _torch()
        out = self.empty(plane.numel(), torch.uint8)
        if plane.dtype == torch.float64:
            self._chk(self.lib.rsseg_band_interval_f64(self.h, C.c_void_p(plane.data_ptr()), plane.numel(), C.c_double(lo), C.c_double(hi),
                                                       int(nan_as_zero), C.c_void_p(out.data_ptr())))
            return out
        self._chk(self.lib.rsseg_band_interval_f32(self.h, C.c_void_p(plane.data_ptr()), plane.numel(), C.c_float(lo), C.c_float(hi),
                                                   int(nan_as_zero), C.c_void_p(out.data_ptr())))
        return out

    def otsu_mask(self, plane, above: bool = True):
        """threshold_segmentation(..., otsu=True) (extract.py:358-371) of a float32 / float64 plane -> (uint8 mask, level or -1
        when the plane has no contrast, min, max)."""
        torch = _torch()
        if plane.dtype not in (torch.float32, torch.float64):
            raise ValueError("otsu_mask: float32 or float64 plane expected")
        out = self.empty(plane.numel(), torch.uint8)
        level, mn, mx = C.c_int(0), C.c_double(0), C.c_double(0)
        self._chk(self.lib.rsseg_otsu_mask(self.h, C.c_void_p(plane.data_ptr()), L.F32 if plane.dtype == torch.float32 else L.F64,
                                           plane.numel(), int(bool(above)), C.c_void_p(out.data_ptr()), C.byref(level), C.byref(mn), C.byref(mx)))
        return out, level.value, mn.value, mx.value

    def fill_holes(self, mask, H: int, W: int):
        """scipy.ndimage.binary_fill_holes of a 0 / 1 uint8 plane (extract.py:314-316)."""
        torch = _torch()
        out = self.empty(H * W, torch.uint8)
        self._chk(self.lib.rsseg_fill_holes_u8(self.h, C.c_void_p(mask.data_ptr()), H, W, C.c_void_p(out.data_ptr())))
        return out

    def mask_op(self, a, b, op: int):
        torch = _torch()
        out = self.empty(a.numel(), torch.uint8)
        self._chk(self.lib.rsseg_mask_op_u8(self.h, C.c_void_p(a.data_ptr()), None if b is None else C.c_void_p(b.data_ptr()), a.numel(), op,
                                            C.c_void_p(out.data_ptr())))
        return out

    def mask_paint(self, final_map, mask, value: int, only_unset: bool = False):
        self._chk(self.lib.rsseg_mask_paint_u8(self.h, C.c_void_p(final_map.data_ptr()), C.c_void_p(mask.data_ptr()), final_map.numel(), value,
                                               int(only_unset)))
        return final_map

    def morph_ellipse(self, q, H: int, W: int, k: int, op: int):
        torch = _torch()
        out = self.empty(H * W, torch.uint8)
        self._chk(self.lib.rsseg_morph_ellipse_u8(self.h, C.c_void_p(q.data_ptr()), H, W, k, op, C.c_void_p(out.data_ptr())))
        return out

    def remove_small_components(self, mask, H: int, W: int, min_area: int):
        torch = _torch()
        out = self.empty(H * W, torch.uint8)
        self._chk(self.lib.rsseg_remove_small_components_u8(self.h, C.c_void_p(mask.data_ptr()), H, W, int(min_area), C.c_void_p(out.data_ptr())))
        return out

    # ---- K9/K10 --------------------------------------------------------------------------------
    def kmeans_fit_predict(self, planes: Sequence, n_clusters: int, seed: int = 42, max_iter: int = 300, tol: float = 1e-4):
        torch = _torch()
        F, n = len(planes), planes[0].numel()
        dt = planes[0].dtype
        if any(p.dtype != dt for p in planes) or dt not in (torch.float32, torch.float64):
            raise ValueError("kmeans planes must all be float32 or all float64")
        labels = self.empty(max(n, 1), torch.int32)
        centers = np.zeros((n_clusters, F), np.float64)
        info = L.KMeansInfo()
        tags = [getattr(p, "_rsseg_minmax", None) for p in planes]
        if dt == torch.float32 and n > 0 and all(t is not None for t in tags):
            lmin = (C.c_double * F)(*[t[0] for t in tags])
            lmax = (C.c_double * F)(*[t[1] for t in tags])
            self._chk(self.lib.rsseg_kmeans_fit_predict_mm(self.h, self._pp(planes), F, L.F32, n, n_clusters, seed, max_iter, tol,
                                                           C.c_void_p(labels.data_ptr()), centers.ctypes.data_as(C.POINTER(C.c_double)),
                                                           C.byref(info), lmin, lmax))
        else:
            self._chk(self.lib.rsseg_kmeans_fit_predict(self.h, self._pp(planes), F, L.F32 if dt == torch.float32 else L.F64, n,
                                                        n_clusters, seed, max_iter, tol, C.c_void_p(labels.data_ptr()),
                                                        centers.ctypes.data_as(C.POINTER(C.c_double)), C.byref(info)))
        meta = dict(n_iter=info.n_iter, tol=info.tol, relocated=info.relocated,
                    scale=np.array(info.scale[:F]), min=np.array(info.min[:F]), mean=np.array(info.mean[:F]),
                    init_indices=np.array(info.init_indices[:n_clusters]), ms_init=info.ms_init, ms_lloyd=info.ms_lloyd,
                    centers=centers)
        return labels[:n], meta

    # ---- K11 -----------------------------------------------------------------------------------
    def forest_load(self, forest: dict):
        f = forest
        ip, lp, dp = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)
        keep = [np.ascontiguousarray(f["tree_off"], np.int64), np.ascontiguousarray(f["left"], np.int32),
                np.ascontiguousarray(f["right"], np.int32), np.ascontiguousarray(f["feature"], np.int32),
                np.ascontiguousarray(f["threshold"], np.float64), np.ascontiguousarray(f["missing_left"], np.uint8),
                np.ascontiguousarray(f["value"], np.float64), np.ascontiguousarray(f["classes"], np.int64)]
        self._chk(self.lib.rsseg_forest_load(self.h, len(keep[0]) - 1, keep[0].ctypes.data_as(lp), keep[1].ctypes.data_as(ip),
                                             keep[2].ctypes.data_as(ip), keep[3].ctypes.data_as(ip), keep[4].ctypes.data_as(dp),
                                             keep[5].ctypes.data_as(C.POINTER(C.c_uint8)), keep[6].ctypes.data_as(dp),
                                             keep[6].shape[1], keep[7].ctypes.data_as(lp), int(f["n_features"])))

    def forest_predict(self, planes: Sequence):
        torch = _torch()
        n = planes[0].numel()
        out = self.empty(n, torch.int64)
        self._chk(self.lib.rsseg_forest_predict(self.h, self._pp(planes), len(planes), n, C.c_void_p(out.data_ptr())))
        return out


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    """Process-wide context on LOCAL_RANK (or device 0)."""
    global _default_ctx
    if _default_ctx is None:
        import os
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_ctx


def host_kmeans_draws(n: int, k: int, dtype, seed: int = 42) -> Tuple[int, np.ndarray]:
    """CPU-only: the k-means++ random draws as the library computes them (for tests)."""
    lib = L.load()
    Lt = 2 + int(np.log(k))
    u = np.zeros(max((k - 1) * Lt, 1), np.float64)
    cid = C.c_int64(0)
    rc = lib.rsseg_host_kmeans_draws(seed, n, L.F32 if np.dtype(dtype) == np.float32 else L.F64, k, C.byref(cid),
                                     u.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != 0:
        raise ValueError("rsseg_host_kmeans_draws failed")
    return cid.value, u[:(k - 1) * Lt]
