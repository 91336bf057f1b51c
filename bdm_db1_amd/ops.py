"""Thin torch-tensor -> C-ABI adapters.  Tensors are containers only: every function passes
``tensor.data_ptr()`` + sizes to libdb1_hip.so on the current HIP stream.  No arithmetic happens in
Python or in torch ops here, and nothing falls back to torch when a call fails."""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional

import torch

from . import lib
from .lib import DB1_BF16, DB1_F32, ACT_CODES

_vp = ctypes.c_void_p


def dt_code(t) -> int:
    dt = t if isinstance(t, torch.dtype) else t.dtype
    if dt == torch.float32:
        return DB1_F32
    if dt == torch.bfloat16:
        return DB1_BF16
    raise lib.Db1Error(f"unsupported dtype {dt} (float32 / bfloat16 only)")


def P(t: Optional[torch.Tensor]):
    if t is None:
        return _vp(0)
    if not t.is_cuda:
        raise lib.Db1Error("the DB1 kernels need device tensors (no CPU path)")
    return _vp(t.data_ptr())


_tls = threading.local()   # .scope = (raw handle, c_void_p, device index) of the stream a model call runs on, while it runs -- per host thread
_DEBUG_STREAM = bool(int(os.environ.get("DB1_DEBUG_STREAM", "0")))


def _scope():
    return getattr(_tls, "scope", None)


class stream_scope:
    """``with ops.stream_scope():`` around a model forward / backward: the current stream is looked up ONCE (torch.cuda.current_stream()
    costs ~7 us of host time and every launch asks for it: a quarter of the host time of an eager inference call).  Nothing inside may
    switch streams or devices -- the model's own code never does; ``DB1_DEBUG_STREAM=1`` checks every launch against torch's current
    stream.  The scope is thread-local (one host thread per GPU / rank is the contract of include/db1_hip.h; two threads driving two
    models never see each other's stream or scratch buffer)."""

    def __enter__(self):
        self.prev = _scope()
        h = torch.cuda.current_stream().cuda_stream
        _tls.scope = (h, _vp(h), torch.cuda.current_device())
        return self

    def __exit__(self, *exc):
        _tls.scope = self.prev
        return False


def stream():
    sc = _scope()
    if sc is not None:
        if _DEBUG_STREAM and (torch.cuda.current_device() != sc[2] or torch.cuda.current_stream().cuda_stream != sc[0]):
            raise lib.Db1Error("the current stream / device changed inside an ops.stream_scope()")
        return sc[1]
    return _vp(torch.cuda.current_stream().cuda_stream)


class _Workspace:
    """The scratch the C ABI asks its caller for (include/db1_hip.h: ``db1_<op>_workspace_bytes``): one torch buffer per (device,
    stream), grown on demand by the caching allocator (stream-ordered: no device synchronisation, no hipMalloc inside an entry
    point).  Every op that needs scratch is finished with it when its work on the stream is done, and ops of one stream run in
    order, so one buffer of the largest size requested so far serves them all.  Under hipGraph capture the capturing stream gets its
    own buffer from the graph's private pool (torch allows allocation there), which lives as long as this cache references it."""

    def __init__(self):
        self.bufs = {}

    def get(self, nbytes: int, device):
        if nbytes <= 0:
            return _vp(0), 0
        key = (device.index if device.index is not None else torch.cuda.current_device(),
               _scope()[0] if _scope() is not None else torch.cuda.current_stream(device).cuda_stream)
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
            self.bufs[key] = buf
        return _vp(buf.data_ptr()), buf.numel()


_workspace = _Workspace()
_ws_query_cache = {}


def reserve_workspace(nbytes: int, device=None):
    """pre-size the current stream's scratch buffer (so that a timed region never contains its first allocation)"""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    _workspace.get(int(nbytes), device)


def _ws(query: str, args: tuple, device):
    """(ws pointer, ws_bytes) for one call: size from the library's own query (memoised per shape)"""
    key = (query, args)
    n = _ws_query_cache.get(key)
    if n is None:
        n = int(getattr(lib.load(), query)(*args))
        _ws_query_cache[key] = n
    return _workspace.get(n, device)


def _strides2(t: torch.Tensor):
    assert t.dim() == 2
    return t.stride(0), t.stride(1)


class KernelTimer:
    """HIP-event timing of individual launches for bench.py's roofline legs: one event pair per launch, recorded on the stream the
    kernel is launched on; nothing is synchronised until ``summary()``.  ``work`` is the ALGORITHMIC work of the launch: FLOPs actually
    executed for the MFMA-bound families ("gemm", "flash_fwd", "flash_bwd"), bytes for the HBM-bound kernels."""

    def __init__(self, only=None):
        self.rec = {}   # family -> [event pairs, summed work, launches]
        self.only = None if only is None else set(only)   # time these families only (the others run uninstrumented)

    def wrap(self, family: str, work: float, fn):
        if self.only is not None and family not in self.only:
            fn()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        r = self.rec.setdefault(family, [[], 0.0, 0])
        r[0].append((e0, e1))
        r[1] += work
        r[2] += 1

    def summary(self):
        """{family: (total ms, work, launches)}"""
        torch.cuda.synchronize()
        return {k: (float(sum(a.elapsed_time(b) for a, b in v[0])), v[1], v[2]) for k, v in self.rec.items()}


_gemm_timer: Optional[KernelTimer] = None


def set_gemm_timer(t: Optional[KernelTimer]):
    global _gemm_timer
    _gemm_timer = t


def _timed(family: str, work: float, run):
    if _gemm_timer is not None:
        _gemm_timer.wrap(family, work, run)
    else:
        run()


def marker(tag: int):
    """an empty kernel named db1_marker_kernel on the current stream (include/db1_hip_test.h): cuts a kernel trace to a region"""
    lib.call("db1_test_marker", int(tag), stream())


def _tri_fraction(M: int, K: int, mode: int, period: int, tm: int = 256, tk: int = 64) -> float:
    """share of the k-tiles a tile GEMM executes under a structural-zero hint for A (db1_gemm_strided_tri): a k-tile is skipped when
    it is zero for EVERY row of the m-tile (mode 1: k > m; mode 2: (k mod period) < m)"""
    if mode == 0:
        return 1.0
    done = total = 0
    for m0 in range(0, M, tm):
        m1 = min(m0 + tm, M) - 1
        for k0 in range(0, K, tk):
            total += 1
            if mode == 1:
                done += 0 if k0 > m1 else 1
            else:
                done += 0 if (k0 % period) + tk - 1 < m0 else 1
    return done / max(total, 1)


def _is_tile_gemm(M, N, K, a, b, out, sa, sb, so) -> bool:
    return bool(lib.load().db1_gemm_would_use_fast(M, N, K, dt_code(a), dt_code(b), dt_code(out), sa[0], sa[1], sb[0], sb[1], so[0], so[1]))


def gemm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, bias: Optional[torch.Tensor] = None,
         alpha: float = 1.0, beta: float = 0.0, useful_flops: Optional[float] = None):
    """out[m,n] = alpha * a[m,k] @ b[k,n] + beta*out + bias[n]; a, b, out are 2-D VIEWS with any strides
    (pass ``w.t()`` for y = x W^T).  Optional leading batch dims are not handled here (see gemm_batched).
    ``useful_flops``: what the timer counts when part of the product is padding (the vocabulary pad of the tied head)."""
    M, K = a.shape
    K2, N = b.shape
    assert K == K2 and out.shape == (M, N), (a.shape, b.shape, out.shape)

    def run():
        ws, wsn = _ws("db1_gemm_workspace_bytes", (M, N, K, dt_code(a), dt_code(b), dt_code(out), a.stride(0), a.stride(1), b.stride(0), b.stride(1),
                                                    out.stride(0), out.stride(1), 1, 1), out.device)
        lib.call("db1_gemm_strided", P(a), P(b), P(out), P(bias), M, N, K, dt_code(a), dt_code(b), dt_code(out),
                 dt_code(bias) if bias is not None else 0, a.stride(0), a.stride(1), b.stride(0), b.stride(1),
                 out.stride(0), out.stride(1), 1, 1, 0, 0, 0, 0, 0, 0, alpha, beta, ws, wsn, stream())

    if _gemm_timer is not None and _is_tile_gemm(M, N, K, a, b, out, a.stride(), b.stride(), out.stride()):
        _gemm_timer.wrap("gemm", 2.0 * M * N * K if useful_flops is None else useful_flops, run)
    else:
        run()
    return out


def gemm_nt_headbias_supported(M, N, K, split_n) -> bool:
    return bool(lib.load().db1_gemm_nt_headbias_supported(M, N, K, split_n))


def gemm_nt_headbias(x, w, out, qu, qv, bias_u, bias_v, split_n):
    """out[:, split_n:] = (x w^T)[:, split_n:];  qu = (x w^T)[:, :split_n] + bias_u, qv = ... + bias_v (bf16; out[:, :split_n] is NOT written)"""
    M, K = x.shape
    N = w.shape[0]
    assert x.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1 and qu.is_contiguous() and qv.is_contiguous()
    assert x.dtype == w.dtype == out.dtype == qu.dtype == bias_u.dtype == torch.bfloat16 and bias_u.numel() == split_n == bias_v.numel()

    def run():
        lib.call("db1_gemm_nt_headbias", P(x), P(w), P(out), P(qu), P(qv), P(bias_u), P(bias_v), M, N, K, split_n, x.stride(0), w.stride(0),
                 out.stride(0), split_n, stream())

    _timed("gemm", 2.0 * M * N * K, run)


def gemm_batched(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, alpha: float = 1.0, beta: float = 0.0, tri=(0, 0)):
    """4-D views [z0, z1, rows, cols] with arbitrary strides (stride 0 broadcasts).
    ``tri`` = (mode, period): structural-zero hint for ``a`` (db1_gemm_strided_tri), an optimisation only."""
    Z0, Z1, M, K = a.shape
    _, _, K2, N = b.shape
    assert K == K2 and out.shape == (Z0, Z1, M, N) and b.shape[:2] == (Z0, Z1), (a.shape, b.shape, out.shape)

    def run():
        ws, wsn = _ws("db1_gemm_workspace_bytes", (M, N, K, dt_code(a), dt_code(b), dt_code(out), a.stride(2), a.stride(3), b.stride(2), b.stride(3),
                                                    out.stride(2), out.stride(3), Z0, Z1), out.device)
        lib.call("db1_gemm_strided_tri", P(a), P(b), P(out), _vp(0), M, N, K, dt_code(a), dt_code(b), dt_code(out), 0,
                 a.stride(2), a.stride(3), b.stride(2), b.stride(3), out.stride(2), out.stride(3), Z0, Z1,
                 a.stride(0), a.stride(1), b.stride(0), b.stride(1), out.stride(0), out.stride(1), alpha, beta, int(tri[0]), int(tri[1]),
                 ws, wsn, stream())

    # the batched contractions of the attention backward (dq_r, dR) run on the same tile kernels: timed with the FLOPs they execute
    # (k-tiles skipped under the structural-zero hint are not counted)
    if _gemm_timer is not None and _is_tile_gemm(M, N, K, a, b, out, a.stride()[2:], b.stride()[2:], out.stride()[2:]):
        _gemm_timer.wrap("gemm", 2.0 * M * N * K * Z0 * Z1 * _tri_fraction(M, K, int(tri[0]), int(tri[1])), run)
    else:
        run()
    return out


def relattn_dqr_supported(B, L, H, D, dtype) -> bool:
    return bool(lib.load().db1_relattn_dqr_supported(B, L, H, D, dt_code(dtype)))


def relattn_dqr(dT, R, dqv):
    """dqv[b, i, h, :] = sum_dist dT[h, b, i, dist] R[dist, h, :]  (dT [H,B,L,L] bf16 zero above the causal diagonal, R [L, H*D], dqv [B,L,H,D])"""
    H, B, L, _ = dT.shape
    D = dqv.shape[-1]
    assert dT.is_contiguous() and R.stride(1) == 1 and dqv.stride(3) == 1 and dqv.stride(2) == D
    ws, wsn = _ws("db1_relattn_dqr_workspace_bytes", (L, H), dT.device)
    # algorithmic bytes: the causal half of dT read once + the output written
    _timed("relattn_dqr", 2.0 * (H * B * L * (L + 1) / 2 + dqv.numel()),
           lambda: lib.call("db1_relattn_dqr", P(dT), P(R), R.stride(0), P(dqv), dqv.stride(1), dqv.stride(0), B, L, H, D, ws, wsn, stream()))


NO_DROP = (0.0, 0, 0, 0)   # (p, seed, site, step[, device step counter: the step used is step + *counter, read when the kernel runs[, rows per step]])


def _drop_dev(drop):
    return P(drop[4]) if len(drop) > 4 and drop[4] is not None else _vp(0)


def _drop_step(drop) -> int:
    return int(drop[3]) & 0xFFFFFFFF     # (a window backward passes first-step offsets below zero on top of a device counter: modulo 2^32 like the kernel's add)


def _drop_rps(drop) -> int:
    """rows per micro-step when the rows are a whole accumulation window's (db1_layernorm_residual_bwd: drop_rows_per_step), else 0"""
    return int(drop[5]) if len(drop) > 5 and drop[5] else 0


def dropout(x, y, drop):
    """y = dropout(x) with the counter-based keep decisions of ``drop = (p, seed, site, step)`` (y may alias x; also its own backward)"""
    p, seed, site, step = drop[:4]
    lib.call("db1_dropout", P(x), P(y), x.numel(), float(p), int(seed), int(site), _drop_step(drop), _drop_dev(drop), dt_code(x), stream())


def relattn_dqr_fused(dT, R, dq, du_acc, dv_acc):
    """dq[b,i,h,:] += sum_dist dT[h,b,i,dist] R[dist,h,:] (dq: a [B,L,H,D] view, e.g. the q slot of dqkv), du_acc += colsum of the incoming dq
    (the (q+u).k branch), dv_acc += colsum of the added term"""
    H, B, L, _ = dT.shape
    D = dq.shape[-1]
    assert dT.is_contiguous() and R.stride(1) == 1 and dq.stride(3) == 1 and dq.stride(2) == D and dq.shape == (B, L, H, D)
    assert du_acc.dtype == torch.float32 and dv_acc.dtype == torch.float32 and du_acc.numel() == H * D == dv_acc.numel()
    ws, wsn = _ws("db1_relattn_dqr_workspace_bytes", (L, H), dT.device)
    # algorithmic bytes: the causal half of dT read once, dq read and written
    _timed("relattn_dqr", 2.0 * (H * B * L * (L + 1) / 2 + 2 * dq.numel()),
           lambda: lib.call("db1_relattn_dqr_fused", P(dT), P(R), R.stride(0), P(dq), dq.stride(1), dq.stride(0), P(du_acc), P(dv_acc),
                            B, L, H, D, ws, wsn, stream()))


def relattn_dqr_groups_supported(B, L, H, D, dtype, ngroups) -> bool:
    return bool(lib.load().db1_relattn_dqr_groups_supported(B, L, H, D, dt_code(dtype), int(ngroups)))


def relattn_dqr_fused_groups(dT, Rg, dq, du_acc, dv_acc):
    """relattn_dqr_fused over a batch of ng = Rg.shape[0] blocks of B / ng sequences, block g with its own R = Rg[g] ([ng, L, H*D] view: a
    whole accumulation window's attention backward in one launch)"""
    H, B, L, _ = dT.shape
    D = dq.shape[-1]
    ng = Rg.shape[0]
    assert dT.is_contiguous() and Rg.dim() == 3 and Rg.stride(2) == 1 and dq.stride(3) == 1 and dq.stride(2) == D and dq.shape == (B, L, H, D)
    assert du_acc.dtype == torch.float32 and dv_acc.dtype == torch.float32 and du_acc.numel() == H * D == dv_acc.numel()
    ws, wsn = _ws("db1_relattn_dqr_groups_workspace_bytes", (L, H, ng), dT.device)
    _timed("relattn_dqr", 2.0 * (H * B * L * (L + 1) / 2 + 2 * dq.numel()),
           lambda: lib.call("db1_relattn_dqr_fused_groups", P(dT), P(Rg), Rg.stride(1), Rg.stride(0), ng, P(dq), dq.stride(1), dq.stride(0), P(du_acc), P(dv_acc),
                            B, L, H, D, ws, wsn, stream()))


def relattn_dqr_parts_rows(H: int) -> int:
    return int(lib.load().db1_relattn_dqr_parts_rows(int(H)))


def relattn_dqr_fused_parts(dT, R, dq, parts):
    """relattn_dqr_fused without its two reduces: parts [2, rows, H * D] float32 receives the per-workgroup column sums (du half, dv half)"""
    H, B, L, _ = dT.shape
    D = dq.shape[-1]
    assert dT.is_contiguous() and R.stride(1) == 1 and dq.stride(3) == 1 and dq.stride(2) == D and dq.shape == (B, L, H, D)
    assert parts.dtype == torch.float32 and parts.is_contiguous() and parts.shape == (2, relattn_dqr_parts_rows(H), H * D)
    ws, wsn = _ws("db1_relattn_dqr_workspace_bytes", (L, H), dT.device)
    _timed("relattn_dqr", 2.0 * (H * B * L * (L + 1) / 2 + 2 * dq.numel()),
           lambda: lib.call("db1_relattn_dqr_fused_parts", P(dT), P(R), R.stride(0), P(dq), dq.stride(1), dq.stride(0), P(parts),
                            B, L, H, D, ws, wsn, stream()))


def layernorm_residual_fwd(x, r, alpha, gamma, beta, y, s_out, mean, rstd, eps, drop=NO_DROP):
    rows, d = x.numel() // x.shape[-1], x.shape[-1]
    nstreams = 2 + (r is not None) + (s_out is not None)   # x [, r] in; y [, s] out
    _timed("layernorm_fwd", float(rows * d * x.element_size() * nstreams),
           lambda: lib.call("db1_layernorm_residual_fwd", P(x), P(r), float(alpha), P(gamma), P(beta), P(y), P(s_out), P(mean), P(rstd),
                            rows, d, float(eps), float(drop[0]), int(drop[1]), int(drop[2]), int(drop[3]), _drop_dev(drop), dt_code(x), dt_code(gamma), stream()))


def layernorm_residual_bwd(dy, s, gamma, mean, rstd, ds, dgamma_acc, dbeta_acc, dr_out=None, drop=NO_DROP):
    """``dr_out``: gradient of the (dropped) residual input r = ds under the forward's keep decisions; None when nothing was dropped"""
    rows, d = dy.numel() // dy.shape[-1], dy.shape[-1]
    ws, wsn = _ws("db1_layernorm_residual_bwd_workspace_bytes", (rows, d, dt_code(dy)), dy.device)
    _timed("layernorm_bwd", float(rows * d * dy.element_size() * (3 + (dr_out is not None))),   # dy, s in; ds [, dr] out
           lambda: lib.call("db1_layernorm_residual_bwd", P(dy), P(s), P(gamma), P(mean), P(rstd), P(ds), P(dr_out), P(dgamma_acc), P(dbeta_acc),
                            rows, d, float(drop[0]), int(drop[1]), int(drop[2]), _drop_step(drop), _drop_dev(drop), _drop_rps(drop), dt_code(dy), dt_code(gamma),
                            ws, wsn, stream()))


def layernorm_bwd_parts_numel(rows: int, d: int, dtype) -> int:
    """floats of the partial-sum matrix [blocks, 2 d] of db1_layernorm_residual_bwd_parts (0: this shape has no partials form)"""
    return int(lib.load().db1_layernorm_residual_bwd_workspace_bytes(int(rows), int(d), dt_code(dtype))) // 4


def layernorm_residual_bwd_parts(dy, s, gamma, mean, rstd, ds, parts, dr_out=None, drop=NO_DROP):
    """layernorm_residual_bwd without the parameter reduce: parts [blocks, 2 d] float32 receives the per-block (dgamma | dbeta) partial sums"""
    rows, d = dy.numel() // dy.shape[-1], dy.shape[-1]
    assert parts.dtype == torch.float32 and parts.is_contiguous()
    _timed("layernorm_bwd", float(rows * d * dy.element_size() * (3 + (dr_out is not None))),
           lambda: lib.call("db1_layernorm_residual_bwd_parts", P(dy), P(s), P(gamma), P(mean), P(rstd), P(ds), P(dr_out), P(parts), parts.numel() * 4,
                            rows, d, float(drop[0]), int(drop[1]), int(drop[2]), _drop_step(drop), _drop_dev(drop), _drop_rps(drop), dt_code(dy), dt_code(gamma),
                            stream()))


def ffn_act_fwd(z, out, act: str):
    rows, n = out.numel() // out.shape[-1], out.shape[-1]
    _timed("ffn_act_fwd", float((z.numel() + out.numel()) * z.element_size()),
           lambda: lib.call("db1_ffn_act_fwd", P(z), P(out), rows, n, ACT_CODES[act], dt_code(z), stream()))


def ffn_act_bwd(z, dout, dz, act: str):
    rows, n = dout.numel() // dout.shape[-1], dout.shape[-1]
    lib.call("db1_ffn_act_bwd", P(z), P(dout), P(dz), rows, n, ACT_CODES[act], dt_code(z), stream())


def ffn_act_bwd_bias(z, dout, dz, dbias_acc, act: str):
    """activation backward + dbias_acc += column sums of dz, one pass"""
    rows, n = dout.numel() // dout.shape[-1], dout.shape[-1]
    assert dbias_acc.dtype == torch.float32 and dbias_acc.numel() == dz.shape[-1]
    ws, wsn = _ws("db1_ffn_act_bwd_bias_workspace_bytes", (rows, n, ACT_CODES[act]), z.device)
    _timed("ffn_act_bwd", float((z.numel() + dout.numel() + dz.numel()) * z.element_size()),
           lambda: lib.call("db1_ffn_act_bwd_bias", P(z), P(dout), P(dz), P(dbias_acc), rows, n, ACT_CODES[act], dt_code(z), ws, wsn, stream()))


def gemm_nt_geglu_fused(M: int, dff: int, K: int, dtype, lda=None, ldw=None, ldz=None, ldact=None) -> bool:
    """does db1_gemm_nt_geglu run the activation inside the GEMM's epilogue at this shape (else: separate launches)"""
    return bool(lib.load().db1_gemm_nt_geglu_fused(M, dff, K, dt_code(dtype), lda or K, ldw or K, ldz or 2 * dff, ldact or dff))


def gemm_nn_geglu_bwd_fused(M: int, dff: int, K: int, dtype, lddy=None, ldw=None, ldz=None, lddz=None) -> bool:
    return bool(lib.load().db1_gemm_nn_geglu_bwd_fused(M, dff, K, dt_code(dtype), lddy or K, ldw or dff, ldz or 2 * dff, lddz or 2 * dff))


def gemm_nt_geglu(x, w1, bias, z, act):
    """z = x w1^T + bias  [M, 2 dff]   and   act = z[:, :dff] * gelu(z[:, dff:])  [M, dff]   (PositionwiseFF's first half, one launch where fused)"""
    M, K = x.shape
    dff = act.shape[1]
    assert w1.shape == (2 * dff, K) and z.shape == (M, 2 * dff) and x.stride(1) == 1 and w1.stride(1) == 1 and z.stride(1) == 1 and act.stride(1) == 1
    assert bias is None or bias.dtype == x.dtype
    ws, wsn = _ws("db1_gemm_workspace_bytes", (M, 2 * dff, K, dt_code(x), dt_code(w1), dt_code(z), x.stride(0), 1, 1, w1.stride(0), z.stride(0), 1, 1, 1), x.device)
    _timed("gemm", 2.0 * M * 2 * dff * K,
           lambda: lib.call("db1_gemm_nt_geglu", P(x), P(w1), P(bias), P(z), P(act), M, dff, K, x.stride(0), w1.stride(0), z.stride(0), act.stride(0),
                            dt_code(x), ws, wsn, stream()))


def gemm_nn_geglu_bwd(dy, w2, z, dz, dbias_acc):
    """dz = GEGLU'(z) applied to dact = dy w2 (w2 [K, dff] row-major; dact is never stored); dbias_acc [2 dff] fp32 += column sums of dz"""
    M, K = dy.shape
    dff = w2.shape[1]
    assert w2.shape[0] == K and z.shape == (M, 2 * dff) and dz.shape == (M, 2 * dff) and dbias_acc.dtype == torch.float32 and dbias_acc.numel() == 2 * dff
    assert dy.stride(1) == 1 and w2.stride(1) == 1 and z.stride(1) == 1 and dz.stride(1) == 1
    ws, wsn = _ws("db1_gemm_nn_geglu_bwd_workspace_bytes", (M, dff, K, dt_code(dy), dy.stride(0), w2.stride(0), z.stride(0), dz.stride(0)), dy.device)
    _timed("gemm", 2.0 * M * dff * K,
           lambda: lib.call("db1_gemm_nn_geglu_bwd", P(dy), P(w2), P(z), P(dz), P(dbias_acc), M, dff, K, dy.stride(0), w2.stride(0), z.stride(0), dz.stride(0),
                            dt_code(dy), ws, wsn, stream()))


def gemm_nn_geglu_bwd_parts(dy, w2, z, dz, parts):
    """gemm_nn_geglu_bwd without the bias reduce (fused shapes): parts [M / 128, 2 dff] float32 receives the column sums of dz per 128-row block"""
    M, K = dy.shape
    dff = w2.shape[1]
    assert w2.shape[0] == K and z.shape == (M, 2 * dff) and dz.shape == (M, 2 * dff) and parts.dtype == torch.float32 and parts.shape == (M // 128, 2 * dff)
    assert dy.stride(1) == 1 and w2.stride(1) == 1 and z.stride(1) == 1 and dz.stride(1) == 1 and parts.is_contiguous()
    _timed("gemm", 2.0 * M * dff * K,
           lambda: lib.call("db1_gemm_nn_geglu_bwd_parts", P(dy), P(w2), P(z), P(dz), P(parts), M, dff, K, dy.stride(0), w2.stride(0), z.stride(0), dz.stride(0),
                            dt_code(dy), stream()))


def colsum_acc(x2d, out_acc):
    rows, cols = x2d.shape
    assert x2d.stride(1) == 1 and out_acc.dtype == torch.float32
    ws, wsn = _ws("db1_colsum_acc_workspace_bytes", (rows, cols), x2d.device)
    lib.call("db1_colsum_acc", P(x2d), P(out_acc), rows, cols, x2d.stride(0), dt_code(x2d), ws, wsn, stream())


def add(a, b, y):
    lib.call("db1_add", P(a), P(b), P(y), a.numel(), dt_code(a), stream())


def add2d(a2d, b2d, y2d):
    """y = a + b on 2-D views with unit inner stride (y may alias b)."""
    rows, cols = y2d.shape
    assert a2d.stride(1) == 1 and b2d.stride(1) == 1 and y2d.stride(1) == 1 and b2d.dtype == y2d.dtype
    lib.call("db1_add2d", P(a2d), a2d.stride(0), P(b2d), b2d.stride(0), P(y2d), y2d.stride(0), rows, cols,
             dt_code(a2d), dt_code(y2d), stream())


def add2d_colsums(a2d, b2d, y2d, sum_a_acc, sum_b_acc):
    """y = a + b (y may alias b) and the float32 column sums of a and of b accumulated into sum_a_acc / sum_b_acc, one pass."""
    rows, cols = y2d.shape
    assert a2d.stride(1) == 1 and b2d.stride(1) == 1 and y2d.stride(1) == 1 and a2d.dtype == y2d.dtype == b2d.dtype
    assert sum_a_acc.dtype == torch.float32 and sum_b_acc.dtype == torch.float32
    ws, wsn = _ws("db1_add2d_colsums_workspace_bytes", (rows, cols), y2d.device)
    _timed("add2d_colsums", float(3 * rows * cols * y2d.element_size()),
           lambda: lib.call("db1_add2d_colsums", P(a2d), a2d.stride(0), P(b2d), b2d.stride(0), P(y2d), y2d.stride(0), P(sum_a_acc), P(sum_b_acc),
                            rows, cols, dt_code(y2d), ws, wsn, stream()))


def zero_segments(base, segments):
    """base[off : off + len] = 0 for every row (off, len) of the int64 device table ``segments`` [n, 2]; one launch"""
    assert base.dtype == torch.float32 and segments.dtype == torch.int64 and segments.dim() == 2 and segments.shape[1] == 2 and segments.is_contiguous()
    lib.call("db1_zero_segments", P(base), P(segments), segments.shape[0], stream())


def cast(x, y):
    lib.call("db1_cast", P(x), P(y), x.numel(), dt_code(x), dt_code(y), stream())


def embed_gather(table, ids, out2d):
    n, d = out2d.shape
    assert ids.dtype == torch.int64 and ids.numel() == n and out2d.stride(1) == 1
    lib.call("db1_embed_gather_fwd", P(table), P(ids), P(out2d), n, d, out2d.stride(0), table.shape[0], dt_code(table), dt_code(out2d), stream())


def vision_pos_add(out2d, row_table, col_table, row_ids, col_ids):
    """out[t] += row_table[row_ids[t]] + col_table[col_ids[t]] (db1_vision_pos_add)"""
    n, d = out2d.shape
    assert out2d.is_contiguous() and row_table.shape == col_table.shape and row_table.shape[1] == d and row_table.dtype == col_table.dtype
    assert row_ids.dtype == torch.int64 and col_ids.dtype == torch.int64 and row_ids.numel() == n == col_ids.numel()
    lib.call("db1_vision_pos_add", P(out2d), P(row_table), P(col_table), P(row_ids), P(col_ids), n, d, row_table.shape[0], dt_code(row_table), dt_code(out2d), stream())


def embed_scatter_add(dout2d, ids, dtable_acc):
    n, d = dout2d.shape
    assert ids.dtype == torch.int64 and dtable_acc.dtype == torch.float32 and dout2d.stride(1) == 1
    ws, wsn = _ws("db1_embed_scatter_add_workspace_bytes", (n,), dout2d.device)
    lib.call("db1_embed_scatter_add_bwd", P(dout2d), P(ids), P(dtable_acc), n, d, dout2d.stride(0), dtable_acc.shape[0], dt_code(dout2d), ws, wsn, stream())


def rl_assemble_fwd(word_table, pos_table, vis, ids, position_id, labels, out):
    B, L, d = out.shape
    nvis = 0 if vis is None else vis.shape[1]
    lib.call("db1_rl_assemble_fwd", P(word_table), P(pos_table), P(vis), P(ids), P(position_id), P(labels), P(out),
             B, L, d, nvis, word_table.shape[0], pos_table.shape[0], dt_code(word_table), dt_code(out), stream())


def rl_assemble_bwd(dout, ids, position_id, dword_acc, dpos_acc, dvis):
    B, L, d = dout.shape
    nvis = 0 if dvis is None else dvis.shape[1]
    ws, wsn = _ws("db1_rl_assemble_bwd_workspace_bytes", (B, L), dout.device)
    lib.call("db1_rl_assemble_bwd", P(dout), P(ids), P(position_id), P(dword_acc), P(dpos_acc), P(dvis), B, L, d, nvis,
             dword_acc.shape[0], dpos_acc.shape[0], dt_code(dout), ws, wsn, stream())


def masked_ce_fwd(logits2d, labels, mask, lse, sums, V):
    T, ld = logits2d.shape[0], logits2d.stride(0)
    ws, wsn = _ws("db1_masked_ce_fwd_workspace_bytes", (T,), logits2d.device)
    _timed("masked_ce_fwd", float(T * V * logits2d.element_size()),
           lambda: lib.call("db1_masked_ce_fwd", P(logits2d), P(labels), P(mask), P(lse), P(sums), T, V, ld, dt_code(logits2d), ws, wsn, stream()))


def masked_ce_bwd(logits2d, labels, mask, lse, sums, dlogits2d, V, gscale=1.0):
    T, ld = logits2d.shape[0], logits2d.stride(0)
    assert dlogits2d.stride(0) == ld
    _timed("masked_ce_bwd", float(2 * T * V * logits2d.element_size()),
           lambda: lib.call("db1_masked_ce_bwd", P(logits2d), P(labels), P(mask), P(lse), P(sums), P(dlogits2d), T, V, ld, float(gscale),
                            dt_code(logits2d), stream()))


def masked_ce_fwd_bwd(logits2d, labels, mask, lse, sums, norm, V, gscale=1.0):
    """lse, sums += (loss, mask) and logits2d overwritten by dlogits in ONE pass (db1_masked_ce_fwd_bwd); norm[1] = the loss normaliser"""
    T, ld = logits2d.shape[0], logits2d.stride(0)
    ws, wsn = _ws("db1_masked_ce_fwd_workspace_bytes", (T,), logits2d.device)
    lib.call("db1_masked_ce_fwd_bwd", P(logits2d), P(labels), P(mask), P(lse), P(sums), P(norm), T, V, ld, float(gscale), dt_code(logits2d), ws, wsn, stream())


def lmhead_ce(h2d, W, labels, mask, lse, sums, V, dh=None, dW_acc=None, beta_dw=1.0, gscale=1.0, chunk_rows=0):
    """tied head + masked CE without the logits tensor (db1_lmhead_ce_fwd / _fwd_bwd): ``dh`` and ``dW_acc`` given -> the training sweep"""
    T, d = h2d.shape
    rows = W.shape[0]
    assert h2d.is_contiguous() and W.is_contiguous() and W.shape[1] == d and h2d.dtype == W.dtype and rows >= V
    train = dh is not None
    assert (dW_acc is not None) == train and (not train or (dh.is_contiguous() and dW_acc.dtype == torch.float32 and dW_acc.shape == (rows, d)))
    ws, wsn = _ws("db1_lmhead_ce_workspace_bytes", (T, rows, d, int(chunk_rows), dt_code(h2d), int(train)), h2d.device)
    flops = 2.0 * T * V * d * (3 if train else 1)

    def run():
        if train:
            lib.call("db1_lmhead_ce_fwd_bwd", P(h2d), P(W), P(labels), P(mask), P(lse), P(sums), P(dh), P(dW_acc), float(beta_dw), float(gscale),
                     T, V, rows, d, int(chunk_rows), dt_code(h2d), ws, wsn, stream())
        else:
            lib.call("db1_lmhead_ce_fwd", P(h2d), P(W), P(labels), P(mask), P(lse), P(sums), T, V, rows, d, int(chunk_rows), dt_code(h2d), ws, wsn, stream())

    _timed("lmhead_ce", flops, run)   # (GEMM-dominated: reported as its own MFMA-bound family by bench.py)


def relattn_add_head_bias(qkv, u, vb, qu, qv, B, Lq, Lk, H, D):
    lib.call("db1_relattn_add_head_bias", P(qkv), P(u), P(vb), P(qu), P(qv), B, Lq, Lk, H, D, dt_code(qkv), dt_code(u), stream())


def relattn_softmax_fwd(AC, T, lse, H, B, Lq, Lk, nd, mlen, shift, scale):
    lib.call("db1_relattn_softmax_fwd", P(AC), P(T), P(lse), H, B, Lq, Lk, nd, mlen, shift, float(scale), stream())


def relattn_softmax_bwd(Pm, dP, dT, H, B, Lq, Lk, nd, mlen, shift, scale):
    lib.call("db1_relattn_softmax_bwd", P(Pm), P(dP), P(dT), H, B, Lq, Lk, nd, mlen, shift, float(scale), stream())


def relattn_decode_supported(B, q, klen, H, D, dtype) -> bool:
    return bool(lib.load().db1_relattn_decode_supported(B, q, klen, H, D, dt_code(dtype)))


def relattn_decode_fwd(qu, qv, k, v, R, out, B, q, klen, mlen, H, D, shift, scale):
    """k, v: views [B, klen, H, D] (any row / batch stride, unit stride inside a head row)"""
    assert k.stride(3) == 1 and k.stride(2) == D and v.stride() == k.stride()
    ws, wsn = _ws("db1_relattn_decode_workspace_bytes", (B, q, klen, H), out.device)
    lib.call("db1_relattn_decode_fwd", P(qu), P(qv), P(k), P(v), k.stride(1), k.stride(0), P(R), R.shape[0], P(out),
             B, q, klen, mlen, H, D, shift, float(scale), ws, wsn, stream())


_tickets = {}


def decode_tickets(device) -> torch.Tensor:
    """the ticket counters of the fused inference launches (db1_linear_decode_tickets_bytes): zero once, every launch leaves them zero.
    One buffer per device: the inference path of a device runs on one stream at a time."""
    t = _tickets.get(device)
    if t is None:
        t = torch.zeros(int(lib.load().db1_linear_decode_tickets_bytes()) // 4, dtype=torch.int32, device=device)
        _tickets[device] = t
    return t


def relattn_decode_ring_fwd(qkv_new, u, vb, kv_ring, state, R, out, B, q, mlen, H, D, shift, scale, fused_merge=False, part=None):
    """attention of q new tokens over a ring of cached K / V (db1_relattn_decode_ring_fwd): kv_ring [B, cap, 2, H, D], state int32[1].
    out None + part (a float tensor of relattn_decode_ring_part_numel elements): the per-chunk partial results only (for linear_decode_attn).
    fused_merge: the chunk that finishes last merges inside the launch (ticket hand-off) instead of a second launch -- measured SLOWER inside
    a graph (q = 1: 13.1 vs 9.3 us, q = 22: 23.9 vs 13.2 us: the drain + ticket + dependent reads cost more than a launch boundary), so off"""
    cap = kv_ring.shape[1]
    if part is not None:
        ws, wsn = P(part), part.numel() * 4
    else:
        ws, wsn = _ws("db1_relattn_decode_ring_workspace_bytes", (B, q, mlen + q, H), qkv_new.device)
    lib.call("db1_relattn_decode_ring_fwd", P(qkv_new), P(u), P(vb), P(kv_ring), P(state), cap, P(R), R.shape[0], P(out), B, q, mlen, H, D, shift,
             float(scale), ws, wsn, P(decode_tickets(qkv_new.device)) if (fused_merge and out is not None) else _vp(0), stream())


def relattn_decode_ring_part_numel(B, q, klen, H) -> int:
    return int(lib.load().db1_relattn_decode_ring_workspace_bytes(int(B), int(q), int(klen), int(H))) // 4


def linear_decode_attn_supported(B, q, H, D, klen, N) -> bool:
    return bool(lib.load().db1_linear_decode_attn_supported(int(B), int(q), int(H), int(D), (int(klen) + 127) // 128, int(N)))


def linear_decode_attn(part, klen, B, q, H, D, W, y):
    """y = merge(part) W^T (db1_linear_decode_attn): part from relattn_decode_ring_fwd(out=None, part=...)"""
    assert W.dtype == torch.bfloat16 and W.is_contiguous() and y.dtype == torch.bfloat16 and y.stride(1) == 1 and W.shape == (y.shape[1], H * D)
    lib.call("db1_linear_decode_attn", P(part), (int(klen) + 127) // 128, B, q, H, D, P(W), P(y), y.stride(0), y.shape[1], stream())


_chain_scratch = {}


def decode_chain_supported(d, dff, H, D, klen) -> bool:
    return bool(lib.load().db1_decode_chain_supported(int(d), int(dff), int(H), int(D), (int(klen) + 127) // 128))


def _chain_key(device):
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()   # "cuda" and "cuda:0" are one device
    sc = _scope()
    return idx, (sc[0] if sc is not None else torch.cuda.current_stream(idx).cuda_stream)


class chain_scratch_scope:
    """``with ops.chain_scratch_scope(scratch, pinned_word):`` -- the chain launches and flag fetches inside use THIS scratch / pinned word
    instead of the (device, stream) ones.  For hipGraph captures (GraphedRingStep): a capture runs on torch's shared capture stream, whose
    key would be new -- its scratch would be allocated AND zero-filled inside the capture, i.e. every replay would start by wiping the
    scratch including the sticky error flag, and every graph captured on that stream would share one scratch whatever stream replays it.
    The owner creates both eagerly (``new_chain_scratch``) before it captures.  Thread-local."""

    def __init__(self, scratch, host_word):
        self.pair = (scratch, host_word)

    def __enter__(self):
        self.prev = getattr(_tls, "chain_override", None)
        _tls.chain_override = self.pair
        return self

    def __exit__(self, *exc):
        _tls.chain_override = self.prev
        return False


def _chain_pinned_word():
    global _chain_flag_pool
    if _chain_flag_pool is None:
        if torch.cuda.is_current_stream_capturing():
            raise lib.Db1Error("db1_decode_chain: the first one-token call must run eagerly (GraphedRingStep warms up before it captures)")
        _chain_flag_pool = [torch.zeros(256, dtype=torch.int32).pin_memory(), 0]
    pool, used = _chain_flag_pool
    if used >= pool.numel():
        raise lib.Db1Error("db1_decode_chain: more than 256 scratch owners (device x stream pairs + captured steps) decode through the persistent launch")
    _chain_flag_pool[1] = used + 1
    return pool[used:used + 1]


def new_chain_scratch(device):
    """(zeroed scratch, pinned host word) owned by the caller -- a captured step; must be called eagerly (never under capture)"""
    if torch.cuda.is_current_stream_capturing():
        raise lib.Db1Error("new_chain_scratch under hipGraph capture: the scratch of a captured step is created before the capture")
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return torch.zeros(int(lib.load().db1_decode_chain_scratch_bytes()), dtype=torch.uint8, device=torch.device("cuda", idx)), _chain_pinned_word()


def decode_chain_scratch(device):
    """the scratch of db1_decode_chain (tagged hand-off rows + error flag): one buffer per (device, STREAM) like every other scratch
    buffer here -- the tag of a word is only the layer index, so two models decoding at the same time on two streams must not see each
    other's rows (or each other's sticky error flag) -- or the one a chain_scratch_scope supplies.  Zeroed once, when it is created:
    never inside a capture (a zero-fill node would wipe the flag at every replay)."""
    ov = getattr(_tls, "chain_override", None)
    if ov is not None:
        return ov[0]
    key = _chain_key(device)
    t = _chain_scratch.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise lib.Db1Error("db1_decode_chain under hipGraph capture without a chain_scratch_scope: the scratch (and its error flag) would be "
                               "allocated and zero-filled inside the graph")
        t = torch.zeros(int(lib.load().db1_decode_chain_scratch_bytes()), dtype=torch.uint8, device=torch.device("cuda", key[0]))
        _chain_scratch[key] = t
    return t


_chain_flag_host = {}
_chain_flag_pool = None


def decode_chain_flag_fetch(device):
    """enqueue a copy of the chain's error flag into pinned host memory behind the launches issued so far (4 bytes, no synchronisation,
    legal under hipGraph capture) -> (pinned int32 tensor, the scratch it watches): the pinned word holds the flag once the stream has
    passed this point"""
    ov = getattr(_tls, "chain_override", None)
    if ov is not None:
        h = ov[1]
    else:
        key = _chain_key(device)
        h = _chain_flag_host.get(key)
        if h is None:
            # one pinned word per (device, stream), handed out from a small pool that is allocated on the first EAGER call: a stream that is
            # capturing a graph may not allocate pinned memory (hipHostMalloc invalidates the capture)
            h = _chain_pinned_word()
            _chain_flag_host[key] = h
    sc = decode_chain_scratch(device)
    off = int(lib.load().db1_decode_chain_error_offset())
    h.copy_(sc[off:off + 4].view(torch.int32), non_blocking=True)
    return h, sc


def decode_chain_clear_error(watch):
    h, sc = watch
    off = int(lib.load().db1_decode_chain_error_offset())
    sc[off:off + 4].zero_()
    h.zero_()


def decode_chain(part, klen, H, x_res, w_o, w1, b1, w2, b2, w_qkv_next, g1, be1, g2, be2, alpha, eps, h1_out, f_out, x_next, qkv_next, slot, w_o_next=None):
    """one new token through o_net (merge of the attention partials) -> LN -> ff1 + GEGLU -> ff2 -> LN -> the next layer's qkv projection, one
    launch (db1_decode_chain); w_qkv_next None = last layer (h1_out and f_out come back for the head's input LayerNorm).  Consecutive launches
    need different ``slot`` values (the layer index)."""
    d, dff = w_o.shape[0], w2.shape[1]
    for t in (x_res, w_o, w1, b1, w2, b2, g1, be1, g2, be2):
        assert t.dtype == torch.bfloat16 and t.is_contiguous()
    lib.call("db1_decode_chain", P(part), (int(klen) + 127) // 128, int(H), P(x_res), P(w_o), P(w1), P(b1), P(w2), P(b2), P(w_qkv_next), P(w_o_next), P(g1), P(be1), P(g2), P(be2),
             float(alpha), float(eps), P(h1_out), P(f_out), P(x_next), P(qkv_next), P(decode_chain_scratch(x_res.device)), int(slot), int(d), int(dff), stream())


def decode_chain_error(device) -> bool:
    """did a hand-off poll of a chain launch run into its limit (results invalid)?  Synchronises."""
    off = int(lib.load().db1_decode_chain_error_offset())
    return bool(decode_chain_scratch(device)[off:off + 4].view(torch.int32).item())


def linear_decode_supported(M, N, K, geglu=False, ln=False, pre=False) -> bool:
    return bool(lib.load().db1_linear_decode_supported(int(M), int(N), int(K), int(geglu), int(bool(ln)) | (2 if pre else 0)))


def linear_decode(x, W, bias, y, geglu=False, ln=None, pre=None):
    """y = x W^T + bias for M <= 64 rows (bf16), optionally through GEGLU (W [2N, K]); inside the same launch (db1_linear_decode)
    pre = (res, alpha, gamma, beta, eps, out): the input rows are x_eff = LayerNorm(alpha * res + x) * gamma + beta, also stored to out;
    ln  = (res, alpha, gamma, beta, eps, out): out = LayerNorm(alpha * res + y) * gamma + beta"""
    M, K = x.shape
    N = y.shape[1]
    assert x.dtype == torch.bfloat16 and W.dtype == torch.bfloat16 and y.dtype == torch.bfloat16 and W.is_contiguous() and W.shape == ((2 * N if geglu else N), K)
    assert x.stride(1) == 1 and y.stride(1) == 1
    ws, wsn = _ws("db1_linear_decode_workspace_bytes", (M, N, K, int(geglu)), y.device)
    dtp = 0

    def pack(t):
        nonlocal dtp
        if t is None:
            return (_vp(0), 0, 1.0, _vp(0), _vp(0), 0.0, _vp(0), 0)
        res, alpha, gamma, beta, eps, out = t
        assert res.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and gamma.dtype == beta.dtype and res.stride(1) == 1 and out.stride(1) == 1
        assert dtp in (0, dt_code(gamma.dtype) + 100)
        dtp = dt_code(gamma.dtype) + 100
        return (P(res), res.stride(0), float(alpha), P(gamma), P(beta), float(eps), P(out), out.stride(0))
    a_pre, a_ln = pack(pre), pack(ln)
    lib.call("db1_linear_decode", P(x), x.stride(0), P(W), P(bias), dt_code(bias.dtype) if bias is not None else 0, P(y), y.stride(0), M, N, K, int(geglu),
             *a_pre, *a_ln, max(dtp - 100, 0), P(decode_tickets(y.device)), ws, wsn, stream())


def ring_advance(state, q, cap):
    lib.call("db1_ring_advance", P(state), int(q), int(cap), stream())


def relattn_flash_supported(B, L, H, D, dtype) -> bool:
    return bool(lib.load().db1_relattn_flash_supported(B, L, H, D, dt_code(dtype)))


def relattn_flash_probs_tiles(L: int) -> int:
    """1 KiB fragment images per (batch, head) of the kept probabilities: the tiles (key block of 32, 16-query tile) on or below the causal
    diagonal, stored as a triangle (db1_relattn_flash_probs_bytes)"""
    return int(lib.load().db1_relattn_flash_probs_bytes(1, int(L), 1)) // 1024


def relattn_flash_probs_full(probs, L: int):
    """(tests / tools) the triangle of kept-probability images [B*H, tiles, 512] expanded to [B*H, L/32, L/16, 512]; tiles above the causal
    diagonal, which do not exist, read NaN"""
    nkb, nt = L // 32, L // 16
    full = torch.full((probs.shape[0], nkb, nt, 512), float("nan"), device=probs.device, dtype=probs.dtype)
    for jb in range(nkb):
        i0 = 2 * jb + jb * (nt - 1 - jb)
        full[:, jb, 2 * jb:] = probs[:, i0:i0 + nt - 2 * jb]
    return full


def relattn_flash_fwd(qu, qv, qkv5, R, out, lse, B, L, H, D, shift, scale, probs=None, mblk=None):
    """qkv5: the packed activations viewed [B, L, 3, H, D]; k / v are addressed inside it by stride.
    probs [B*H, relattn_flash_probs_tiles(L), 512] bf16 + mblk [B*H, L/32, L] f32 (optional): the unnormalised probabilities (a triangle of
    fragment images, see the header) and their reference maxima, kept for the stored-probabilities backward."""
    assert probs is None or probs.numel() * 2 == int(lib.load().db1_relattn_flash_probs_bytes(B, L, H)), "probs: db1_relattn_flash_probs_bytes(B, L, H) bytes"
    k, v = qkv5[:, :, 1], qkv5[:, :, 2]
    vis = L * (L + 1) / 2 if shift >= L else (shift * (shift + 1) / 2 + (L - shift) * shift)   # visible (query, key) pairs
    _timed("flash_fwd", 3 * 2.0 * B * H * vis * D,   # (q+u).k, (q+v).R, P.v over the visible pairs
           lambda: lib.call("db1_relattn_flash_fwd", P(qu), P(qv), P(k), P(v), k.stride(1), k.stride(0), P(R), P(out), P(lse),
                            B, L, H, D, shift, float(scale), P(probs) if probs is not None else _vp(0), P(mblk) if mblk is not None else _vp(0), stream()))


def relattn_flash_bwd(qu, qv, qkv5, R, out, dout, lse, delta, dqkv5, dT, B, L, H, D, shift, scale, store_probs=True, probs=None, mblk=None):
    """probs + mblk (from relattn_flash_fwd): nothing is recomputed.  Otherwise store_probs: give the library the scratch for P and dS
    (the key side then runs without recomputation); False = both sides recompute (no scratch)."""
    k, v = qkv5[:, :, 1], qkv5[:, :, 2]
    if probs is not None:
        ws, wsn = _ws("db1_relattn_flash_bwd_workspace_bytes", (B, L, H, 1), out.device)
    else:
        ws, wsn = _ws("db1_relattn_flash_bwd_workspace_bytes", (B, L, H, 0), out.device) if store_probs else (_vp(0), 0)
    dq, dk, dv = dqkv5[:, :, 0], dqkv5[:, :, 1], dqkv5[:, :, 2]
    vis = L * (L + 1) / 2 if shift >= L else (shift * (shift + 1) / 2 + (L - shift) * shift)
    _timed("flash_bwd", 6 * 2.0 * B * H * vis * D,   # twice the forward's algorithmic work (bwd_q + bwd_kv; dq_r / dR run as separate kernels)
           lambda: lib.call("db1_relattn_flash_bwd", P(qu), P(qv), P(k), P(v), k.stride(1), k.stride(0), P(R), P(out), P(dout), P(lse), P(delta),
                            P(dq), P(dk), P(dv), dq.stride(1), dq.stride(0), P(dT), B, L, H, D, shift, float(scale),
                            P(probs) if probs is not None else _vp(0), P(mblk) if mblk is not None else _vp(0), ws, wsn, stream()))


def patch_normalize(pixels, patches, p):
    n, C, Hh, Ww = pixels.shape
    lib.call("db1_patch_normalize", P(pixels), P(patches), n, C, Hh, Ww, p, dt_code(pixels), dt_code(patches), stream())


def im2col3x3(x, cols, N, C, p):
    lib.call("db1_im2col3x3", P(x), P(cols), N, C, p, cols.shape[-1], dt_code(x), stream())


def col2im3x3(dcols, dx, N, C, p):
    lib.call("db1_col2im3x3", P(dcols), P(dx), N, C, p, dcols.shape[-1], dt_code(dx), stream())


def nhwc_to_nchw(x, y, N, C, hw):
    lib.call("db1_nhwc_to_nchw", P(x), P(y), N, C, hw, dt_code(x), stream())


def nchw_to_nhwc(x, y, N, C, hw):
    lib.call("db1_nchw_to_nhwc", P(x), P(y), N, C, hw, dt_code(x), stream())


def groupnorm_gelu_fwd(x, gamma, beta, y, mean, rstd, N, C, hw, groups=32, eps=1e-5):
    lib.call("db1_groupnorm_gelu_fwd", P(x), P(gamma), P(beta), P(y), P(mean), P(rstd), N, C, hw, groups, float(eps),
             dt_code(x), dt_code(gamma), stream())


def groupnorm_gelu_bwd(dy, x, gamma, beta, mean, rstd, dx, dgamma_acc, dbeta_acc, N, C, hw, groups=32):
    lib.call("db1_groupnorm_gelu_bwd", P(dy), P(x), P(gamma), P(beta), P(mean), P(rstd), P(dx), P(dgamma_acc), P(dbeta_acc),
             N, C, hw, groups, dt_code(x), dt_code(gamma), stream())


# ---- channels-last vision pipeline (bf16 path)
def patch_normalize_nhwc(pixels, patches, p):
    n, C, Hh, Ww = pixels.shape
    lib.call("db1_patch_normalize_nhwc", P(pixels), P(patches), n, C, Hh, Ww, p, dt_code(pixels), dt_code(patches), stream())


def im2col3x3_nhwc(x, cols, N, C, p):
    lib.call("db1_im2col3x3_nhwc", P(x), P(cols), N, C, p, cols.shape[-1], dt_code(x), stream())


def col2im3x3_nhwc(dcols, dx, N, C, p):
    lib.call("db1_col2im3x3_nhwc", P(dcols), P(dx), N, C, p, dcols.shape[-1], dt_code(dx), stream())


def conv_weight_permute(w, wp, Cout, Cin):
    lib.call("db1_conv_weight_permute", P(w), P(wp), Cout, Cin, wp.shape[-1], dt_code(w), dt_code(wp), stream())


def conv_wgrad_unpermute(gp, g_acc, Cout, Cin):
    assert gp.dtype == torch.float32 and g_acc.dtype == torch.float32
    lib.call("db1_conv_wgrad_unpermute", P(gp), P(g_acc), Cout, Cin, gp.shape[-1], stream())


def conv_weight_permute_t(w, wp, Cout, Cin):
    lib.call("db1_conv_weight_permute_t", P(w), P(wp), Cout, Cin, dt_code(w), dt_code(wp), stream())


def conv3x3_implicit_fwd(x, w_op, bias, y, n_patches, sign=1, res=None):
    """64 -> 64 channel 3x3 conv on 16x16 patches, channels-last bf16, no column matrix (sign=-1: data gradient)"""
    if res is not None:   # y = conv + bias + res
        assert res.dtype == torch.bfloat16 and res.is_contiguous()
        lib.call("db1_conv3x3_implicit_fwd_res", P(x), P(w_op), P(bias), P(res), P(y), n_patches, sign, dt_code(bias) if bias is not None else 0, stream())
        return
    lib.call("db1_conv3x3_implicit_fwd", P(x), P(w_op), P(bias), P(y), n_patches, sign, dt_code(bias) if bias is not None else 0, stream())


def conv1_fused_fwd(x_cl, w_op, bias, cols, y, n_patches):
    """3 -> 64 channel 3x3 conv on 16x16 patches + its column matrix in one kernel (db1_conv1_fused_fwd)"""
    assert x_cl.dtype == torch.bfloat16 and w_op.shape == (64, 32) and cols.shape[1] == 32 and y.shape[1] == 64
    lib.call("db1_conv1_fused_fwd", P(x_cl), P(w_op), P(bias), P(cols), P(y), n_patches, dt_code(bias) if bias is not None else 0, stream())


def conv3x3_implicit_wgrad(dy, x, gp_acc, n_patches, gbias_acc=None):
    assert gp_acc.dtype == torch.float32 and gp_acc.shape[-1] == 576
    ws, wsn = _ws("db1_conv3x3_implicit_wgrad_workspace_bytes", (int(n_patches),), dy.device)   # fixed-order partial sums: bit-reproducible
    lib.call("db1_conv3x3_implicit_wgrad", P(dy), P(x), P(gp_acc), P(gbias_acc), n_patches, ws, wsn, stream())


def groupnorm_gelu_nhwc_fwd(x, gamma, beta, y, mean, rstd, N, C, hw, groups=32, eps=1e-5):
    lib.call("db1_groupnorm_gelu_nhwc_fwd", P(x), P(gamma), P(beta), P(y), P(mean), P(rstd), N, C, hw, groups, float(eps),
             dt_code(x), dt_code(gamma), stream())


def groupnorm_gelu_nhwc_bwd(dy, x, gamma, beta, mean, rstd, dx, dgamma_acc, dbeta_acc, N, C, hw, groups=32, res=None):
    """``res``: added to dx in fp32 before its rounding (the residual branch's gradient), instead of a separate add pass"""
    assert res is None or (res.shape == dx.shape and res.dtype == dx.dtype and res.is_contiguous() and res.data_ptr() != dx.data_ptr())
    ws, wsn = _ws("db1_groupnorm_gelu_nhwc_bwd_workspace_bytes", (int(N),), x.device)   # per-sample rows, summed in a fixed order (no atomics)
    lib.call("db1_groupnorm_gelu_nhwc_bwd", P(dy), P(x), P(gamma), P(beta), P(mean), P(rstd), P(dx), P(res) if res is not None else _vp(0), P(dgamma_acc), P(dbeta_acc),
             N, C, hw, groups, dt_code(x), dt_code(gamma), ws, wsn, stream())


def sumsq_acc(x, acc):
    _timed("sumsq", float(x.numel() * x.element_size()), lambda: lib.call("db1_sumsq_acc", P(x), P(acc), x.numel(), dt_code(x), stream()))


def grad_norm_sq(x, acc):
    """acc[0] = sum(x^2), deterministic (db1_grad_norm_sq: fixed-order partial sums, no atomics): what the engine's global-norm clip uses"""
    ws, wsn = _ws("db1_grad_norm_sq_workspace_bytes", (x.numel(),), x.device)
    _timed("sumsq", float(x.numel() * x.element_size()), lambda: lib.call("db1_grad_norm_sq", P(x), P(acc), x.numel(), dt_code(x), ws, wsn, stream()))


def adam_step(p32, g, m, v, p_work, lr, beta1, beta2, eps, wd, adamw, step, gscale=1.0, clip=0.0, norm_sq=None):
    # p, m, v read + written (24 B), g read, bf16 working copy written
    _timed("adam", float(p32.numel() * (24 + g.element_size() + (2 if p_work is not None else 0))),
           lambda: lib.call("db1_adam_step", P(p32), P(g), P(m), P(v), P(p_work), p32.numel(), float(lr), float(beta1), float(beta2), float(eps),
                            float(wd), int(bool(adamw)), int(step), float(gscale), float(clip), P(norm_sq), dt_code(g),
                            dt_code(p_work) if p_work is not None else 0, stream()))


def mulaw_discretize(x, ids, is_action, num_bins=1024, mu=100.0, M=256.0):
    assert x.dtype == torch.float32 and ids.dtype == torch.int32
    lib.call("db1_mulaw_discretize", P(x), P(ids), x.numel(), int(bool(is_action)), num_bins, float(mu), float(M), stream())


def mulaw_decode(ids, out, is_action, num_bins=1024, mu=100.0, M=256.0, oob_flag=None):
    assert ids.dtype in (torch.int32, torch.int64) and out.dtype == torch.float32 and ids.numel() == out.numel()
    assert oob_flag is None or oob_flag.dtype == torch.int32
    lib.call("db1_mulaw_decode", P(ids), P(out), ids.numel(), int(ids.dtype == torch.int64), int(bool(is_action)), num_bins, float(mu),
             float(M), P(oob_flag), stream())


GEMM_KERNELS = {0: "strided-fp32", 1: "tile128", 2: "tile256", 3: "pp-k64", 4: "pp-k32", 5: "w4", 6: "skinny", 7: "w4n"}


def gemm_kernel_choice(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, beta: float = 0.0, ws_bytes: int = -1):
    """(kernel name, split-K?, tail-call?) that ``gemm(a, b, out)`` would take (db1_gemm_kernel_choice)"""
    M, K = a.shape
    N = b.shape[1]
    code = lib.load().db1_gemm_kernel_choice(M, N, K, dt_code(a), dt_code(b), dt_code(out), a.stride(0), a.stride(1), b.stride(0), b.stride(1),
                                             out.stride(0), out.stride(1), 1, 1, float(beta), int(ws_bytes))
    return GEMM_KERNELS[code & 15], bool(code & 16), bool(code & 32)


def gemm_force_generic(on: bool):
    """test hook (include/db1_hip_test.h), thread-local"""
    lib.load().db1_test_gemm_force_generic(1 if on else 0)


def gemm_tile_override(tile: int):
    """test / tuning hook (include/db1_hip_test.h), thread-local: 0 = measured heuristics; 128 / 256 / 512 / 1024 pin one bf16 tile kernel"""
    lib.load().db1_test_gemm_tile_override(int(tile))


def flash_fwd2(mode):
    """test hook (include/db1_hip_test.h), thread-local: 0 / False = the compiled key-block loop, 1 / True = the default (hand-scheduled
    4-wave forward), 2 = the hand-scheduled 8-wave forward"""
    lib.load().db1_test_flash_fwd2(int(mode))
