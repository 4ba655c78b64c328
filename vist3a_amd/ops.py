"""Thin torch-tensor wrappers over the C ABI (include/vist3a_hip.h).

PyTorch supplies device memory and the current HIP stream only; all arithmetic happens inside
libvist3a_hip.so.  Every wrapper validates dtype/device/contiguity and raises on any non-zero return."""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import torch

from . import lib as L

bf16 = torch.bfloat16
f32 = torch.float32


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk2d(t: torch.Tensor, name: str, dtypes) -> None:
    if not t.is_cuda:
        raise ValueError(f"{name} must be a device tensor (the HIP path has no CPU fallback)")
    if t.dtype not in dtypes:
        raise TypeError(f"{name} has dtype {t.dtype}, expected one of {dtypes}")
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name} must be 2-D with a contiguous last dim, got shape {tuple(t.shape)} stride {t.stride()}")


class GemmProbe:
    """Optional HIP-event instrumentation of the GEMM launches that resolve to one tile configuration (= one kernel symbol):
    accumulates algorithmic FLOPs and event pairs so bench.py can report that kernel's live roofline numbers."""

    def __init__(self, tile: int, fp8: bool = False, stride: int = 1):
        """stride > 1: only every stride-th matching launch is bracketed by events (an event pair costs ~5 us of stream time; pick a
        stride coprime to the period of the launch sequence so that every shape is sampled)"""
        self.tile, self.fp8, self.flops, self.events, self.active = tile, fp8, 0.0, [], False
        self.stride, self.seen = max(1, int(stride)), 0

    def take(self) -> bool:
        self.seen += 1
        return self.seen % self.stride == 0

    def summary(self):
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.events]
        n = len(ms)
        return dict(launches=n, total_ms=sum(ms), avg_ms=(sum(ms) / n if n else 0.0), flops=self.flops,
                    flops_per_launch=(self.flops / n if n else 0.0))


_probe: Optional[GemmProbe] = None


def set_gemm_probe(p: Optional[GemmProbe]) -> None:
    global _probe
    _probe = p


class FlopMeter:
    """Algorithmic FLOPs of the matrix-pipe launches issued through this module while the meter is installed (`set_flop_meter`), by kind:
    GEMM 2MNK, convolution 2 x output pixels x Cout x taps x Cin (REAL channels; the fp32-equivalent form counts the fp32 convolution once,
    not its three bf16 products), attention 4 B H Nq Nk D over the keys that take part.  bench.py divides a stage's total by the stage's
    time for `stage_roofline`.  Launches replayed from a hipGraph do not pass through Python and are not counted."""

    def __init__(self):
        self.by_kind = {}

    def add(self, kind: str, flops: float) -> None:
        self.by_kind[kind] = self.by_kind.get(kind, 0.0) + float(flops)

    @property
    def total(self) -> float:
        return sum(self.by_kind.values())


_meter: Optional[FlopMeter] = None


def set_flop_meter(m: Optional[FlopMeter]) -> None:
    global _meter
    _meter = m


_gemm_ws = {}   # per (device, thread) split-K partial buffer (see _attn_ws)
# development A/B knob: V3A_TILE_OVERRIDE="MxNxK:tile,..." forces a tile for plain launches (tile = -1, no batch / split-K / transposed tail) of a shape
_tile_override = {tuple(int(v) for v in e.split(":")[0].split("x")): int(e.split(":")[1])
                  for e in os.environ.get("V3A_TILE_OVERRIDE", "").split(",") if ":" in e}


def gemm(
    a: torch.Tensor,
    w: torch.Tensor,
    bias: Optional[torch.Tensor] = None,
    *,
    out: Optional[torch.Tensor] = None,
    act: int = L.ACT_NONE,
    residual: Optional[torch.Tensor] = None,
    scale: Optional[torch.Tensor] = None,
    rows_per_batch: int = 0,
    round_after_scale: bool = False,
    out_f32: bool = False,
    bias_row: bool = False,
    tile: int = -1,
    residual2: Optional[torch.Tensor] = None,
    res_row_mod: int = 0,
    relu_out: bool = False,
    out_rows: Optional[tuple] = None,
    a_scale: Optional[torch.Tensor] = None,
    w_scale: Optional[torch.Tensor] = None,
    split_k: int = 1,
    batch: Optional[tuple] = None,
    row_sumsq: Optional[torch.Tensor] = None,
    t_out: Optional[torch.Tensor] = None,
    t_col0: int = 0,
) -> torch.Tensor:
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T); see v3a_gemm_bf16_nt for the epilogue order.
    batch = (count, a_stride, w_stride, out_stride[, residual_stride]) in elements: `count` problems of this shape in one launch, problem z
    offset by z * stride in each operand (0 = shared; residual_stride defaults to out_stride); a / w / out / residual describe problem 0.
    row_sumsq (f32 [M, N // 32], written): per (row, 32-column block) sum of squares of bf16(acc + bias) - the statistics of an RMS norm the
    consumer applies itself (ops.xattn_probs q_row_sumsq); plain bias epilogue only.
    out_rows=(group, skip, off) scatters output row m to m + (m//group)*skip + off (out must be given).
    t_out / t_col0 (transposed tail): output columns n >= t_col0 go to t_out[n - t_col0, m] (bf16 [N - t_col0, >= M]) instead of `out`, which then
    holds the first t_col0 columns only - the fused q | k | v projection: q | k row-major, V^T for the flash kernel, one launch.  Bias only.

    scale: f32 [N] (LayerScale) or [nbatch, N] together with rows_per_batch (AdaLN gate).
    split_k > 1 (bf16 only): K cut into split_k slices computed side by side, summed by a second launch (few output tiles, long K).
    a_scale / w_scale (f32 [M] / [N]): a and w are e4m3 bytes from `quantize_fp8_rows`; runs v3a_gemm_fp8_nt (tile then indexes its tiles)."""
    in_dt = (bf16,) if a_scale is None else (torch.uint8, torch.float8_e4m3fn)
    _chk2d(a, "a", in_dt)
    _chk2d(w, "w", in_dt)
    M, K = a.shape
    N, K2 = w.shape
    if K != K2:
        raise ValueError(f"K mismatch: a {tuple(a.shape)} vs w {tuple(w.shape)}")
    if t_out is not None:
        _chk2d(t_out, "t_out", (bf16,))
        if out is None or t_col0 <= 0 or t_col0 % 192 or t_col0 >= N or t_out.shape[0] < N - t_col0 or t_out.shape[1] < M or out.shape[1] < t_col0:
            raise ValueError("t_out needs an explicit out [M, >= t_col0], 0 < t_col0 < N in whole 192-column tiles, t_out [N - t_col0, >= M]")
    if out is None:
        if out_rows is not None:
            raise ValueError("out_rows needs an explicit out tensor")
        out = torch.empty((M, N), device=a.device, dtype=f32 if out_f32 else bf16)
    _chk2d(out, "out", (f32,) if out_f32 else (bf16,))
    # weight-streaming shapes (<= 128 rows against a big matrix) go to the skinny kernel: the tile GEMM would occupy N/128 CUs
    plain = t_out is None and split_k <= 1 and batch is None and row_sumsq is None and a_scale is None and scale is None and residual2 is None and out_rows is None and not relu_out and tile < 0 and res_row_mod == 0
    if plain and K % 512 == 0 and K >= 1024:
        if M <= 128 and N >= 512 and not bias_row:
            return _gemm_skinny(a, w, bias, out, act, residual, out_f32, transposed=False)
        if N <= 128 and M >= 512 and (bias is None or bias_row):
            return _gemm_skinny(w, a, bias, out, act, residual, out_f32, transposed=True)
    flags = L.GEMM_RELU_OUT if relu_out else 0
    if bias is not None:
        if bias.dtype != f32 or not bias.is_contiguous() or bias.numel() != (M if bias_row else N):
            raise ValueError("bias must be contiguous f32 of length N (or M with bias_row)")
        if bias_row:
            flags |= L.GEMM_BIAS_ROW
    sstride = 0
    if scale is not None:
        if scale.dtype != f32 or scale.stride(-1) != 1:
            raise ValueError("scale must be f32 with contiguous last dim")
        if scale.dim() == 2:
            if rows_per_batch <= 0:
                raise ValueError("2-D scale needs rows_per_batch")
            flags |= L.GEMM_SCALE_PER_BATCH
            sstride = scale.stride(0)
        if round_after_scale:
            flags |= L.GEMM_ROUND_AFTER_SCALE
    ldr = 0
    if residual is not None:
        _chk2d(residual, "residual", (bf16, f32))
        if residual.shape[1] != N or (res_row_mod == 0 and residual.shape[0] != M):
            raise ValueError("residual shape mismatch")
        if residual.dtype == f32:
            flags |= L.GEMM_RES_F32
        ldr = residual.stride(0)
    if out_f32:
        flags |= L.GEMM_OUT_F32
    if _tile_override and tile < 0 and batch is None and split_k == 1 and t_out is None and a_scale is None:
        tile = _tile_override.get((M, N, K), tile)
    args = L.GemmArgs(
        _ptr(a), _ptr(w), _ptr(out), _ptr(bias), _ptr(residual), _ptr(scale),
        M, N, K, a.stride(0), w.stride(0), out.stride(0), ldr,
        rows_per_batch, sstride, act, flags, tile,
        _ptr(residual2), residual2.stride(0) if residual2 is not None else 0, res_row_mod,
        *(out_rows if out_rows is not None else (0, 0, 0)),
    )
    if t_out is not None:
        if a_scale is not None:
            raise ValueError("t_out applies to the bf16 GEMM")
        args.C_t, args.ldct, args.t_col0 = t_out.data_ptr(), t_out.stride(0), t_col0
    if row_sumsq is not None:
        nbat = int(batch[0]) if batch is not None else 1
        if a_scale is not None or row_sumsq.dtype != f32 or not row_sumsq.is_contiguous() or row_sumsq.numel() < nbat * M * (N // 32) or N % 32:
            raise ValueError("row_sumsq must be contiguous f32 [batch * M, N // 32] (bf16 GEMM, N % 32 == 0)")
        if residual is not None or residual2 is not None or relu_out or out_rows is not None or act != L.ACT_NONE or scale is not None or bias_row:
            raise ValueError("row_sumsq is a by-product of the plain bias epilogue only")
        args.row_sumsq = row_sumsq.data_ptr()
    nb = 1
    if batch is not None:
        if a_scale is not None or split_k > 1:
            raise ValueError("batch applies to the plain bf16 GEMM")
        nb = int(batch[0])
        args.batch, args.a_batch_stride, args.b_batch_stride, args.c_batch_stride = nb, int(batch[1]), int(batch[2]), int(batch[3])
        args.res_batch_stride = int(batch[4]) if len(batch) > 4 else (int(batch[3]) if residual is not None else 0)
    if split_k > 1:
        if a_scale is not None:
            raise ValueError("split_k applies to the bf16 GEMM")
        nbytes = L.load().v3a_gemm_split_workspace_bytes(M, N, split_k)
        wk = (a.device, threading.get_ident())
        ws = _gemm_ws.get(wk)
        if ws is None or ws.numel() < nbytes:
            ws = _gemm_ws[wk] = torch.empty(nbytes, device=a.device, dtype=torch.uint8)
        args.split_k, args.workspace = split_k, ws.data_ptr()
    if _meter is not None:
        _meter.add("gemm_fp8" if a_scale is not None else "gemm", 2.0 * M * N * K * nb)
    if a_scale is not None:
        for t, n, nm in ((a_scale, M, "a_scale"), (w_scale, N, "w_scale")):
            if t is None or t.dtype != f32 or not t.is_contiguous() or t.numel() != n:
                raise ValueError(f"{nm} must be contiguous f32 of length {n}")
        fargs = L.GemmFp8Args(args, _ptr(a_scale), _ptr(w_scale))
        pr = _probe
        timed = pr is not None and pr.active and pr.fp8 and (tile if tile >= 0 else L.load().v3a_gemm_fp8_pick_tile(M, N)) == pr.tile and pr.take()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        L.check(L.load().v3a_gemm_fp8_nt(C.byref(fargs), _stream()), "v3a_gemm_fp8_nt")
        if timed:
            e1.record()
            pr.events.append((e0, e1))
            pr.flops += 2.0 * M * N * K
        return out
    pr = _probe
    # the tile the C side really runs (batch / split-K slices count towards the tile choice; a transposed tail forces its own tile)
    if pr is not None and pr.active and not pr.fp8 and (tile if tile >= 0 else L.load().v3a_gemm_pick_tile_ex(M, N, act, max(nb, split_k), int(t_out is not None))) == pr.tile and pr.take():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.load().v3a_gemm_bf16_nt(C.byref(args), _stream()), "v3a_gemm_bf16_nt")
        e1.record()
        pr.events.append((e0, e1))
        pr.flops += 2.0 * M * N * K * nb
        return out
    L.check(L.load().v3a_gemm_bf16_nt(C.byref(args), _stream()), "v3a_gemm_bf16_nt")
    return out


_skinny_ws = {}


def _gemm_skinny(x, wbig, bias, out, act, residual, out_f32, transposed):
    """x [Msmall<=128, K] against wbig [Nbig, K]: out[m][n] (or out[n][m] when transposed) through v3a_gemm_skinny_bf16."""
    Ms, K = x.shape
    Nb = wbig.shape[0]
    if bias is not None and (bias.dtype != f32 or not bias.is_contiguous() or bias.numel() != Nb):
        raise ValueError("bias must be contiguous f32 over the weight rows")
    lib = L.load()
    need = lib.v3a_gemm_skinny_workspace_bytes(Ms, Nb, K)
    if need < 0:
        L.check(int(need), "v3a_gemm_skinny_workspace_bytes")
    # per stream and per host thread: the partial sums are live until the finish kernel, and two threads (virtual ranks of
    # seqpar.ThreadWorld) may be inside the two-launch C call at the same time
    key = (x.device.index, torch.cuda.current_stream().cuda_stream, threading.get_ident())
    ws = _skinny_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _skinny_ws[key] = torch.empty(max(need, 1 << 22), device=x.device, dtype=torch.uint8)
    flags, ldr = 0, 0
    if residual is not None:
        _chk2d(residual, "residual", (bf16, f32))
        if tuple(residual.shape) != tuple(out.shape):
            raise ValueError("residual shape mismatch")
        if residual.dtype == f32:
            flags |= L.GEMM_RES_F32
        ldr = residual.stride(0)
    if out_f32:
        flags |= L.GEMM_OUT_F32
    args = L.GemmSkinnyArgs(_ptr(x), _ptr(wbig), _ptr(out), _ptr(bias), _ptr(residual), Ms, Nb, K, x.stride(0), wbig.stride(0),
                            out.stride(0), ldr, act, flags, int(transposed), _ptr(ws), ws.numel())
    if _meter is not None:
        _meter.add("gemm_skinny", 2.0 * Ms * Nb * K)
    L.check(lib.v3a_gemm_skinny_bf16(C.byref(args), _stream()), "v3a_gemm_skinny_bf16")
    return out


class ConvWeight:
    """A convolution weight packed for v3a_conv_bf16: w[CoutPad][Kpad] bf16 (k = tap-major, channel-minor), the
    K-chunk table, f32 bias.  Cin/Cout are padded to multiples of 8 with zeros (activations must carry CinPad)."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, device="cuda", dilation=(1, 1, 1)):
        """dilation (dT, dH, dW): tap (t, h, w) reads the input t dT / h dH / w dW positions from the window origin - the K-chunk table
        carries the scaled offsets (4 bits each: (k - 1) d <= 15), the kernels are unchanged; `k_eff` is the window extent (k - 1) d + 1."""
        w = weight.detach()
        if w.dim() == 4:  # conv2d -> kT = 1
            w = w[:, :, None]
        if w.dim() == 3:  # conv1d
            w = w[:, :, None, None]
        Cout, Cin, kT, kH, kW = w.shape
        self.Cout, self.Cin, self.k = Cout, Cin, (kT, kH, kW)
        self.dil = tuple(int(v) for v in dilation)
        if len(self.dil) != 3 or min(self.dil) < 1 or any((k - 1) * d_ > 15 for k, d_ in zip(self.k, self.dil)):
            raise ValueError(f"dilation {dilation} with kernel {self.k}: every (k - 1) * d must be <= 15 (4-bit tap offsets of the K-chunk table)")
        self.k_eff = tuple((k - 1) * d_ + 1 for k, d_ in zip(self.k, self.dil))
        self.CinP, self.CoutP = (Cin + 7) // 8 * 8, (Cout + 7) // 8 * 8
        K = kT * kH * kW * self.CinP
        self.Kpad = (K + 63) // 64 * 64
        wp = torch.zeros(self.CoutP, kT, kH, kW, self.CinP, dtype=f32)
        wp[:Cout, :, :, :, :Cin] = w.float().permute(0, 2, 3, 4, 1).cpu()
        full = torch.zeros(self.CoutP, self.Kpad, dtype=f32)
        full[:, :K] = wp.reshape(self.CoutP, K)
        self.w = full.to(device=device, dtype=bf16).contiguous()
        tab = torch.zeros(self.Kpad // 8, dtype=torch.int64)
        idx = torch.arange(K // 8)
        tap, c8 = idx // (self.CinP // 8), idx % (self.CinP // 8)
        dt, dh, dw = tap // (kH * kW) * self.dil[0], (tap // kW) % kH * self.dil[1], tap % kW * self.dil[2]
        tab[: K // 8] = (c8 * 8) | (dw << 16) | (dh << 20) | (dt << 24) | (1 << 31)
        tab = torch.where(tab >= 2 ** 31, tab - 2 ** 32, tab)
        self.ktab = tab.to(torch.int32).to(device).contiguous()
        b = torch.zeros(self.CoutP, dtype=f32)
        if bias is not None:
            b[:Cout] = bias.detach().float().cpu()
        self.bias = b.to(device)
        self.has_bias = bias is not None
        # second packing for the halo-tile kernel (csrc/conv_halo.hip; layout documented at v3a_conv_args.w_halo):
        # [Cout/96][kT][Cin/48][9][96][48], the six 16-byte chunks of row n rotated by 3 * ((n >> 3) & 1)
        self.w_halo = None
        if kH == 3 and kW == 3 and kT in (1, 3) and self.CinP % 48 == 0 and self.CoutP % 96 == 0 and self.dil == (1, 1, 1):
            nN, nC = self.CoutP // 96, self.CinP // 48
            h = wp.reshape(nN, 96, kT, 9, nC, 6, 8).permute(0, 2, 4, 3, 1, 5, 6).contiguous()   # [nN, kT, nC, 9, 96, 6 chunks, 8]
            rot = 3 * ((torch.arange(96) >> 3) & 1)
            src = (torch.arange(6)[None, :] - rot[:, None]) % 6                                  # position s of row n holds chunk (s - rot) % 6
            h = torch.gather(h, 5, src.view(1, 1, 1, 1, 96, 6, 1).expand(nN, kT, nC, 9, 96, 6, 8))
            self.w_halo = h.to(device=device, dtype=bf16).contiguous()


def conv(
    x: torch.Tensor, cw: ConvWeight, *, out: Optional[torch.Tensor] = None,
    stride=(1, 1, 1), pad=(0, 0, 0), out_size=None, ups2: bool = False, replicate: bool = False,
    act: int = L.ACT_NONE, residual: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None,
    out_f32: bool = False, tile: int = -1, residual2: Optional[torch.Tensor] = None, res_row_mod: int = 0,
    relu_out: bool = False, out_rows: Optional[tuple] = None, form_scale: Optional[float] = None,
) -> torch.Tensor:
    """x: channels-last [T,H,W,CinP] bf16 contiguous -> out [oT,oH,oW,CoutP].  pad = LEADING pad per dim.  Spatial
    dims follow PyTorch's symmetric-pad formula; the temporal dim is causal (all padding leading) when
    pad_T == k_T - 1 and symmetric otherwise.  `out_size` overrides.
    tile: -1 = automatic (the halo-tile kernel for wide 3x3(x3) layers, else the implicit GEMM with a heuristic tile), >= 0 = that
    implicit-GEMM tile, -2 = force the halo-tile kernel (error if the layer is not of its form), -3 = never the halo-tile kernel.
    An H-STRIP of a spatially sharded image carries one explicit halo row above and below (the neighbours' boundary rows, zeros at the image
    border) and is convolved VALID in H: pad = (pT, 0, pW) with out_size = (T, H - 2, W) - or, with ups2, pad = (pT, -1, pW) and out_size =
    (T, 2 (H - 2), 2 W) - both forms of the halo-tile kernel as well (v3a_conv_args).
    form_scale (with tile = -1): choose halo-tile vs implicit GEMM as if the layer were form_scale times larger - a rank's strip takes the
    kernel form the whole image would take, which keeps a sharded decode bit-identical to the unsharded one."""
    if x.dim() != 4 or not x.is_contiguous() or x.dtype != bf16 or not x.is_cuda:
        raise ValueError("x must be a contiguous device bf16 tensor [T,H,W,C]")
    T, H, W, Cin = x.shape
    if Cin != cw.CinP:
        raise ValueError(f"x has {Cin} channels, packed weight expects {cw.CinP}")
    kT, kH, kW = cw.k
    eT_, eH_, eW_ = cw.k_eff                      # window extents (= the kernel sizes without dilation)
    eH, eW = (2 * H, 2 * W) if ups2 else (H, W)
    if out_size is None:
        if kT > 1 and pad[0] == eT_ - 1:
            oT = (T + pad[0] - eT_) // stride[0] + 1
        else:
            oT = (T + 2 * pad[0] - eT_) // stride[0] + 1
        oH = (eH + 2 * pad[1] - eH_) // stride[1] + 1
        oW = (eW + 2 * pad[2] - eW_) // stride[2] + 1
    else:
        oT, oH, oW = out_size
    M = oT * oH * oW
    if out is None:
        out = torch.empty((oT, oH, oW, cw.CoutP), device=x.device, dtype=f32 if out_f32 else bf16)
    o2 = out.view(-1, out.shape[-1])
    flags = L.GEMM_RELU_OUT if relu_out else 0
    ldr = 0
    r2 = None
    if residual is not None:
        r2 = residual.view(-1, residual.shape[-1])
        if r2.dtype == f32:
            flags |= L.GEMM_RES_F32
        ldr = r2.stride(0)
    if out_f32:
        flags |= L.GEMM_OUT_F32
    args = L.ConvArgs(
        _ptr(x), _ptr(cw.w), _ptr(cw.ktab), _ptr(o2), _ptr(cw.bias if cw.has_bias else None), _ptr(r2), _ptr(scale),
        T, H, W, Cin, oT, oH, oW, cw.CoutP, cw.Kpad,
        stride[0], stride[1], stride[2], pad[0], pad[1], pad[2],
        int(ups2), int(replicate), o2.stride(0), ldr, act, flags, tile,
        _ptr(residual2), residual2.view(-1, residual2.shape[-1]).stride(0) if residual2 is not None else 0, res_row_mod,
        *(out_rows if out_rows is not None else (0, 0, 0)),
        _ptr(cw.w_halo), kT if cw.w_halo is not None else 0,
    )
    if form_scale is not None and tile == -1 and cw.w_halo is not None:
        args.tile = -2 if L.load().v3a_conv_halo_tiles(C.byref(args)) * form_scale >= 512 else -3
    if _meter is not None:
        _meter.add("conv", 2.0 * M * cw.Cout * kT * kH * kW * cw.Cin)
    L.check(L.load().v3a_conv_bf16(C.byref(args), _stream()), "v3a_conv_bf16")
    return out


# ------------------------------------------------------------------------------------------------ fp32-equivalent ("split bf16") path
# A PAIR is a contiguous bf16 tensor [2, ...]: plane 0 = hi = bf16(x), plane 1 = lo = bf16(x - hi); x = hi + lo (include/vist3a_hip.h,
# v3a_conv_split).  The layers the reference runs with autocast off (anysplat_stitched.py:335) carry their activations in this form.

def pair_value(p: torch.Tensor) -> torch.Tensor:
    """the fp32 value of a pair (tests / glue; the kernels never materialise it)"""
    return p[0].float() + p[1].float()


def split_f32(x: torch.Tensor) -> torch.Tensor:
    """contiguous f32 [...] (numel % 8 == 0) -> pair [2, ...]"""
    if x.dtype != f32 or not x.is_contiguous() or not x.is_cuda or x.numel() % 8:
        raise ValueError("x must be a contiguous device f32 tensor with numel % 8 == 0")
    out = torch.empty((2,) + tuple(x.shape), device=x.device, dtype=bf16)
    L.check(L.load().v3a_split_f32(_ptr(x), _ptr(out[0]), _ptr(out[1]), x.numel(), _stream()), "v3a_split_f32")
    return out


def layernorm_pair(x: torch.Tensor, *, weight: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, eps: float = 1e-5,
                   M: Optional[int] = None, in_rows: tuple = (0, 0, 0)) -> torch.Tensor:
    """F.layer_norm of f32 rows -> pair [2, M, d]; in_rows as in `layernorm`."""
    _chk2d(x, "x", (f32,))
    d = x.shape[1]
    if M is None:
        M = x.shape[0]
    for t in (weight, bias):
        if t is not None and (t.dtype != f32 or not t.is_contiguous() or t.numel() != d):
            raise ValueError("affine weight/bias must be contiguous f32 [d]")
    out = torch.empty((2, M, d), device=x.device, dtype=bf16)
    L.check(L.load().v3a_layernorm_pair(_ptr(x), _ptr(out[0]), _ptr(out[1]), _ptr(weight), _ptr(bias), M, d, x.stride(0), d, eps,
                                        *in_rows, _stream()), "v3a_layernorm_pair")
    return out


def bilinear_cl_pair(x: torch.Tensor, size, *, align_corners: bool, add: Optional[torch.Tensor] = None,
                     table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pair [2,T,h,w,C] -> pair [2,T,H,W,C] (+ pair `add` of the output's shape, + f32 `table` [H*W, C])."""
    if x.dtype != bf16 or not x.is_contiguous() or x.dim() != 5 or x.shape[0] != 2:
        raise ValueError("x must be a contiguous bf16 pair [2,T,h,w,C]")
    _, T, h, w, Cc = x.shape
    H, W = size
    out = torch.empty((2, T, H, W, Cc), device=x.device, dtype=bf16)
    if add is not None and (add.dtype != bf16 or not add.is_contiguous() or tuple(add.shape) != tuple(out.shape)):
        raise ValueError("add must be a contiguous pair with the output's shape")
    if table is not None and (table.dtype != f32 or not table.is_contiguous() or table.numel() != H * W * Cc):
        raise ValueError("table must be contiguous f32 [H*W, C]")
    L.check(L.load().v3a_bilinear_cl_pair(_ptr(x[0]), _ptr(x[1]), _ptr(out[0]), _ptr(out[1]), _ptr(add[0]) if add is not None else None,
                                          _ptr(add[1]) if add is not None else None, _ptr(table), T, h, w, H, W, Cc, int(align_corners),
                                          _stream()), "v3a_bilinear_cl_pair")
    return out


class ConvWeightSplit:
    """An fp32 convolution weight packed for v3a_conv_split: w = wh + wl (bf16 pair), matrix [CoutP][Kpad] = (wh | wl | wh) along K, each
    range tap-major / channel-minor like ConvWeight; the chunk table reads x's LO plane for the first range (bit 28) and its HI plane for
    the other two:  acc = xl.wh + xh.wl + xh.wh  (small terms first).  f32 bias."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, device="cuda"):
        w = weight.detach()
        if w.dim() == 4:
            w = w[:, :, None]
        if w.dim() == 3:
            w = w[:, :, None, None]
        Cout, Cin, kT, kH, kW = w.shape
        self.Cout, self.Cin, self.k = Cout, Cin, (kT, kH, kW)
        self.CinP, self.CoutP = (Cin + 7) // 8 * 8, (Cout + 7) // 8 * 8
        K = kT * kH * kW * self.CinP
        self.Kpad = (3 * K + 63) // 64 * 64
        wp = torch.zeros(self.CoutP, kT, kH, kW, self.CinP, dtype=f32)
        wp[:Cout, :, :, :, :Cin] = w.float().permute(0, 2, 3, 4, 1).cpu()
        wp = wp.reshape(self.CoutP, K)
        wh = wp.to(bf16)
        wl = (wp - wh.float()).to(bf16)
        full = torch.zeros(self.CoutP, self.Kpad, dtype=bf16)
        full[:, :K], full[:, K:2 * K], full[:, 2 * K:3 * K] = wh, wl, wh
        self.w = full.to(device).contiguous()
        tab = torch.zeros(self.Kpad // 8, dtype=torch.int64)
        idx = torch.arange(K // 8)
        tap, c8 = idx // (self.CinP // 8), idx % (self.CinP // 8)
        dt, dh, dw = tap // (kH * kW), (tap // kW) % kH, tap % kW
        e = (c8 * 8) | (dw << 16) | (dh << 20) | (dt << 24) | (1 << 31)
        tab[: K // 8], tab[K // 8: 2 * K // 8], tab[2 * K // 8: 3 * K // 8] = e | (1 << 28), e, e
        tab = torch.where(tab >= 2 ** 31, tab - 2 ** 32, tab)
        self.ktab = tab.to(torch.int32).to(device).contiguous()
        b = torch.zeros(self.CoutP, dtype=f32)
        if bias is not None:
            b[:Cout] = bias.detach().float().cpu()
        self.bias = b.to(device)
        self.has_bias = bias is not None
        # second packing for the halo-tile form (csrc/conv_halo_split.hip; layout at v3a_conv_split_args in the header):
        # [Cout/BN][Cin/16][9][2 planes][BN][2 chunk positions][8], position s of row n holds chunk s ^ ((n >> 3) & 1)
        self.w_halo = None
        BN = 128 if self.CoutP % 128 == 0 else (64 if self.CoutP % 64 == 0 else (32 if self.CoutP % 32 == 0 else 0))
        if kT == 1 and kH == 3 and kW == 3 and self.CinP % 16 == 0 and BN:
            nN, nS = self.CoutP // BN, self.CinP // 16
            planes = torch.stack([wh.float(), wl.float()])                      # [2, CoutP, 9 * CinP] (tap-major, channel-minor)
            h = planes.reshape(2, nN, BN, 9, nS, 2, 8).permute(1, 4, 3, 0, 2, 5, 6).contiguous()   # [nN, nS, 9, 2, BN, 2 chunks, 8]
            rot = (torch.arange(BN) >> 3) & 1
            src = torch.arange(2)[None, :] ^ rot[:, None]                       # position s of row n <- chunk s ^ rot(n)
            h = torch.gather(h, 5, src.view(1, 1, 1, 1, BN, 2, 1).expand(nN, nS, 9, 2, BN, 2, 8))
            self.w_halo = h.to(device=device, dtype=bf16).contiguous()


def conv_split(
    x: torch.Tensor, cw: ConvWeightSplit, *, out: Optional[torch.Tensor] = None, stride=(1, 1, 1), pad=(0, 0, 0), out_size=None,
    act: int = L.ACT_NONE, residual: Optional[torch.Tensor] = None, out_f32: bool = False, tile: int = -1,
    residual2: Optional[torch.Tensor] = None, res_row_mod: int = 0, relu_out: bool = False, out_rows: Optional[tuple] = None,
    form_frames: Optional[int] = None,
) -> torch.Tensor:
    """fp32-equivalent convolution (v3a_conv_split).  x: pair [2,T,H,W,CinP] -> pair [2,oT,oH,oW,CoutP] (or f32 [oT,oH,oW,CoutP] with
    out_f32).  residual: an f32 tensor (table; res_row_mod as in `conv`) or a pair; residual2: a pair.  No bf16 rounding anywhere between
    the accumulator and the store; geometry arguments as in `conv`.
    tile: -1 = automatic (the halo-tile form for wide 3x3 layers, else the implicit GEMM with a heuristic tile), >= 0 = that implicit-GEMM
    tile, -2 = force the halo-tile form, -3 = never the halo-tile form.
    form_frames (with tile = -1): choose between the two forms as if the clip had this many frames - a rank that holds a few views of a
    scene takes the form the whole scene would take, so view-sharded results stay bit-identical to the unsharded ones (the two forms
    differ in fp32 summation order)."""
    if x.dim() != 5 or x.shape[0] != 2 or not x.is_contiguous() or x.dtype != bf16 or not x.is_cuda:
        raise ValueError("x must be a contiguous device bf16 pair [2,T,H,W,C]")
    _, T, H, W, Cin = x.shape
    if Cin != cw.CinP:
        raise ValueError(f"x has {Cin} channels, packed weight expects {cw.CinP}")
    kT, kH, kW = cw.k
    if out_size is None:
        if kT > 1 and pad[0] == kT - 1:
            oT = (T + pad[0] - kT) // stride[0] + 1
        else:
            oT = (T + 2 * pad[0] - kT) // stride[0] + 1
        oH = (H + 2 * pad[1] - kH) // stride[1] + 1
        oW = (W + 2 * pad[2] - kW) // stride[2] + 1
    else:
        oT, oH, oW = out_size
    if out is None:
        if out_rows is not None:
            raise ValueError("out_rows needs an explicit out tensor")
        out = torch.empty((oT, oH, oW, cw.CoutP) if out_f32 else (2, oT, oH, oW, cw.CoutP), device=x.device, dtype=f32 if out_f32 else bf16)
    if out_f32:
        if out.dtype != f32:
            raise ValueError("out must be f32")
        o_hi, o_lo = out.view(-1, out.shape[-1]), None
    else:
        if out.dtype != bf16 or out.shape[0] != 2 or not out.is_contiguous():
            raise ValueError("out must be a contiguous bf16 pair")
        o_hi, o_lo = out[0].view(-1, out.shape[-1]), out[1]
    flags = (L.GEMM_RELU_OUT if relu_out else 0) | (L.GEMM_OUT_F32 if out_f32 else 0)
    r_hi = r_lo = None
    ldr = 0
    if residual is not None:
        if residual.dtype == f32:
            flags |= L.GEMM_RES_F32
            r_hi = residual.view(-1, residual.shape[-1])
        else:
            if residual.dtype != bf16 or residual.shape[0] != 2 or not residual.is_contiguous():
                raise ValueError("residual must be f32 or a contiguous bf16 pair")
            r_hi, r_lo = residual[0].view(-1, residual.shape[-1]), residual[1]
        ldr = r_hi.stride(0)
    q_hi = q_lo = None
    if residual2 is not None:
        if residual2.dtype != bf16 or residual2.shape[0] != 2 or not residual2.is_contiguous():
            raise ValueError("residual2 must be a contiguous bf16 pair")
        q_hi, q_lo = residual2[0].view(-1, residual2.shape[-1]), residual2[1]
    cargs = L.ConvArgs(
        _ptr(x[0]), _ptr(cw.w), _ptr(cw.ktab), _ptr(o_hi), _ptr(cw.bias if cw.has_bias else None), _ptr(r_hi), None,
        T, H, W, Cin, oT, oH, oW, cw.CoutP, cw.Kpad,
        stride[0], stride[1], stride[2], pad[0], pad[1], pad[2],
        0, 0, o_hi.stride(0), ldr, act, flags, tile,
        _ptr(q_hi), q_hi.stride(0) if q_hi is not None else 0, res_row_mod,
        *(out_rows if out_rows is not None else (0, 0, 0)),
        _ptr(cw.w_halo), 1 if cw.w_halo is not None else 0,
    )
    args = L.ConvSplitArgs(cargs, _ptr(x[1]), _ptr(o_lo), _ptr(r_lo), _ptr(q_lo))
    if form_frames is not None and tile == -1 and cw.w_halo is not None:
        per_frame = L.load().v3a_conv_split_halo_tiles(C.byref(args)) // T          # 0: the layer has no halo form
        args.c.tile = -2 if per_frame * form_frames >= 128 else -3
    if _meter is not None:
        _meter.add("conv_f32_equivalent", 2.0 * oT * oH * oW * cw.Cout * kT * kH * kW * cw.Cin)
    L.check(L.load().v3a_conv_split(C.byref(args), _stream()), "v3a_conv_split")
    return out


_attn_ws = {}   # per (device, thread) workspace of the key-split attention: stream-ordered reuse within a thread; virtual ranks
                # (seqpar.ThreadWorld: threads sharing one stream) must not share it - their main / merge launches interleave


def attention(
    q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, *,
    B: int, H: int, Nq: int, Nk: int, D: int,
    q_batch_stride: int, k_batch_stride: int, vt_batch_stride: int, o_batch_stride: int,
    scale: Optional[float] = None, kv_period: int = 0, kv_valid: int = 0,
    rel_bias: Optional[torch.Tensor] = None, rel_bias_center: int = 0, key_bias: Optional[torch.Tensor] = None,
    key_bias_first: int = 0, kv_seg: int = 0, k_seg_stride: int = 0, vt_seg_stride: int = 0, kv_split: int = 1,
) -> torch.Tensor:
    """q,k,out: 2-D views [B*N, >=H*D] (row stride = their stride(0)); vt: 2-D [H*D, >=B*vt_batch_stride].
    rel_bias (D=64 only): fp32 [H, n] table, entry (key - query + rel_bias_center) is added to the scaled score.
    key_bias (D=128 only): fp32 [B, >=Nk], added to the scaled score of every query (log-multiplicity of merged keys).
    kv_seg (D=128 only): k / vt are the FIRST of Nk / kv_seg equally laid out segments k_seg_stride / vt_seg_stride elements apart
    (the per-rank slabs of an all-gather), read in place.
    kv_period / kv_valid: key k takes part only if (k % kv_period) < kv_valid (padded per-view token layout); not together with a bias.
    kv_split > 1 (D=128 only): the keys of every query block are divided among kv_split workgroups whose partial softmaxes a second
    launch merges - for launches with too few query blocks to fill the chip; deterministic, not bit-identical to kv_split = 1."""
    ws = None
    if kv_split > 1:
        nbytes = L.load().v3a_attention_split_workspace_bytes(B, H, Nq, D, kv_split)
        wk = (q.device, threading.get_ident())
        ws = _attn_ws.get(wk)
        if ws is None or ws.numel() < nbytes:
            ws = _attn_ws[wk] = torch.empty(nbytes, device=q.device, dtype=torch.uint8)
    if key_bias is not None and (key_bias.dtype != f32 or key_bias.dim() != 2 or key_bias.shape[0] != B or key_bias.stride(1) != 1):
        raise ValueError("key_bias must be fp32 [B, n] with a contiguous last dim")
    if rel_bias is not None and (rel_bias.dtype != f32 or rel_bias.dim() != 2 or rel_bias.shape[0] != H or not rel_bias.is_contiguous()):
        raise ValueError("rel_bias must be contiguous fp32 [H, n]")
    for t, n in ((q, "q"), (k, "k"), (vt, "vt"), (out, "out")):
        _chk2d(t, n, (bf16,))
    args = L.AttnArgs(
        _ptr(q), _ptr(k), _ptr(vt), _ptr(out),
        q_batch_stride, k_batch_stride, vt_batch_stride, o_batch_stride,
        q.stride(0), k.stride(0), vt.stride(0), out.stride(0),
        B, H, Nq, Nk, D, float(scale if scale is not None else D ** -0.5), kv_period, kv_valid,
        _ptr(rel_bias), rel_bias.shape[1] if rel_bias is not None else 0, rel_bias_center,
        _ptr(key_bias), key_bias.stride(0) if key_bias is not None else 0, key_bias_first,
        kv_seg, k_seg_stride, vt_seg_stride, kv_split, _ptr(ws),
    )
    if _meter is not None:
        _meter.add("attention", 4.0 * B * H * Nq * (Nk * kv_valid / kv_period if kv_period > 0 else Nk) * D)
    L.check(L.load().v3a_attention_fwd_bf16(C.byref(args), _stream()), "v3a_attention_fwd_bf16")
    return out


def xattn_probs(
    q: torch.Tensor, k: torch.Tensor, out: torch.Tensor, *, B: int, H: int, Nq: int, Nk: int, Lkp: int,
    q_batch_stride: int, k_batch_stride: int, p_batch_stride: int, scale: Optional[float] = None,
    key_bias: Optional[torch.Tensor] = None, key_bias_first: int = 0, q_row_sumsq: Optional[torch.Tensor] = None, q_eps: float = 1e-6,
) -> torch.Tensor:
    """Normalised cross-attention probabilities over <= 128 per-prompt keys (csrc/xattn_probs.hip, head_dim 128):
    out[b*p_batch_stride/ld + m, h*Lkp + j] = bf16(softmax_j(scale * q_h . k_h[j] + key_bias[b, j])), zeros for Nk <= j < Lkp.
    q, k, out: 2-D bf16 views with row stride stride(0); key_bias fp32 [B, >= Nk] applies to keys >= key_bias_first.
    q_row_sumsq (f32 [B * Nq, parts], ops.gemm row_sumsq of the projection that made q): q is un-normalised; the scores of a query are
    multiplied by rsqrt(sum(parts) / (H * 128) + q_eps), the row factor of the RMS norm across heads (fold its weight into k)."""
    for t, n in ((q, "q"), (k, "k"), (out, "out")):
        _chk2d(t, n, (bf16,))
    if key_bias is not None and (key_bias.dtype != f32 or key_bias.dim() != 2 or key_bias.shape[0] != B or key_bias.stride(1) != 1):
        raise ValueError("key_bias must be fp32 [B, n] with a contiguous last dim")
    if q_row_sumsq is not None and (q_row_sumsq.dtype != f32 or q_row_sumsq.dim() != 2 or not q_row_sumsq.is_contiguous()
                                    or q_row_sumsq.shape[0] < B * Nq or q_batch_stride != Nq * q.stride(0)):
        raise ValueError("q_row_sumsq must be contiguous f32 [B * Nq, parts] over densely stacked batch items of q")
    args = L.XattnProbsArgs(
        _ptr(q), _ptr(k), _ptr(out), _ptr(key_bias), q_batch_stride, k_batch_stride, p_batch_stride,
        q.stride(0), k.stride(0), out.stride(0), B, H, Nq, Nk, 128, Lkp,
        key_bias.stride(0) if key_bias is not None else 0, key_bias_first, float(scale if scale is not None else 128 ** -0.5),
        _ptr(q_row_sumsq), q_row_sumsq.shape[1] if q_row_sumsq is not None else 0, float(q_eps),
    )
    if _meter is not None:
        _meter.add("attention", 2.0 * B * H * Nq * Nk * 128)   # scores only: P.V is the finishing GEMM's
    L.check(L.load().v3a_xattn_probs_bf16(C.byref(args), _stream()), "v3a_xattn_probs_bf16")
    return out


def quantize_fp8(x: torch.Tensor, scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 [rows, cols] (row stride = x.stride(0)) -> e4m3 bytes (uint8 storage) of x / scale, RNE, clamped to +-448."""
    _chk2d(x, "x", (bf16,))
    rows, cols = x.shape
    if out is None:
        out = torch.empty(rows, cols, device=x.device, dtype=torch.uint8)
    if out.dtype != torch.uint8 or out.dim() != 2 or out.stride(1) != 1 or tuple(out.shape) != (rows, cols):
        raise ValueError("out must be a uint8 [rows, cols] tensor with a contiguous last dim")
    L.check(L.load().v3a_quantize_fp8(_ptr(x), _ptr(out), rows, cols, x.stride(0), out.stride(0), float(scale), _stream()), "v3a_quantize_fp8")
    return out


def quantize_fp8_rows(x: torch.Tensor, out: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None):
    """(e4m3 bytes [rows, cols], f32 [rows]): per-row dynamic quantisation, scale = max(amax, 1e-12) / 448 (v3a_quantize_fp8_rows)."""
    _chk2d(x, "x", (bf16,))
    rows, cols = x.shape
    out = torch.empty((rows, cols), device=x.device, dtype=torch.uint8) if out is None else out
    scale = torch.empty(rows, device=x.device, dtype=f32) if scale is None else scale
    if out.dtype not in (torch.uint8, torch.float8_e4m3fn) or out.stride(1) != 1 or scale.dtype != f32 or not scale.is_contiguous() or scale.numel() != rows:
        raise ValueError("out must be bytes with unit inner stride, scale contiguous f32 [rows]")
    L.check(L.load().v3a_quantize_fp8_rows(_ptr(x), _ptr(out), _ptr(scale), rows, cols, x.stride(0), out.stride(0), _stream()), "v3a_quantize_fp8_rows")
    return out, scale


def attention_fp8(q8: torch.Tensor, k8: torch.Tensor, vt8: torch.Tensor, out: torch.Tensor, *, B: int, H: int, Nq: int, Nk: int,
                  q_batch_stride: int, k_batch_stride: int, vt_batch_stride: int, o_batch_stride: int, scale: Optional[float] = None,
                  q_scale: float = 1.0, k_scale: float = 1.0, v_scale: float = 1.0, kv_seg: int = 0, k_seg_stride: int = 0,
                  vt_seg_stride: int = 0, kv_split: int = 1) -> torch.Tensor:
    """fp8 (e4m3) flash attention, head dim 128: q8/k8 uint8 [B*N, >=H*128], vt8 uint8 [H*128, >=B*vt_batch_stride], out bf16.
    kv_seg / k_seg_stride / vt_seg_stride (bytes) and kv_split as in `attention`."""
    for t, n in ((q8, "q8"), (k8, "k8"), (vt8, "vt8")):
        _chk2d(t, n, (torch.uint8,))
    _chk2d(out, "out", (bf16,))
    ws = None
    if kv_split > 1:
        nbytes = L.load().v3a_attention_split_workspace_bytes(B, H, Nq, 128, kv_split)
        wk = (q8.device, threading.get_ident())
        ws = _attn_ws.get(wk)
        if ws is None or ws.numel() < nbytes:
            ws = _attn_ws[wk] = torch.empty(nbytes, device=q8.device, dtype=torch.uint8)
    args = L.AttnFp8Args(_ptr(q8), _ptr(k8), _ptr(vt8), _ptr(out), q_batch_stride, k_batch_stride, vt_batch_stride, o_batch_stride,
                         q8.stride(0), k8.stride(0), vt8.stride(0), out.stride(0), B, H, Nq, Nk, 128,
                         float(scale if scale is not None else 128 ** -0.5), float(q_scale), float(k_scale), float(v_scale),
                         kv_seg, k_seg_stride, vt_seg_stride, kv_split, _ptr(ws))
    if _meter is not None:
        _meter.add("attention_fp8", 4.0 * B * H * Nq * Nk * 128)
    L.check(L.load().v3a_attention_fwd_fp8(C.byref(args), _stream()), "v3a_attention_fwd_fp8")
    return out


def layernorm(
    x: torch.Tensor, *, out: Optional[torch.Tensor] = None,
    weight: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
    scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
    rows_per_batch: int = 0, eps: float = 1e-6, out_dtype: torch.dtype = bf16,
    M: Optional[int] = None, in_rows: tuple = (0, 0, 0), out_rows: tuple = (0, 0, 0), rms: bool = False,
    fp8_scale: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """in_rows/out_rows = (group, skip, off): logical row m maps to physical row m + (m//group)*skip + off.
    rms=True: no mean subtraction (T5LayerNorm).
    fp8_scale (f32 [rows of out]): `out` is e4m3 bytes, the per-row quantisation (ops.quantize_fp8_rows) of the bf16 result."""
    _chk2d(x, "x", (bf16, f32))
    d = x.shape[1]
    if M is None:
        M = x.shape[0]
    if out is None:
        out = torch.empty((M, d), device=x.device, dtype=out_dtype)
    if fp8_scale is not None:
        _chk2d(out, "out", (torch.uint8, torch.float8_e4m3fn))
        if fp8_scale.dtype != f32 or not fp8_scale.is_contiguous() or fp8_scale.numel() < out.shape[0]:
            raise ValueError("fp8_scale must be contiguous f32 with one entry per output row")
    else:
        _chk2d(out, "out", (bf16, f32))
    mstride = 0
    if scale is not None:
        if scale.dtype != f32 or shift is None or shift.dtype != f32 or scale.dim() != 2 or shift.dim() != 2:
            raise ValueError("scale/shift must both be 2-D f32")
        if scale.stride(0) != shift.stride(0) or scale.stride(1) != 1 or shift.stride(1) != 1:
            raise ValueError("scale/shift must share a row stride and be contiguous in the last dim")
        mstride = scale.stride(0)
    for t in (weight, bias):
        if t is not None and (t.dtype != f32 or not t.is_contiguous()):
            raise ValueError("affine weight/bias must be contiguous f32")
    args = L.LayerNormArgs(
        _ptr(x), _ptr(out), _ptr(weight), _ptr(bias), _ptr(scale), _ptr(shift),
        M, d, x.stride(0), out.stride(0), rows_per_batch, mstride, eps,
        int(x.dtype == f32), int(out.dtype == f32), *in_rows, *out_rows, int(rms), _ptr(fp8_scale),
    )
    L.check(L.load().v3a_layernorm(C.byref(args), _stream()), "v3a_layernorm")
    return out


def rmsnorm_rope(
    x: torch.Tensor, weight: torch.Tensor, *, out: Optional[torch.Tensor] = None,
    rope: Optional[torch.Tensor] = None, head_dim: int = 0, tokens_per_batch: int = 0, eps: float = 1e-6,
    weight2: Optional[torch.Tensor] = None, fp8_scale: float = 0.0,
) -> torch.Tensor:
    """weight2: x is [M, 2d] = [q | k] of a fused projection; both halves are normalised (q with weight, k with weight2) in one launch.
    fp8_scale > 0: `out` (required, uint8, same shape) receives e4m3(bf16(result) / fp8_scale) - the fp8 attention's operand format."""
    _chk2d(x, "x", (bf16,))
    M, d = x.shape
    if weight2 is not None:
        if d % 2 or weight2.dtype != f32 or not weight2.is_contiguous() or weight2.numel() != d // 2:
            raise ValueError("weight2 must be contiguous f32 [d] for x [M, 2d]")
        d //= 2
    if out is None:
        out = torch.empty(tuple(x.shape), device=x.device, dtype=torch.uint8 if fp8_scale > 0 else bf16)
    _chk2d(out, "out", (torch.uint8, torch.float8_e4m3fn) if fp8_scale > 0 else (bf16,))
    if tuple(out.shape) != tuple(x.shape):
        raise ValueError("out must have the shape of x")
    if weight.dtype != f32 or not weight.is_contiguous() or weight.numel() != d:
        raise ValueError("weight must be contiguous f32 [d]")
    if rope is not None and (rope.dtype != f32 or not rope.is_contiguous()):
        raise ValueError("rope table must be contiguous f32 [tokens, head_dim/2, 2]")
    args = L.RmsNormRopeArgs(
        _ptr(x), _ptr(out), _ptr(weight), _ptr(rope), M, d, x.stride(0), out.stride(0),
        head_dim, tokens_per_batch, eps, _ptr(weight2), float(fp8_scale),
    )
    L.check(L.load().v3a_rmsnorm_rope(C.byref(args), _stream()), "v3a_rmsnorm_rope")
    return out


def rownorm_act(x: torch.Tensor, weight: torch.Tensor, *, bias: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None, mode: int = 1, act: int = L.ACT_NONE, eps: float = 1e-6) -> torch.Tensor:
    """Channel norm (+SiLU) over the last dim of a channels-last bf16 tensor (any leading dims, contiguous)."""
    if x.dtype != bf16 or not x.is_cuda or not x.is_contiguous():
        raise ValueError("x must be contiguous device bf16")
    d = x.shape[-1]
    M = x.numel() // d
    if out is None:
        out = torch.empty_like(x)
    if weight.dtype != f32 or weight.numel() != d or not weight.is_contiguous():
        raise ValueError("weight must be contiguous f32 [d]")
    args = L.RowNormArgs(_ptr(x), _ptr(out), _ptr(weight), _ptr(bias), M, d, d, d, eps, mode, act)
    L.check(L.load().v3a_rownorm_act(C.byref(args), _stream()), "v3a_rownorm_act")
    return out


def softmax_rows(s: torch.Tensor, scale: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk2d(s, "s", (f32,))
    M, N = s.shape
    if out is None:
        out = torch.empty((M, N), device=s.device, dtype=bf16)
    _chk2d(out, "out", (bf16,))
    L.check(L.load().v3a_softmax_rows(_ptr(s), _ptr(out), M, N, s.stride(0), out.stride(0), float(scale), _stream()),
            "v3a_softmax_rows")
    return out


def qknorm_rope2d(qk: torch.Tensor, C: int, qw, qb, kw, kb, cos_sin: Optional[torch.Tensor], rows_per_frame: int,
                  n_special: int, n_valid: int, wp: int, eps: float = 1e-5) -> torch.Tensor:
    _chk2d(qk, "qk", (bf16,))
    L.check(L.load().v3a_qknorm_rope2d(_ptr(qk), qk.shape[0], qk.stride(0), C, _ptr(qw), _ptr(qb), _ptr(kw), _ptr(kb),
                                       _ptr(cos_sin), rows_per_frame, n_special, n_valid, wp, eps, _stream()), "v3a_qknorm_rope2d")
    return qk


def latent_upsample_t_cl(z: torch.Tensor) -> torch.Tensor:
    """z [C,Tl,H,W] f32 contiguous -> [4(Tl-1)+1,H,W,C] bf16."""
    if z.dtype != f32 or not z.is_contiguous() or z.dim() != 4:
        raise ValueError("z must be contiguous f32 [C,Tl,H,W]")
    Cc, Tl, H, W = z.shape
    out = torch.empty(((Tl - 1) * 4 + 1, H, W, Cc), device=z.device, dtype=bf16)
    L.check(L.load().v3a_latent_upsample_t_cl(_ptr(z), _ptr(out), Cc, Tl, H, W, _stream()), "v3a_latent_upsample_t_cl")
    return out


def bilinear_cl(x: torch.Tensor, size, *, align_corners: bool, add: Optional[torch.Tensor] = None,
                table: Optional[torch.Tensor] = None, relu: bool = False, out_f32: bool = False) -> torch.Tensor:
    if x.dtype != bf16 or not x.is_contiguous() or x.dim() != 4:
        raise ValueError("x must be contiguous bf16 [T,h,w,C]")
    T, h, w, Cc = x.shape
    H, W = size
    out = torch.empty((T, H, W, Cc), device=x.device, dtype=f32 if out_f32 else bf16)
    if add is not None and (add.dtype != bf16 or not add.is_contiguous() or add.numel() != out.numel()):
        raise ValueError("add must be contiguous bf16 with the output's shape")
    if table is not None and (table.dtype != f32 or not table.is_contiguous() or table.numel() != H * W * Cc):
        raise ValueError("table must be contiguous f32 [H*W, C]")
    L.check(L.load().v3a_bilinear_cl(_ptr(x), _ptr(out), _ptr(add), _ptr(table), T, h, w, H, W, Cc, int(align_corners), int(relu),
                                     int(out_f32), _stream()), "v3a_bilinear_cl")
    return out


def depth_unproject(raw: torch.Tensor, cam: torch.Tensor, S: int, H: int, W: int):
    """raw [S*H*W, ld] f32, cam [S,16] f32 -> depth [S,H,W], conf [S,H,W], pts [S,H,W,3] (all f32)."""
    _chk2d(raw, "raw", (f32,))
    dev = raw.device
    depth = torch.empty(S, H, W, device=dev, dtype=f32)
    conf = torch.empty(S, H, W, device=dev, dtype=f32)
    pts = torch.empty(S, H, W, 3, device=dev, dtype=f32)
    L.check(L.load().v3a_depth_unproject(_ptr(raw), raw.stride(0), _ptr(cam.contiguous()), _ptr(depth), _ptr(conf), _ptr(pts), S, H, W,
                                         _stream()), "v3a_depth_unproject")
    return depth, conf, pts


def voxelize_fuse(pts: torch.Tensor, feat: torch.Tensor, nfeat: int, conf_col: int, voxel_size: float):
    """pts [M,3] f32, feat [M,ld] f32 -> dict(voxel_pts [U,3], voxel_feat [U,nfeat], keys [U,3] i32, inverse [M] i32, counts [U] i32)."""
    _chk2d(pts, "pts", (f32,))
    _chk2d(feat, "feat", (f32,))
    M, dev = pts.shape[0], pts.device
    lib = L.load()
    ws = torch.empty(int(lib.v3a_voxelize_workspace_bytes(M)), device=dev, dtype=torch.uint8)
    keys = torch.empty(M, 3, device=dev, dtype=torch.int32)
    inv = torch.empty(M, device=dev, dtype=torch.int32)
    cnt = torch.empty(M, device=dev, dtype=torch.int32)
    vp = torch.empty(M, 3, device=dev, dtype=f32)
    vf = torch.empty(M, nfeat, device=dev, dtype=f32)
    meta = torch.zeros(2, device=dev, dtype=torch.int32)
    L.check(lib.v3a_voxelize_fuse(_ptr(pts.contiguous()), _ptr(feat), feat.stride(0), nfeat, conf_col, M, float(voxel_size), _ptr(ws), ws.numel(),
                                  _ptr(keys), _ptr(inv), _ptr(cnt), _ptr(vp), _ptr(vf), nfeat, C.c_void_p(meta.data_ptr()),
                                  C.c_void_p(meta.data_ptr() + 4), _stream()), "v3a_voxelize_fuse")
    U, bad = (int(v) for v in meta.tolist())  # one host sync: the Gaussian count sizes every downstream tensor
    if bad:
        raise RuntimeError("voxel coordinate outside [-2^20, 2^20): points too far for the 63-bit key")
    return dict(voxel_pts=vp[:U], voxel_feat=vf[:U], keys=keys[:U], inverse=inv, counts=cnt[:U])


def conf_quantile_compact(conf: torch.Tensor, q: float, pts: torch.Tensor, feat: torch.Tensor, nfeat: int):
    """rows of (pts [M,3], feat[:, :nfeat]) whose confidence exceeds torch.quantile(conf, q), in row-major order ->
    dict(pts [K,3], feat [K,nfeat], threshold (0-dim f32)).  anysplat_stitched.py:381-387, 441-446."""
    _chk2d(pts, "pts", (f32,))
    _chk2d(feat, "feat", (f32,))
    if conf.dtype != f32 or not conf.is_contiguous() or conf.numel() != pts.shape[0]:
        raise ValueError("conf must be contiguous f32 with one value per row")
    M, dev = pts.shape[0], pts.device
    lib = L.load()
    ws = torch.empty(int(lib.v3a_conf_compact_workspace_bytes(M)), device=dev, dtype=torch.uint8)
    op, of = torch.empty(M, 3, device=dev, dtype=f32), torch.empty(M, nfeat, device=dev, dtype=f32)
    thr = torch.empty((), device=dev, dtype=f32)
    cnt = torch.zeros(1, device=dev, dtype=torch.int32)
    L.check(lib.v3a_conf_quantile_compact(_ptr(conf), float(q), _ptr(pts.contiguous()), _ptr(feat), feat.stride(0), nfeat, M, _ptr(ws), ws.numel(),
                                          _ptr(thr), _ptr(op), _ptr(of), nfeat, _ptr(cnt), _stream()), "v3a_conf_quantile_compact")
    K = int(cnt.item())  # one host sync: the Gaussian count sizes every downstream tensor
    return dict(pts=op[:K], feat=of[:K], threshold=thr)


def gaussian_adapter(pts: torch.Tensor, feats: torch.Tensor, sh_mask: torch.Tensor, sh_degree: int = 4, opacity_exponent: float = 1.0):
    _chk2d(feats, "feats", (f32,))
    U, dev, dsh = feats.shape[0], feats.device, (sh_degree + 1) ** 2
    e = lambda *s: torch.empty(*s, device=dev, dtype=f32)
    means, cov, sh, op, sc, rot = e(U, 3), e(U, 3, 3), e(U, 3, dsh), e(U), e(U, 3), e(U, 4)
    L.check(L.load().v3a_gaussian_adapter(_ptr(pts.contiguous()), _ptr(feats), feats.stride(0), U, sh_degree, float(opacity_exponent),
                                          _ptr(sh_mask), _ptr(means), _ptr(cov), _ptr(sh), _ptr(op), _ptr(sc), _ptr(rot), _stream()),
            "v3a_gaussian_adapter")
    return dict(means=means, covariances=cov, harmonics=sh, opacities=op, scales=sc, rotations=rot)


def linear_f32(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, act: int = L.ACT_NONE,
               residual: Optional[torch.Tensor] = None, gamma: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk2d(x, "x", (f32,))
    _chk2d(w, "w", (f32,))
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device, dtype=f32)
    if _meter is not None:
        _meter.add("linear_f32", 2.0 * M * N * K)
    L.check(L.load().v3a_linear_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(residual), _ptr(gamma), M, N, K, x.stride(0), N,
                                    residual.stride(0) if residual is not None else 0, act, _stream()), "v3a_linear_f32")
    return y


def attention_small_f32(qkv: torch.Tensor, H: int) -> torch.Tensor:
    _chk2d(qkv, "qkv", (f32,))
    S, C3 = qkv.shape
    Cc = C3 // 3
    out = torch.empty(S, Cc, device=qkv.device, dtype=f32)
    hd = Cc // H
    L.check(L.load().v3a_attention_small_f32(_ptr(qkv.contiguous()), _ptr(out), S, H, hd, hd ** -0.5, _stream()), "v3a_attention_small_f32")
    return out


def unipc_cfg_step(dit_out: torch.Tensor, tok: Optional[torch.Tensor], sample: torch.Tensor, last_sample: Optional[torch.Tensor],
                   m_prev1: Optional[torch.Tensor], m_prev2: Optional[torch.Tensor], m_out: torch.Tensor, sample_corrected: torch.Tensor,
                   prev: torch.Tensor, *, batch: int, guidance: Optional[float], coeffs: dict) -> None:
    """One launch for unpatchify + CFG + UniPC step + next-step patchify (csrc/denoise_step.hip).  `coeffs` = the host scalars of
    `UniPCMultistepScheduler.plan_step()`; latents [1, C, T, H, W] fp32, token buffers as WanDiT keeps them."""
    if sample.dtype != f32 or not sample.is_contiguous() or sample.dim() != 5 or sample.shape[0] != 1:
        raise ValueError("sample must be contiguous f32 [1, C, T, H, W]")
    _, Cc, T, H, W = sample.shape
    N = T * (H // 2) * (W // 2)
    if batch not in (1, 2) or (guidance is not None and batch != 2):
        raise ValueError("batch must be 1 or 2, and 2 (cond | uncond row blocks) when guidance is given")
    for name, t_ in (("dit_out", dit_out), ("tok", tok), ("last_sample", last_sample), ("m_prev1", m_prev1), ("m_prev2", m_prev2),
                     ("m_out", m_out), ("sample_corrected", sample_corrected), ("prev", prev)):
        if t_ is not None and t_.device != sample.device:
            raise ValueError(f"{name} is on {t_.device}, the latents on {sample.device}")
    for name, t_ in (("m_out", m_out), ("sample_corrected", sample_corrected), ("prev", prev), ("last_sample", last_sample),
                     ("m_prev1", m_prev1), ("m_prev2", m_prev2)):
        if t_ is not None and (t_.dtype != f32 or not t_.is_contiguous() or t_.numel() != sample.numel()):
            raise ValueError(f"{name} must be contiguous f32 with the latents' shape")
    for name, t_ in (("dit_out", dit_out), ("tok", tok)):
        if t_ is not None and (t_.dtype != bf16 or not t_.is_contiguous() or t_.shape != (batch * N, 4 * Cc)):
            raise ValueError(f"{name} must be contiguous bf16 [{batch * N}, {4 * Cc}]")
    c = coeffs
    a = L.UniPCStepArgs(_ptr(dit_out), _ptr(tok), _ptr(sample), _ptr(last_sample), _ptr(m_prev1), _ptr(m_prev2), _ptr(m_out),
                        _ptr(sample_corrected), _ptr(prev), Cc, T, H, W, batch, int(guidance is not None), float(guidance or 0.0), c["sigma"],
                        c["corr_order"], c["cc1"], c["cc2"], c["cc3"], c["c_rho_last"], c["c_rho0"], c["c_inv_rk"],
                        c["pred_order"], c["pc1"], c["pc2"], c["pc3"], c["p_rho0"], c["p_inv_rk"])
    L.check(L.load().v3a_unipc_cfg_step(C.byref(a), _stream()), "v3a_unipc_cfg_step")


# ------------------------------------------------------------------------------------------------ 3DGS rasteriser
def gs_project(means: torch.Tensor, covars: torch.Tensor, sh: torch.Tensor, viewmat: torch.Tensor, campos: torch.Tensor,
               K: torch.Tensor, width: int, height: int, *, sh_degree: int = 4, sh_layout: int = 1, near_plane: float = 1e-10,
               far_plane: float = 1e10, radius_clip: float = 0.1, eps2d: float = 0.3):
    """fully_fused_projection + SH colours for C cameras at once (viewmat [C,4,4], campos [C,3], K [C,3,3] -> outputs [C,U,...])
    or one camera (viewmat [4,4] -> outputs [U,...]).  sh: [U,3,K] (sh_layout=1, Gaussians.harmonics) or [U,K,3] (0)."""
    for t, n in ((means, "means"), (covars, "covars"), (sh, "sh"), (viewmat, "viewmat"), (campos, "campos"), (K, "K")):
        if t.dtype != f32 or not t.is_cuda or not t.is_contiguous():
            raise ValueError(f"{n}: contiguous fp32 device tensor required")
    single = viewmat.dim() == 2
    Cn = 1 if single else viewmat.shape[0]
    if campos.numel() != 3 * Cn or K.numel() != 9 * Cn or viewmat.numel() != 16 * Cn:
        raise ValueError("viewmat / campos / K disagree on the number of cameras")
    U, dev = means.shape[0], means.device
    sh_k = sh.shape[2] if sh_layout == 1 else sh.shape[1]
    lead = () if single else (Cn,)
    radii = torch.empty(*lead, U, device=dev, dtype=torch.int32)
    e = lambda *s: torch.empty(*lead, *s, device=dev, dtype=f32)
    m2, dep, con, col = e(U, 2), e(U), e(U, 3), e(U, 4)
    a = L.GsProjectArgs(_ptr(means), _ptr(covars), _ptr(sh), sh_layout, sh_k, sh_degree, _ptr(viewmat), _ptr(campos), _ptr(K), U, Cn,
                        width, height, near_plane, far_plane, radius_clip, eps2d, _ptr(radii), _ptr(m2), _ptr(dep), _ptr(con), _ptr(col))
    L.check(L.load().v3a_gs_project(C.byref(a), _stream()), "v3a_gs_project")
    return dict(radii=radii, means2d=m2, depths=dep, conics=con, colors=col)


class GsWorkspace:
    """Grow-only scratch for v3a_gs_rasterize, shared by the camera batches of one video."""

    def __init__(self):
        self.buf, self.cap, self.key = None, 0, None

    def get(self, U, Cn, width, height, need):
        key = (U, Cn, width, height)
        if self.buf is None or self.key != key or need > self.cap:
            self.cap = max(int(need * 1.25), 1 << 20)
            nbytes = L.load().v3a_gs_rasterize_workspace_bytes(U, Cn, width, height, self.cap)
            if nbytes < 0:
                L.check(int(nbytes), "v3a_gs_rasterize_workspace_bytes")
            self.buf, self.key = None, key
            self.buf = torch.empty(nbytes, device="cuda", dtype=torch.uint8)
        return self.buf


def gs_rasterize(proj: dict, opacities: torch.Tensor, width: int, height: int, *, background: Optional[torch.Tensor] = None,
                 clamp_rgb: bool = True, workspace: Optional[GsWorkspace] = None, return_order: bool = False):
    """isect_tiles + sort + rasterize_to_pixels for the cameras of `proj` -> dict(color [C,H,W,3], depth [C,H,W], alpha [C,H,W],
    n_isect); without the C axis when `proj` came from a single-camera gs_project call."""
    U, dev = opacities.shape[0], opacities.device
    if opacities.dtype != f32 or not opacities.is_contiguous():
        raise ValueError("opacities: contiguous fp32 [U] required")
    if U == 0:
        raise ValueError("no Gaussians to draw")
    single = proj["radii"].dim() == 1
    Cn = 1 if single else proj["radii"].shape[0]
    lead = () if single else (Cn,)
    wsp = workspace or GsWorkspace()
    color = torch.empty(*lead, height, width, 3, device=dev, dtype=f32)
    depth = torch.empty(*lead, height, width, device=dev, dtype=f32)
    alpha = torch.empty(*lead, height, width, device=dev, dtype=f32)
    ntiles = ((width + 15) // 16) * ((height + 15) // 16)
    n_host = C.c_long(0)
    need = max(wsp.cap, 1)
    for _ in range(2):
        buf = wsp.get(U, Cn, width, height, need)
        offs = torch.empty(Cn * ntiles + 1, device=dev, dtype=torch.int32) if return_order else None
        ids = torch.empty(wsp.cap, device=dev, dtype=torch.int32) if return_order else None
        a = L.GsRasterizeArgs(_ptr(proj["radii"]), _ptr(proj["means2d"]), _ptr(proj["depths"]), _ptr(proj["conics"]), _ptr(proj["colors"]),
                              _ptr(opacities), _ptr(background) if background is not None else None, U, Cn, width, height, int(clamp_rgb),
                              _ptr(color), _ptr(depth), _ptr(alpha), _ptr(buf), buf.numel(), wsp.cap, C.pointer(n_host),
                              _ptr(offs) if return_order else None, _ptr(ids) if return_order else None)
        rc = L.load().v3a_gs_rasterize(C.byref(a), _stream())
        if rc == -4:  # V3A_ERR_WORKSPACE: the count is known now, grow once and redo
            need = n_host.value
            continue
        L.check(rc, "v3a_gs_rasterize")
        break
    else:
        raise RuntimeError("v3a_gs_rasterize: workspace still too small after growing")
    out = dict(color=color, depth=depth, alpha=alpha, n_isect=int(n_host.value))
    if return_order:
        out["tile_offsets"], out["flatten_ids"] = offs, ids[: n_host.value]
    return out
