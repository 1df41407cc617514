"""Tensor-level wrappers over the C-ABI (include/pbe_hip.h).

torch is used for device memory and the current HIP stream only; every arithmetic step is a
kernel of libpbe_hip.so.  All wrappers raise if handed CPU tensors — there is no fallback.
Activations are fp16; ``[B, H, W, C]`` / ``[rows, C]`` (channels last, contiguous).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import lib as _l

ACT_NONE, ACT_SILU, ACT_GELU, ACT_QUICK_GELU, ACT_GEGLU = 0, 1, 2, 3, 4
_ws = {}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, what: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _l.PbeError(f"{what}: expected a tensor on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise _l.PbeError(f"{what}: expected dtype {dtype}, got {t.dtype}")
    return t


def _h(t, what):
    return _req(t, torch.float16, what)


def _f(t, what):
    return _req(t, torch.float32, what)


SPLITK_WS_BYTES = 64 << 20

# ---- per-shape tile choice measured on MI355X (tools/autotune.py writes the table) -------------
import json as _json
import os as _os

_TUNED_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tuned_mi355x.json")
try:
    if _os.environ.get("PBE_NO_TUNED"):          # evaluate the built-in heuristic (tools / tests only)
        raise OSError("tuned table disabled by PBE_NO_TUNED")
    with open(_TUNED_PATH) as _tuned_file:
        _TUNED = _json.load(_tuned_file)
except (OSError, ValueError):
    _TUNED = {}
_RECORD = None            # set to a dict by tools/autotune.py to collect the shapes a workload uses
_PLANS = None             # set to a list by tools/layer_diff.py: (key, tile config, split-K factor, BM, BN, workgroups) per GEMM / conv launch
_PIN_SCALE = 1            # see pinned_batch_scale
_PIN_CACHE = {}
_PIN_MISSES = []         # (key, tile, split-K planned at the scaled batch, tile, split-K the sub-batch launch takes instead)


class pinned_batch_scale:
    """Inside this context every GEMM / conv is launched with the tile config and split-K factor the library would pick for
    the SAME layer at `scale` times the batch.  The k order of a tile does not depend on the tile shape; only the split-K
    factor changes the fp32 summation order, so a sub-batch evaluated this way reproduces the bits of the full batch
    (used by the shared guidance prefix: B samples evaluated once must equal the 2B duplicated evaluation)."""

    def __init__(self, scale: int):
        self.scale = int(scale)

    def __enter__(self):
        global _PIN_SCALE
        self.prev, _PIN_SCALE = _PIN_SCALE, self.scale

    def __exit__(self, *exc):
        global _PIN_SCALE
        _PIN_SCALE = self.prev
        return False


def _plan_of(desc, conv: bool):
    out, need = (C.c_int32 * 6)(), C.c_size_t()
    lib = _l.load()
    _l.check((lib.pbe_conv3x3_plan if conv else lib.pbe_gemm_plan)(C.byref(desc), out, C.byref(need)), "plan")
    return list(out), int(need.value)


def _pinned_cfg(desc, key_fn, conv: bool, field: str = "M") -> int:
    """tile_cfg for `desc` under pinned_batch_scale: the plan of the scaled problem, cached per (shape key + every other input of
    the plan: fused residual / output alignment, second source, workspace, forced tile).  If the launch cannot take the pinned
    (tile, split-K) pair - the library falls back to its heuristic, which changes the fp32 summation order - the miss is recorded
    in _PIN_MISSES and warned about once per shape (the result is still correct, only not bit-identical to the full batch)."""
    key = key_fn(_PIN_SCALE)
    ck = (key, bool(desc.resid), int(getattr(desc, "ldc", 0)) & 3, int(getattr(desc, "ldr", 0)) & 3, bool(getattr(desc, "A2", None) or getattr(desc, "X2", None)),
          bool(desc.workspace), int(desc.workspace_bytes), _FORCE_CFG)
    hit = _PIN_CACHE.get(ck)
    if hit is None:
        big = type(desc).from_buffer_copy(desc)
        if conv:
            big.B = desc.B * _PIN_SCALE
        else:
            setattr(big, field, getattr(desc, field) * _PIN_SCALE)
            if field == "M":                     # extended epilogue: the scaled problem's statistics planes are as long as its rows
                if big.ln_stats:
                    big.ln_stats_ld = max(big.ln_stats_ld, big.M)
                if big.row_stats_out:
                    big.row_stats_ld = max(big.row_stats_ld, big.M)
        big.tile_cfg = int(_FORCE_CFG) if _FORCE_CFG is not None else int(_TUNED.get(key, -1))
        pl, _ = _plan_of(big, conv)
        hit = pl[0] | (max(1, pl[1]) << 8)
        small = type(desc).from_buffer_copy(desc)
        small.tile_cfg = hit
        got, _ = _plan_of(small, conv)
        if got[1] != max(1, pl[1]):
            import warnings
            _PIN_MISSES.append((key, pl[0], pl[1], got[0], got[1]))
            warnings.warn(f"pinned_batch_scale({_PIN_SCALE}): {key} plans tile {pl[0]} / split-K {pl[1]} at the scaled batch but the "
                          f"sub-batch launch runs tile {got[0]} / split-K {got[1]} (different fp32 summation order)")
        _PIN_CACHE[ck] = hit
    return hit


def _launch_note(desc, key: str, conv: bool):
    if _PLANS is not None:
        pl, _ = _plan_of(desc, conv)
        _PLANS.append((key, *pl[:5]))
_TIMES = None             # set to a dict by tools/shape_profile.py: key -> [(start event, end event), ...] around each launch


class _timed:
    """Bracket one launch with events when shape profiling is on (no-op otherwise)."""

    def __init__(self, key):
        self.key = key

    def __enter__(self):
        if _TIMES is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if _TIMES is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _TIMES.setdefault(self.key, []).append((self.e0, e1))
        return False


_FORCE_CFG = None         # set by tools/autotune_insitu.py: every GEMM / conv launch takes this tile_cfg (A/B inside the real sampler)


def _tile_cfg(key: str) -> int:
    if _RECORD is not None:
        _RECORD[key] = _RECORD.get(key, 0) + 1
    if _FORCE_CFG is not None:
        return int(_FORCE_CFG)
    return int(_TUNED.get(key, -1))


def _splitk_ws(device):
    """Per-device scratch for split-K partial sums (fp32 slabs); ops on one stream run in order, so one buffer is enough."""
    key = (device.index, "splitk")
    ws = _ws.get(key)
    if ws is None:
        ws = torch.empty(SPLITK_WS_BYTES, dtype=torch.uint8, device=device)
        _ws[key] = ws
    return ws


def _rows(t: torch.Tensor, what: str) -> Tuple[int, int, int]:
    """(rows, cols, ld) of a 2-D view whose last dim is contiguous."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise _l.PbeError(f"{what}: expected a 2-D tensor with unit inner stride, got {tuple(t.shape)} / {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


# ---------------------------------------------------------------------------------------------
class RowStats:
    """Partial (sum, sum of squares) of every row of a [M, C] tensor: `buf` fp32 [parts, ld, 2], as a GEMM epilogue (row_stats=True)
    or pbe_row_stats_f16 wrote them; what a LayerNorm-folding GEMM consumes (ln=...)."""
    __slots__ = ("buf", "parts", "ld", "row0")

    def __init__(self, buf, parts, ld, row0=0):
        self.buf, self.parts, self.ld, self.row0 = buf, parts, ld, row0

    def ptr(self) -> int:
        return self.buf.data_ptr() + 8 * self.row0


def row_stats(x: torch.Tensor) -> RowStats:
    """One-partial row statistics of a contiguous-row fp16 [M, C] tensor (the fallback when x's producer did not emit them)."""
    _h(x, "row_stats x")
    M, Cc, ldx = _rows(x, "row_stats x")
    buf = torch.empty((1, M, 2), dtype=torch.float32, device=x.device)
    with _timed(f"rs:{M}:{Cc}"):
        _l.check(_l.load().pbe_row_stats_f16(_p(x), _p(buf), M, Cc, ldx, _stream()), "pbe_row_stats_f16")
    return RowStats(buf, 1, M)


_EX_PARTS = {}


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, a2: Optional[torch.Tensor] = None,
         rowvec: Optional[torch.Tensor] = None, group_rows: int = 0, resid: Optional[torch.Tensor] = None, act: int = ACT_NONE,
         alpha: float = 1.0, bias_per_row: bool = False, out: Optional[torch.Tensor] = None, alpha_cols: int = 0, ln=None,
         row_stats=False, vt: Optional[torch.Tensor] = None, vt_col0: int = 0, vt_tokens: int = 0):
    """out[m, n] = act(alpha * sum_k [a | a2][m, k] w[n, k] + bias + rowvec[m // group_rows, n]) + resid[m, n].
    2-D operands, or 3-D [batch, rows, cols] for a strided batch (w may have batch 1).

    Extended epilogue (2-D operands; include/pbe_hip.h, pbe_gemm_desc): alpha_cols = alpha on columns < alpha_cols only;
    ln = (RowStats of a's rows, colsum fp32 [N], eps): LayerNorm(a) is folded in (w = W * gamma, bias = W beta + b);
    row_stats = True (or a RowStats to fill, e.g. a slice of a shared buffer): also returns the RowStats of the output rows -> (out, stats);
    vt [B, N - vt_col0, >= vt_tokens]: columns >= vt_col0 are written transposed there (V^T), `out` then has vt_col0 columns."""
    _h(a, "gemm a"); _h(w, "gemm w")
    batch = 1
    sA = sW = sC = sR = 0
    if a.dim() == 3:
        batch = a.shape[0]
        if a.stride(2) != 1 or w.dim() != 3 or w.stride(2) != 1 or w.shape[0] not in (1, batch):
            raise _l.PbeError("gemm: bad batched operands")
        M, K, lda, sA = a.shape[1], a.shape[2], a.stride(1), a.stride(0)
        N, Kw, ldw = w.shape[1], w.shape[2], w.stride(1)
        sW = w.stride(0) if w.shape[0] == batch else 0
        if out is None:
            out = torch.empty((batch, M, N), dtype=torch.float16, device=a.device)
        ldc, sC = out.stride(1), out.stride(0)
        ldr = 0
        if resid is not None:
            _h(resid, "gemm resid")
            ldr = resid.stride(-2)
            sR = resid.stride(0) if (resid.dim() == 3 and resid.shape[0] == batch) else 0
    else:
        M, K, lda = _rows(a, "gemm a")
        N, Kw, ldw = _rows(w, "gemm w")
        if out is None:
            out = torch.empty((M, vt_col0 if vt is not None else (N // 2 if act == ACT_GEGLU else N)), dtype=torch.float16, device=a.device)
        ldc = _rows(out, "gemm out")[2]
        ldr = 0
        if resid is not None:
            ldr = _rows(_h(resid, "gemm resid"), "gemm resid")[2]
    K1, lda2 = K, 0
    if a2 is not None:
        _, K2, lda2 = _rows(_h(a2, "gemm a2"), "gemm a2")
        K = K1 + K2
    if Kw != K:
        raise _l.PbeError(f"gemm: K mismatch, activations {K} vs weights {Kw}")
    if bias is not None:
        _f(bias, "gemm bias")
        if bias.numel() != (M if bias_per_row else N):
            raise _l.PbeError("gemm: bias length mismatch")
    ldv = 0
    if rowvec is not None:
        _h(rowvec, "gemm rowvec")
        ldv = rowvec.stride(0) if rowvec.dim() == 2 else 0
        if group_rows <= 0:
            raise _l.PbeError("gemm: rowvec needs group_rows")
    ex = bool(alpha_cols or ln is not None or row_stats is not False or vt is not None)
    kp = "gx" if ex else "g"                   # the extended-epilogue tiles are tuned under their own keys
    d = _l.GemmDesc(_p(a), _p(a2), _p(w), _p(_h(out, "gemm out")), _p(bias), _p(rowvec), _p(resid), M, N, K, K1, lda, lda2, ldw, ldc, ldr,
                    ldv, group_rows, sA, sW, sC, sR, batch, float(alpha), act, 1 if bias_per_row else 0,
                    _splitk_ws(a.device).data_ptr(), SPLITK_WS_BYTES, _tile_cfg(f"{kp}:{M}:{N}:{K}:{batch}"))
    stats = None
    if ex:
        if a.dim() != 2:
            raise _l.PbeError("gemm: the extended epilogue takes 2-D operands")
        d.alpha_cols = int(alpha_cols)
        if ln is not None:
            st, colsum, eps = ln
            _f(colsum, "gemm ln colsum")
            if colsum.numel() != N or st.ld - st.row0 < M:
                raise _l.PbeError("gemm: LayerNorm fold needs colsum [N] and row statistics for every row of a")
            d.ln_stats, d.ln_parts, d.ln_stats_ld, d.ln_colsum, d.ln_eps = st.ptr(), st.parts, st.ld, _p(colsum), float(eps)
        if vt is not None:
            _h(vt, "gemm vt")
            if vt.dim() != 3 or vt.stride(2) != 1 or vt.shape[1] != N - vt_col0 or vt.shape[0] * vt_tokens != M:
                raise _l.PbeError(f"gemm: vt must be [M / vt_tokens, N - vt_col0, >= vt_tokens], got {tuple(vt.shape)}")
            d.VT, d.vt_col0, d.vt_tokens, d.vt_bs, d.vt_rs = _p(vt), int(vt_col0), int(vt_tokens), vt.stride(0), vt.stride(1)
    if _PIN_SCALE != 1:
        if a.dim() == 3:                        # per-sample strided batch (V^T projection): the batch count scales, not M
            d.tile_cfg = _pinned_cfg(d, lambda sc: f"g:{M}:{N}:{K}:{batch * sc}", False, "batch")
        else:
            d.tile_cfg = _pinned_cfg(d, lambda sc: f"{kp}:{M * sc}:{N}:{K}:1", False)
    if row_stats is not False:                  # one partial per column tile of the plan this launch will take
        ck = (M, N, K, int(d.tile_cfg), bool(resid is not None), bool(vt is not None))
        parts = _EX_PARTS.get(ck)
        if parts is None:
            d.row_stats_out = 8                 # (any non-null value: the plan only needs to know the form)
            pl, _ = _plan_of(d, False)
            parts = _EX_PARTS[ck] = pl[5]
        if isinstance(row_stats, RowStats):
            stats = row_stats
            if stats.parts < parts:
                raise _l.PbeError(f"gemm: row_stats buffer holds {stats.parts} partials, the plan writes {parts}")
            stats = RowStats(stats.buf, parts, stats.ld, stats.row0)
        else:
            stats = RowStats(torch.empty((parts, M, 2), dtype=torch.float32, device=a.device), parts, M)
        d.row_stats_out, d.row_stats_ld = stats.ptr(), stats.ld
    _launch_note(d, f"{kp}:{M}:{N}:{K}:{batch}", False)
    with _timed(f"{kp}:{M}:{N}:{K}:{batch}|a{act}{'r' if resid is not None else ''}{'v' if rowvec is not None else ''}{'L' if ln is not None else ''}{'S' if stats is not None else ''}{'T' if vt is not None else ''}"):
        _l.check(_l.load().pbe_gemm_f16(C.byref(d), _stream()), "pbe_gemm_f16")
    return (out, stats) if row_stats is not False else out


def conv_out_hw(h: int, w: int, stride: int, pad: int, upsample: bool) -> Tuple[int, int]:
    hv, wv = (h * 2, w * 2) if upsample else (h, w)
    extra = 2 if pad else 1
    return (hv + extra - 3) // stride + 1, (wv + extra - 3) // stride + 1


def conv3x3(x: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], *, x2: Optional[torch.Tensor] = None,
            rowvec: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, stride: int = 1, pad: int = 1,
            upsample: bool = False, act: int = ACT_NONE, group_stats: int = 0) -> torch.Tensor:
    """NHWC 3x3 conv on the matrix cores; ``wp`` is the packed [Cout, 9*Cin] weight (pack_conv3x3).
    group_stats = G > 0: the output feeds a GroupNorm(G) - where the planned tile can, the copy-out also leaves the norm's partial
    statistics (include/pbe_hip.h, group_stats_out) and the returned tensor carries them (`_pbe_gstats`); ops.groupnorm then runs its
    normalisation pass only."""
    _h(x, "conv3x3 x"); _h(wp, "conv3x3 w")
    if x.dim() != 4 or not x.is_contiguous():
        raise _l.PbeError("conv3x3: x must be a contiguous [B,H,W,C] tensor")
    B, H, W, C1 = x.shape
    C2 = 0
    if wp.dim() == 3:                            # pack_conv3x3_up_phases: the upsampling conv as four 2x2 convs on the source grid (4 / 9 of the MACs)
        if not upsample or x2 is not None or resid is not None or rowvec is not None or stride != 1 or pad != 1:
            raise _l.PbeError("conv3x3: phase-packed weights are for the plain nearest-2x upsampling conv")
        if wp.shape[0] != 4 or not wp.is_contiguous() or wp.shape[2] != 4 * C1:
            raise _l.PbeError(f"conv3x3: phase-packed weight must be [4, Cout, {4 * C1}], got {tuple(wp.shape)}")
        Cout = wp.shape[1]
        y = torch.empty((B, 2 * H, 2 * W, Cout), dtype=torch.float16, device=x.device)
        if bias is not None:
            _f(bias, "conv3x3 bias")
        key = f"c:{B}:{H}:{W}:{C1}:0:{Cout}:1:1:2"
        d = _l.Conv3x3Desc(_p(x), None, _p(wp), _p(y), _p(bias), None, None, B, H, W, C1, 0, Cout, 1, 1, 2, 0, act, _splitk_ws(x.device).data_ptr(),
                           SPLITK_WS_BYTES, _tile_cfg(key), conv_kblock(C1, 0))
        if _PIN_SCALE != 1:
            d.tile_cfg = _pinned_cfg(d, lambda sc: f"c:{B * sc}:{H}:{W}:{C1}:0:{Cout}:1:1:2", True)
        _launch_note(d, key, True)
        with _timed(key):
            _l.check(_l.load().pbe_conv3x3_f16(C.byref(d), _stream()), "pbe_conv3x3_f16 (upsample, phase form)")
        return y
    if x2 is not None:
        _h(x2, "conv3x3 x2")
        if x2.dim() != 4 or not x2.is_contiguous() or x2.shape[:3] != x.shape[:3]:
            raise _l.PbeError("conv3x3: x2 must match x in [B,H,W]")
        C2 = x2.shape[3]
    Cout = wp.shape[0]
    if wp.dim() != 2 or not wp.is_contiguous() or wp.shape[1] != 9 * (C1 + C2):
        raise _l.PbeError(f"conv3x3: packed weight must be [Cout, {9 * (C1 + C2)}], got {tuple(wp.shape)}")
    Ho, Wo = conv_out_hw(H, W, stride, pad, upsample)
    y = torch.empty((B, Ho, Wo, Cout), dtype=torch.float16, device=x.device)
    ldv = 0
    if rowvec is not None:
        _h(rowvec, "conv3x3 rowvec")
        if rowvec.dim() != 2 or rowvec.shape[0] != B or rowvec.stride(1) != 1:
            raise _l.PbeError("conv3x3: rowvec must be [B, >=Cout] with unit inner stride")
        ldv = rowvec.stride(0)
    if resid is not None:
        _h(resid, "conv3x3 resid")
        if tuple(resid.shape) != tuple(y.shape) or not resid.is_contiguous():
            raise _l.PbeError("conv3x3: resid must match the output shape")
    if bias is not None:
        _f(bias, "conv3x3 bias")
    d = _l.Conv3x3Desc(_p(x), _p(x2), _p(wp), _p(y), _p(bias), _p(rowvec), _p(resid), B, H, W, C1, C2, Cout, stride, pad,
                       1 if upsample else 0, ldv, act, _splitk_ws(x.device).data_ptr(), SPLITK_WS_BYTES,
                       _tile_cfg(f"c:{B}:{H}:{W}:{C1}:{C2}:{Cout}:{stride}:{pad}:{int(bool(upsample))}"), conv_kblock(C1, C2), None, 0, None)
    gs_buf = gs_blocks = None
    if group_stats > 0 and USE_CONV_GROUP_STATS and Cout % group_stats == 0 and Ho * Wo >= 64:
        gs_buf = torch.empty((B, (Ho * Wo) // 64, group_stats, 2), dtype=torch.float32, device=x.device)       # room for the smallest row block (64)
        gs_blocks = C.c_int32(0)
        d.group_stats_out, d.group_stats_groups, d.group_stats_blocks = gs_buf.data_ptr(), group_stats, C.cast(C.pointer(gs_blocks), C.c_void_p)
    if _PIN_SCALE != 1:
        d.tile_cfg = _pinned_cfg(d, lambda sc: f"c:{B * sc}:{H}:{W}:{C1}:{C2}:{Cout}:{stride}:{pad}:{int(bool(upsample))}", True)
    _launch_note(d, f"c:{B}:{H}:{W}:{C1}:{C2}:{Cout}:{stride}:{pad}:{int(bool(upsample))}", True)
    with _timed(f"c:{B}:{H}:{W}:{C1}:{C2}:{Cout}:{stride}:{pad}:{int(bool(upsample))}"):
        _l.check(_l.load().pbe_conv3x3_f16(C.byref(d), _stream()), "pbe_conv3x3_f16")
    if gs_blocks is not None and gs_blocks.value > 0:
        y._pbe_gstats = GroupStats(gs_buf, B, int(gs_blocks.value), group_stats, y._version)
    return y


def conv3x3_small(x: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], *, stride: int = 1, pad: int = 1,
                  act: int = ACT_NONE, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3 conv for tiny Cin (channel-padded NHWC input, Cp in {8,16}): im2col + GEMM."""
    _h(x, "conv3x3_small x")
    B, H, W, Cp = x.shape
    Ho, Wo = conv_out_hw(H, W, stride, pad, False)
    cols = torch.empty((B * Ho * Wo, 9 * Cp), dtype=torch.float16, device=x.device)
    _l.check(_l.load().pbe_im2col3x3_f16(_p(x), _p(cols), B, H, W, Cp, stride, pad, _stream()), "pbe_im2col3x3_f16")
    y = gemm(cols, wp, bias, act=act, resid=None if resid is None else resid.reshape(B * Ho * Wo, -1))
    return y.view(B, Ho, Wo, wp.shape[0])


USE_CONV_GROUP_STATS = True      # conv3x3(group_stats=G): let the conv's copy-out produce the following GroupNorm's statistics (tools flip it for A/B)


class GroupStats:
    """Partial (sum, sumsq) per (sample, row block, group) of a conv output, written by the conv's copy-out: the first B * blocks * groups * 2
    floats of `buf`, laid out [B][blocks][groups][2] (pbe_groupnorm_f16's partial layout)."""
    __slots__ = ("buf", "blocks", "groups", "batch", "version")

    def __init__(self, buf, batch, blocks, groups, version):
        self.buf, self.batch, self.blocks, self.groups, self.version = buf, batch, blocks, groups, version      # version: the tensor's _version when written

    def view(self):
        return self.buf.view(-1)[: self.batch * self.blocks * self.groups * 2].view(self.batch, self.blocks, self.groups, 2)


def groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, silu: bool, *, x2: Optional[torch.Tensor] = None,
              groups: int = 32) -> torch.Tensor:
    """GroupNorm(+SiLU) of NHWC x (optionally the channel concat x | x2); output has C1+C2 channels."""
    _h(x, "groupnorm x"); _f(gamma, "groupnorm gamma"); _f(beta, "groupnorm beta")
    if not x.is_contiguous():
        raise _l.PbeError("groupnorm: x must be contiguous")
    B, C1 = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C1)
    C2 = 0
    if x2 is not None:
        _h(x2, "groupnorm x2")
        C2 = x2.shape[-1]
        if not x2.is_contiguous() or x2.numel() != B * HW * C2:
            raise _l.PbeError("groupnorm: x2 shape mismatch")
    if gamma.numel() != C1 + C2 or beta.numel() != C1 + C2:
        raise _l.PbeError("groupnorm: affine length mismatch")
    lib = _l.load()
    st = getattr(x, "_pbe_gstats", None)
    if st is not None and x2 is None and st.groups == groups and st.batch == B and st.version == x._version:      # (an in-place edit since the conv voids them)
        # the producing conv left this tensor's statistics: normalisation pass only (one read, one write)
        y = torch.empty_like(x)
        with _timed(f"n:{B}:{HW}:{C1}:0"):
            _l.check(lib.pbe_groupnorm_apply_f16(_p(x), _p(st.buf), st.blocks, _p(gamma), _p(beta), _p(y), B, HW, C1, groups, float(eps), 1 if silu else 0,
                                                 _stream()), "pbe_groupnorm_apply_f16")
        return y
    need = lib.pbe_groupnorm_workspace_bytes(B, HW)
    key = (x.device.index, "gn")
    ws = _ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 21), dtype=torch.uint8, device=x.device)
        _ws[key] = ws
    y = torch.empty(tuple(x.shape[:-1]) + (C1 + C2,), dtype=torch.float16, device=x.device)
    with _timed(f"n:{B}:{HW}:{C1}:{C2}"):
        _l.check(lib.pbe_groupnorm_f16(_p(x), _p(x2), _p(gamma), _p(beta), _p(y), B, HW, C1, C2, groups, float(eps), 1 if silu else 0,
                                       _p(ws), ws.numel(), _stream()), "pbe_groupnorm_f16")
    return y


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    _h(x, "layernorm x"); _f(gamma, "layernorm gamma"); _f(beta, "layernorm beta")
    x2 = x.reshape(-1, x.shape[-1])
    rows, Cc, ldx = _rows(x2, "layernorm x")
    y = torch.empty((rows, Cc), dtype=torch.float16, device=x.device)
    with _timed(f"l:{rows}:{Cc}"):
        _l.check(_l.load().pbe_layernorm_f16(_p(x2), _p(gamma), _p(beta), _p(y), rows, Cc, ldx, Cc, float(eps), _stream()), "pbe_layernorm_f16")
    return y.view(x.shape)


def layernorm_f8(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5):
    """LayerNorm with an fp8 (OCP e4m3) output: returns (y8 uint8 [rows, C], row_scale fp32 [rows]); y = y8 * row_scale[:, None]."""
    _h(x, "layernorm_f8 x"); _f(gamma, "layernorm_f8 gamma"); _f(beta, "layernorm_f8 beta")
    x2 = x.reshape(-1, x.shape[-1])
    rows, Cc, ldx = _rows(x2, "layernorm_f8 x")
    y = torch.empty((rows, Cc), dtype=torch.uint8, device=x.device)
    sc = torch.empty((rows,), dtype=torch.float32, device=x.device)
    with _timed(f"l8:{rows}:{Cc}"):
        _l.check(_l.load().pbe_layernorm_f8(_p(x2), _p(gamma), _p(beta), _p(y), _p(sc), rows, Cc, ldx, Cc, float(eps), _stream()), "pbe_layernorm_f8")
    return y, sc


def gemm_f8(a8: torch.Tensor, a_scale: torch.Tensor, w8: torch.Tensor, w_scale: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
            resid: Optional[torch.Tensor] = None, act: int = ACT_NONE, alpha: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[m, n] = act(alpha * a_scale[m] * w_scale[n] * sum_k a8[m, k] w8[n, k] + bias) + resid with OCP e4m3 operands (uint8
    tensors) and fp16 output.  2-D operands, or 3-D [batch, rows, K] for a strided batch (w8 / w_scale may have batch 1)."""
    _req(a8, torch.uint8, "gemm_f8 a8"); _req(w8, torch.uint8, "gemm_f8 w8"); _f(a_scale, "gemm_f8 a_scale"); _f(w_scale, "gemm_f8 w_scale")
    batch = 1
    sA = sW = sC = 0
    ssa = ssw = 0
    if a8.dim() == 3:
        batch = a8.shape[0] if a8.stride(0) != 0 else max(a8.shape[0], w8.shape[0])
        if a8.stride(2) != 1 or w8.dim() != 3 or w8.stride(2) != 1:
            raise _l.PbeError("gemm_f8: bad batched operands")
        M, K, lda, sA = a8.shape[1], a8.shape[2], a8.stride(1), a8.stride(0)
        N, Kw, ldw, sW = w8.shape[1], w8.shape[2], w8.stride(1), w8.stride(0)
        ssa = a_scale.stride(0) if a_scale.dim() == 2 else 0
        ssw = w_scale.stride(0) if w_scale.dim() == 2 else 0
        if out is None:
            out = torch.empty((batch, M, N), dtype=torch.float16, device=a8.device)
        ldc, sC = out.stride(1), out.stride(0)
    else:
        M, K, lda = _rows(a8, "gemm_f8 a8")
        N, Kw, ldw = _rows(w8, "gemm_f8 w8")
        if out is None:
            out = torch.empty((M, N // 2 if act == ACT_GEGLU else N), dtype=torch.float16, device=a8.device)
        ldc = _rows(out, "gemm_f8 out")[2]
    if Kw != K:
        raise _l.PbeError(f"gemm_f8: K mismatch, activations {K} vs weights {Kw}")
    if a_scale.numel() < M or w_scale.numel() < N:
        raise _l.PbeError("gemm_f8: scale vectors too short")
    ldr = 0
    if resid is not None:
        ldr = _rows(_h(resid, "gemm_f8 resid"), "gemm_f8 resid")[2]
    if bias is not None:
        _f(bias, "gemm_f8 bias")
    d = _l.GemmDesc(_p(a8), None, _p(w8), _p(_h(out, "gemm_f8 out")), _p(bias), None, _p(resid), M, N, K, K, lda, 0, ldw, ldc, ldr,
                    0, 0, sA, sW, sC, 0, batch, float(alpha), act, 0, None, 0, _tile_cfg(f"g8:{M}:{N}:{K}:{batch}"),
                    _p(a_scale), _p(w_scale), ssa, ssw, 1)
    _launch_note(d, f"g8:{M}:{N}:{K}:{batch}", False)
    with _timed(f"g8:{M}:{N}:{K}:{batch}|a{act}{'r' if resid is not None else ''}"):
        _l.check(_l.load().pbe_gemm_f16(C.byref(d), _stream()), "pbe_gemm_f16 (fp8 operands)")
    return out


def pack_linear_f8(w: torch.Tensor):
    """[N, K] fp32 weight -> (OCP e4m3 bytes uint8 [N, K], fp32 scale [N]) with one scale per output channel: w ~ w8 * scale[:, None].
    One-off, at pack time (BASELINE configs[4])."""
    w2 = w.detach().reshape(w.shape[0], -1).float()
    amax = w2.abs().amax(1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    w8 = (w2 / scale[:, None]).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
    return w8, scale.contiguous()


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, B: int, H: int, Nq: int, Nk: int, D: int, scale: float, *,
              q_strides: Tuple[int, int], k_strides: Tuple[int, int], vt_strides: Tuple[int, int],
              out: Optional[torch.Tensor] = None, q_prescaled: bool = False) -> torch.Tensor:
    """softmax(q k^T scale) v -> [B, Nq, H*D].  q/k: element (b,n,h,d) at b*bs + n*rs + h*D + d of the given
    (possibly sliced) tensors; vt: element (b,h,d,n) at b*bs + (h*D+d)*rs + n.  strides = (bs, rs) in elements."""
    _h(q, "attention q"); _h(k, "attention k"); _h(vt, "attention vt")
    if out is None:
        out = torch.empty((B, Nq, H * D), dtype=torch.float16, device=q.device)
    d = _l.AttnDesc(_p(q), _p(k), _p(vt), _p(out), B, H, Nq, Nk, D, q_strides[0], q_strides[1], k_strides[0], k_strides[1],
                    vt_strides[0], vt_strides[1], out.stride(0), out.stride(1), float(scale), 1 if q_prescaled else 0)
    with _timed(f"a:{B}:{H}:{Nq}:{Nk}:{D}"):
        _l.check(_l.load().pbe_attention_f16(C.byref(d), _stream()), "pbe_attention_f16")
    return out


def softmax_rows(x: torch.Tensor, scale: float) -> torch.Tensor:
    _h(x, "softmax_rows x")
    x2 = x.reshape(-1, x.shape[-1])
    rows, cols, ldx = _rows(x2, "softmax_rows x")
    y = torch.empty((rows, cols), dtype=torch.float16, device=x.device)
    _l.check(_l.load().pbe_softmax_rows_f16(_p(x2), _p(y), rows, cols, ldx, cols, float(scale), _stream()), "pbe_softmax_rows_f16")
    return y.view(x.shape)


def geglu(h: torch.Tensor) -> torch.Tensor:
    _h(h, "geglu h")
    if not h.is_contiguous():
        raise _l.PbeError("geglu: h must be contiguous")
    F = h.shape[-1] // 2
    M = h.numel() // (2 * F)
    y = torch.empty(tuple(h.shape[:-1]) + (F,), dtype=torch.float16, device=h.device)
    _l.check(_l.load().pbe_geglu_f16(_p(h), _p(y), M, F, _stream()), "pbe_geglu_f16")
    return y


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    _req(t, torch.int64, "timestep_embedding t")
    t = t.contiguous()
    y = torch.empty((t.shape[0], dim), dtype=torch.float16, device=t.device)
    _l.check(_l.load().pbe_timestep_embedding_f16(_p(t), _p(y), t.shape[0], dim, float(max_period), _stream()), "pbe_timestep_embedding_f16")
    return y


def nchw_to_nhwc(x: torch.Tensor, cp: Optional[int] = None) -> torch.Tensor:
    """fp32 NCHW -> fp16 NHWC with the channel dim zero-padded to cp."""
    _f(x, "nchw_to_nhwc x")
    x = x.contiguous()
    B, Cc, H, W = x.shape
    cp = Cc if cp is None else cp
    y = torch.empty((B, H, W, cp), dtype=torch.float16, device=x.device)
    _l.check(_l.load().pbe_nchw_f32_to_nhwc_f16(_p(x), _p(y), B, Cc, H * W, cp, _stream()), "pbe_nchw_f32_to_nhwc_f16")
    return y


def nhwc_to_nchw(x: torch.Tensor, c: Optional[int] = None) -> torch.Tensor:
    """fp16 NHWC [B,H,W,ld] -> fp32 NCHW of the first c channels."""
    _h(x, "nhwc_to_nchw x")
    if not x.is_contiguous():
        raise _l.PbeError("nhwc_to_nchw: x must be contiguous")
    B, H, W, ld = x.shape
    c = ld if c is None else c
    y = torch.empty((B, c, H, W), dtype=torch.float32, device=x.device)
    _l.check(_l.load().pbe_nhwc_f16_to_nchw_f32(_p(x), _p(y), B, c, H * W, ld, _stream()), "pbe_nhwc_f16_to_nchw_f32")
    return y


def plms_pack_input(x: torch.Tensor, z_inpaint: torch.Tensor, mask: torch.Tensor, dup: int) -> torch.Tensor:
    _f(x, "plms x"); _f(z_inpaint, "plms z_inpaint"); _f(mask, "plms mask")
    B, _, H, W = x.shape
    if tuple(z_inpaint.shape) != (B, 4, H, W) or tuple(mask.shape) != (B, 1, H, W):
        raise _l.PbeError(f"plms_pack_input: shape mismatch x={tuple(x.shape)} z={tuple(z_inpaint.shape)} mask={tuple(mask.shape)}")
    x9 = torch.empty((dup * B, H, W, 16), dtype=torch.float16, device=x.device)
    _l.check(_l.load().pbe_plms_pack_input(_p(x.contiguous()), _p(z_inpaint.contiguous()), _p(mask.contiguous()), _p(x9), B, H * W, dup,
                                           _stream()), "pbe_plms_pack_input")
    return x9


def plms_update(eps_out: torch.Tensor, dup: int, cfg_scale: float, x: torch.Tensor, hist, coef8, want_e_t: bool = True,
                want_pred: bool = True):
    """Fused CFG combine + multistep weights + x_prev / pred_x0 (plms.py:188-189, 202-219, 230-244)."""
    _h(eps_out, "plms eps_out"); _f(x, "plms x")
    B, _, H, W = x.shape
    ld = eps_out.shape[-1]
    e_t = torch.empty_like(x) if want_e_t else None
    pred = torch.empty_like(x) if want_pred else None
    x_prev = torch.empty_like(x)
    h = [None, None, None]
    for i, t in enumerate(hist[:3]):
        h[i] = _f(t, "plms history")
    arr = (C.c_float * 8)(*[float(v) for v in coef8])
    _l.check(_l.load().pbe_plms_update(_p(eps_out), ld, dup, float(cfg_scale), _p(x), _p(h[0]), _p(h[1]), _p(h[2]), arr, _p(e_t),
                                       _p(x_prev), _p(pred), B, H * W, _stream()), "pbe_plms_update")
    return x_prev, pred, e_t


def axpy_(y: torch.Tensor, a: float, x: torch.Tensor) -> torch.Tensor:
    """y += a * x in place (fp32): the sigma_t * noise term of a stochastic DDIM step (ddim.py:236-238)."""
    _f(y, "axpy y"); _f(x, "axpy x")
    if not y.is_contiguous() or not x.is_contiguous() or x.numel() != y.numel():
        raise _l.PbeError("axpy_: contiguous tensors of equal size")
    _l.check(_l.load().pbe_axpy_f32(_p(y), float(a), _p(x), y.numel(), _stream()), "pbe_axpy_f32")
    return y


def qsample_blend(x0: torch.Tensor, noise: torch.Tensor, mask: torch.Tensor, img: torch.Tensor, sqrt_ac: float, sqrt_1m_ac: float) -> torch.Tensor:
    """img_orig = q_sample(x0, t); img_orig * mask + (1 - mask) * img (plms.py:150-153, ddim.py:178-181), fp32 NCHW."""
    for t, n in ((x0, "x0"), (noise, "noise"), (mask, "mask"), (img, "img")):
        _f(t, f"qsample_blend {n}")
    x0, noise, mask, img = x0.contiguous(), noise.contiguous(), mask.contiguous(), img.contiguous()
    B, Cc, H, W = img.shape
    if x0.shape != img.shape or noise.shape != img.shape or mask.shape[0] != B or mask.shape[1] not in (1, Cc) or tuple(mask.shape[2:]) != (H, W):
        raise _l.PbeError("qsample_blend: x0 / noise must match img, mask [B, 1 or C, H, W]")
    out = torch.empty_like(img)
    _l.check(_l.load().pbe_qsample_blend_f32(_p(x0), _p(noise), _p(mask), _p(img), float(sqrt_ac), float(sqrt_1m_ac), _p(out), B, Cc, H * W,
                                             mask.shape[1], _stream()), "pbe_qsample_blend_f32")
    return out


def posterior_sample(moments: torch.Tensor, eps: torch.Tensor, scale: float) -> torch.Tensor:
    _h(moments, "posterior moments"); _f(eps, "posterior eps")
    B, H, W, ld = moments.shape
    z = torch.empty((B, 4, H, W), dtype=torch.float32, device=moments.device)
    _l.check(_l.load().pbe_posterior_sample(_p(moments), ld, _p(eps.contiguous()), _p(z), B, H * W, float(scale), _stream()), "pbe_posterior_sample")
    return z


def scale_latent(z: torch.Tensor, inv_scale: float) -> torch.Tensor:
    _f(z, "scale_latent z")
    z = z.contiguous()
    B, Cc, H, W = z.shape
    y = torch.empty((B, H, W, 8), dtype=torch.float16, device=z.device)
    _l.check(_l.load().pbe_scale_latent_f16(_p(z), _p(y), B, Cc, H * W, float(inv_scale), _stream()), "pbe_scale_latent_f16")
    return y


def image_post(x: torch.Tensor) -> torch.Tensor:
    _h(x, "image_post x")
    B, H, W, ld = x.shape
    y = torch.empty((B, 3, H, W), dtype=torch.float32, device=x.device)
    _l.check(_l.load().pbe_image_post_f32(_p(x), _p(y), B, H * W, ld, _stream()), "pbe_image_post_f32")
    return y


def clip_patchify(pixels: torch.Tensor, patch: int, kp: int) -> torch.Tensor:
    _f(pixels, "clip_patchify pixels")
    pixels = pixels.contiguous()
    B, _, S, _ = pixels.shape
    g = S // patch
    y = torch.empty((B * g * g, kp), dtype=torch.float16, device=pixels.device)
    _l.check(_l.load().pbe_clip_patchify_f16(_p(pixels), _p(y), B, S, patch, kp, _stream()), "pbe_clip_patchify_f16")
    return y


def resize_bilinear(x: torch.Tensor, size, antialias: bool = True) -> torch.Tensor:
    """fp32 [B, C, H, W] -> [B, C, h, w], bilinear, align_corners=False (the mask resize of scripts/inference.py:332)."""
    _f(x, "resize_bilinear x")
    x = x.contiguous()
    B, Cc, H, W = x.shape
    h, w = int(size[0]), int(size[1])
    y = torch.empty((B, Cc, h, w), dtype=torch.float32, device=x.device)
    _l.check(_l.load().pbe_resize_bilinear_f32(_p(x), _p(y), B * Cc, H, W, h, w, 1 if antialias else 0, _stream()), "pbe_resize_bilinear_f32")
    return y


def u8_to_planes(src: torch.Tensor, mean=None, std=None, mask_mode: int = 0) -> torch.Tensor:
    """uint8 [B, H, W, C] (C = 3) or [B, H, W] (mask) on the GPU -> fp32 [B, C, H, W]: (v/255 - mean) / std, or the mask forms
    mask_mode 1: (1 - v/255) thresholded at 0.5 (scripts/inference.py:311-315), 2: 1 - v/255 (test_bench_dataset.py)."""
    _req(src, torch.uint8, "u8_to_planes src")
    src = src.contiguous()
    if src.dim() == 3:
        src = src.unsqueeze(-1)
    B, H, W, Cc = src.shape
    y = torch.empty((B, Cc, H, W), dtype=torch.float32, device=src.device)
    m = (C.c_float * 3)(*(list(mean) + [0.0] * 3)[:3]) if mean is not None else None
    sd = (C.c_float * 3)(*(list(std) + [1.0] * 3)[:3]) if std is not None else None
    _l.check(_l.load().pbe_u8_to_planes_f32(_p(src), _p(y), B, Cc, H * W, m, sd, mask_mode, _stream()), "pbe_u8_to_planes_f32")
    return y


def mul_planes(x: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    """x [B, C, H, W] * m [B, 1, H, W] (inpaint_image = image * mask)."""
    _f(x, "mul_planes x"); _f(m, "mul_planes m")
    x, m = x.contiguous(), m.contiguous()
    B, Cc, H, W = x.shape
    y = torch.empty_like(x)
    _l.check(_l.load().pbe_mul_planes_f32(_p(x), _p(m), _p(y), B, Cc, H * W, _stream()), "pbe_mul_planes_f32")
    return y


def planes_to_canvas(src: torch.Tensor, canvas: torch.Tensor, y0: int, x0: int, a=(1.0, 1.0, 1.0), b=(0.0, 0.0, 0.0)) -> None:
    """One fp32 [3, H, W] (or [1, H, W]: broadcast to 3 channels) image -> canvas[y0:y0+H, x0:x0+W, :] (uint8 [Hc, Wc, 3]) as
    trunc(255 * clamp(x * a[c] + b[c], 0, 1))."""
    _f(src, "planes_to_canvas src"); _req(canvas, torch.uint8, "planes_to_canvas canvas")
    src = src.contiguous()
    Cc, H, W = src.shape
    if not canvas.is_contiguous() or canvas.dim() != 3 or canvas.shape[2] != 3 or Cc not in (1, 3):
        raise _l.PbeError("planes_to_canvas: canvas must be a contiguous [Hc, Wc, 3] uint8 tensor, src [3 or 1, H, W]")
    aa, bb = (C.c_float * 3)(*[float(v) for v in a]), (C.c_float * 3)(*[float(v) for v in b])
    _l.check(_l.load().pbe_planes_to_u8_canvas(_p(src), _p(canvas), H, W, canvas.shape[0], canvas.shape[1], int(y0), int(x0), aa, bb, 1 if Cc == 1 else 0,
                                               _stream()), "pbe_planes_to_u8_canvas")


def bcast_row(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, B: int, y_bs: int) -> None:
    """out[bi * y_bs + c] = a[c] + b[c] for bi < B."""
    _h(a, "bcast_row a"); _h(b, "bcast_row b"); _h(out, "bcast_row out")
    _l.check(_l.load().pbe_bcast_row_f16(_p(a), _p(b), _p(out), B, a.numel(), y_bs, _stream()), "pbe_bcast_row_f16")


# ---- weight packing (one-off, at load time) ---------------------------------------------------
def pack_conv3x3(w: torch.Tensor, cin_pad: Optional[int] = None, split: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """OIHW fp32 -> packed fp16 [Cout, 9*Cin] in the K order the kernels gather in.
    * cin_pad is None (Cin % 64 == 0, implicit-GEMM conv): k = ((ci // cb) * 9 + ky*3+kx) * cb + ci % cb with
      cb = conv_kblock(Cin) — channel-block major, tap minor, so a pixel's 9 shifted reads stay within 9*cb/64
      k-tiles of each other (L2 reuse) while cb/64 consecutive k-tiles walk the same pixel (+64 channels).
    * cin_pad given (tiny Cin, im2col + GEMM path): k = (ky*3+kx) * cin_pad + ci, zero-padded channels."""
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    wp = w.permute(0, 2, 3, 1)                                   # [Co, 3, 3, Ci]
    if cin_pad is not None:
        if cin_pad != ci:
            wp = torch.nn.functional.pad(wp, (0, cin_pad - ci))
        return wp.reshape(co, -1).to(torch.float16).contiguous()
    assert ci % 64 == 0, "implicit-GEMM conv needs Cin % 64 == 0 (use cin_pad for the im2col path)"
    cb = conv_kblock(*(split if split else (ci, 0)))       # split = (C1, C2) when the conv reads a channel concat
    assert not split or sum(split) == ci
    wp = wp.reshape(co, 9, ci // cb, cb).permute(0, 2, 1, 3)      # [Co, Ci/cb, 9, cb]
    return wp.reshape(co, 9 * ci).to(torch.float16).contiguous()


def pack_conv3x3_up_phases(w: torch.Tensor) -> torch.Tensor:
    """OIHW fp32 weight of a 3x3 conv that follows a nearest-2x upsample (openaimodel.py:109-119, model.py:44-53) -> fp16 [4, Cout, 4*Cin]:
    output pixel (2y + py, 2x + px) reads only the 2x2 source block rows {y - 1 + py, y + py} x columns {x - 1 + px, x + px}, so per phase
    (py, px) the 9 taps collapse to 4 with summed weights (py = 0: ky {0} | {1, 2}; py = 1: ky {0, 1} | {2}; same in x), summed in fp32
    before the single fp16 rounding.  K order per phase: k = ((ci // 64) * 4 + ty * 2 + tx) * 64 + ci % 64.  4 / 9 of the MACs of the fused
    upsample gather."""
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3 and ci % 64 == 0
    w32 = w.detach().float()
    sets = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
    out = []
    for py in (0, 1):
        for px in (0, 1):
            taps = []
            for ty in (0, 1):
                for tx in (0, 1):
                    acc = 0
                    for ky in sets[py][ty]:
                        for kx in sets[px][tx]:
                            acc = acc + w32[:, :, ky, kx]
                    taps.append(acc)                          # [Co, Ci]
            wp = torch.stack(taps, 1)                         # [Co, 4, Ci]
            wp = wp.reshape(co, 4, ci // 64, 64).permute(0, 2, 1, 3).reshape(co, 4 * ci)
            out.append(wp)
    return torch.stack(out, 0).to(torch.float16).contiguous()


def conv_kblock(c1: int, c2: int = 0) -> int:
    """Channel block of the conv K order (deterministic, shared by pack_conv3x3 and conv3x3): 64, the kernels' k-tile.  k = ((ci // 64)
    * 9 + tap) * 64 + ci % 64: the 9 taps of a 64-channel block are consecutive k-tiles, which is what the halo-resident tiles need
    (one halo image per block serves its 9 taps) and keeps the gather kernel's shifted re-reads within 9 k-tiles (L2 hits)."""
    if c1 % 64 or c2 % 64:
        raise _l.PbeError(f"conv: C1={c1} / C2={c2} are not multiples of 64")
    return 64


def pack_linear(w: torch.Tensor) -> torch.Tensor:
    return w.reshape(w.shape[0], -1).to(torch.float16).contiguous()


def pack_linear_ln(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """Linear(LayerNorm(x)) with the LayerNorm FOLDED into the GEMM (pbe_gemm_desc.ln_stats): returns (W * gamma as fp16 [N, K], the
    fp32 bias W beta + b, colsum[n] = sum_k of the fp16 values of row n) - LN(x) W^T + b = rstd (x (W gamma)^T - mean colsum) + W beta + b."""
    w32 = w.detach().reshape(w.shape[0], -1).float()
    wg = (w32 * gamma.detach().float()[None, :]).to(torch.float16).contiguous()
    c2 = w32.double() @ beta.detach().double()
    if bias is not None:
        c2 = c2 + bias.detach().double()
    return wg, c2.float().contiguous(), wg.double().sum(1).float().contiguous()


def pack_geglu(w: torch.Tensor, b: torch.Tensor):
    """GEGLU projection [2F, K] (rows 0..F-1 = value, F..2F-1 = gate, attention.py:41-45) -> rows interleaved
    (x_0, g_0, x_1, g_1, ...) so the GEMM epilogue sees each (value, gate) pair in one accumulator quad."""
    F = w.shape[0] // 2
    wi = torch.stack([w[:F], w[F:]], 1).reshape(2 * F, -1)
    bi = torch.stack([b[:F], b[F:]], 1).reshape(2 * F)
    return wi.to(torch.float16).contiguous(), bi.detach().float().contiguous()


def tune(key: int, value: int) -> None:
    _l.check(_l.load().pbe_tune(key, value), "pbe_tune")


# ---- profiling --------------------------------------------------------------------------------
def prof_enable(on: bool) -> None:
    _l.load().pbe_prof_enable(1 if on else 0)


def prof_reset() -> None:
    _l.load().pbe_prof_reset()


def prof_collect():
    lib = _l.load()
    buf = (C.c_double * (5 * 16))()
    n = lib.pbe_prof_collect(buf, 16)
    out = {}
    for k in range(n):
        out[lib.pbe_prof_class_name(k).decode()] = {"launches": int(buf[5 * k]), "ms": buf[5 * k + 1], "work": buf[5 * k + 2],
                                                    "bytes": buf[5 * k + 3], "roofline_ms": buf[5 * k + 4]}
    return out
