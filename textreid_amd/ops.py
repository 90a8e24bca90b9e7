"""Tensor-level wrappers over the C ABI (no autograd here).

Every function takes fp32 CUDA tensors, allocates outputs through PyTorch's
caching allocator (plumbing) and enqueues the HIP kernels of
libtextreid_hip.so on the current stream.  Nothing in this module computes
with torch ops, and nothing falls back to the CPU: a CPU tensor raises.
"""

import ctypes
import os
import math

import torch

from . import lib as L
from .lib import GemmDesc, call

A_KC, A_MC, A_CONV = L.TRID_A_KC, L.TRID_A_MC, L.TRID_A_CONV
B_KC, B_NC, B_CONV = L.TRID_B_KC, L.TRID_B_NC, L.TRID_B_CONV

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
STATS_ROWS = 128  # rows per BatchNorm-statistics partial (GEMM tile height)
# GEMM arithmetic (trid_gemm_desc.precision):
#   6 (default): fp32 operands split on the fly into 3 bf16 planes, 6 bf16 MFMA products per
#      multiply-add, fp32 accumulate -- dropped terms <= 2^-26, i.e. fp32-class results (parity
#      tests pass at the same error level as the exact path) at ~1.5x the fp32-MFMA rate;
#   0: exact fp32-input MFMA (v_mfma_f32_32x32x2_f32);
#   3: 2 planes / 3 products (~2^-17 per product): faster, NOT parity-safe, never the default.
#  16: fp32 operands scaled per tensor by a power of two and split into TWO fp16 planes (11 + 11 significand bits),
#      3 fp16 MFMA products per multiply-add in two fp32 accumulators (error <= 3 * 2^-22 per product: fp32-class, half
#      the MFMA work of 6); needs the operands' largest magnitudes as device scalars (`amax`).
GEMM_PRECISION = int(__import__("os").environ.get("TRID_GEMM_PRECISION", "6"))
# arithmetic of the image encoder's convolutions (95 % of the step's FLOPs); 6 = same as everything else
CONV_PRECISION = int(__import__("os").environ.get("TRID_CONV_PRECISION", "16"))


def conv_precision():
    """16 only while the global mode is the fp32-class default; TRID_GEMM_PRECISION=0/1/3 overrides everything."""
    return CONV_PRECISION if GEMM_PRECISION == 6 else GEMM_PRECISION


_raw_stream = torch._C._cuda_getCurrentRawStream  # (the raw handle without building a torch.cuda.Stream object: ~0.3 us instead of ~7)
_cur_device = torch._C._cuda_getDevice


_SLOW_STREAM = os.environ.get("TRID_SLOW_STREAM", "0") == "1"  # (A/B runs: the torch.cuda.Stream-object path)


def stream(device=None):
    """hipStream_t of torch's current stream on `device` (default: the current device) as an integer.  Called ~900 times per
    train step: torch.cuda.current_stream() cost 4.4 ms of host time per step (tools/exp/host_profile.py)."""
    if _SLOW_STREAM:
        return torch.cuda.current_stream(device).cuda_stream
    return _raw_stream(_cur_device() if device is None or device.index is None else device.index)


# Generation counter of raw-pointer parameter / buffer writes.  The library's own writers (FusedAdam.step, the EMA
# kernel, the running-statistics update of bn_finalize) go through device pointers, so torch's `tensor._version`
# does not move; anything cached as a function of parameter VALUES (the eval path's BatchNorm-folded filters) keys
# on this counter as well.
_param_generation = [0]


def note_parameter_write():
    _param_generation[0] += 1


def parameter_generation():
    return _param_generation[0]


# Optional live profiling hook (bench.py): when PROFILE is a dict with a "match" function
# (kernel key -> label or None), every labelled GEMM launch is bracketed by events on the launch stream.
PROFILE = None


def _uses_split(M, N, K, a_mode, conv, prec=None):
    """Mirror of the split-kernel eligibility test in trid_gemm_f32()."""
    pr = GEMM_PRECISION if prec is None else prec
    return (pr in (1, 3, 6, 16) and K % 8 == 0 and K >= 32 and M >= 64 and N >= (32 if pr == 16 else 64) and (M >= 96 or N >= 96)
            and (a_mode != A_CONV or conv[2] % 8 == 0))


def _gemm_tile(M, N, stats):
    """Mirror of dispatch_tile() in csrc/gemm.hip."""
    bm = 128 if M >= 96 else (64 if M > 32 else 32)
    bn = 128 if N >= 96 else (64 if N > 32 else 32)
    if stats:
        bm = 128
    if bm == 128:
        return (128, bn)
    if bn == 128:
        return (bm, 128)
    return (64, 64)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("textreid_amd kernels need CUDA (HIP) tensors; got a CPU tensor -- there is no CPU fallback")
    return t.data_ptr()


def empty(shape, like=None, dtype=torch.float32, device=None):
    return torch.empty(shape, dtype=dtype, device=device if device is not None else like.device)


# --------------------------------------------------------------------------- GEMM
_amax_pool = {}
_amax_pool_capture = {}


def begin_capture():
    """engine.graph.CapturedTrainStep, right before a stream capture: slots handed out INSIDE the capture come from
    pools that are allocated (and zero-filled: a recorded fill, re-executed by every replay) inside that capture -
    a pool filled once in eager mode would hand the replays slots that still hold the previous replay's maxima."""
    _amax_pool_capture.clear()
    _finalize_ws_capture.clear()


_finalize_ws = {}
_finalize_ws_capture = {}
# bf16 mode (conv precision 1): the residual blocks' DATA gradients (dL/d block output, dL/d conv input) are bf16 tensors,
# as the gradients of bf16 tensors are under autocast; 0: fp32 gradient tensors, only the GEMM operands rounded (A/B runs)
BF16_GRADS = os.environ.get("TRID_BF16_GRADS", "1") != "0"
# ... and the raw conv outputs y of the residual blocks are bf16 tensors (BatchNorm statistics are those of the rounded
# tensor, as under autocast); 0: y stays fp32 (A/B runs)
BF16_Y = os.environ.get("TRID_BF16_Y", "1") != "0"
_FINALIZE_SPLIT = os.environ.get("TRID_BN_FINALIZE_SPLIT", "1") != "0"  # (0: always one workgroup per channel, for A/B runs)


def bn_finalize_ws(device):
    """Scratch of trid_bn_finalize_* (range sums of its two-launch form), private to a (device, stream): kernels on one
    stream are serial.  (Inside a capture it comes from the capture's pool, see begin_capture.)"""
    if not _FINALIZE_SPLIT:
        return None
    key = (device, stream(device))
    cache = _finalize_ws_capture if torch.cuda.is_current_stream_capturing() else _finalize_ws
    ws = cache.get(key)
    if ws is None:
        ws = torch.empty(int(L.load().trid_bn_finalize_ws_bytes()), dtype=torch.uint8, device=device)
        cache[key] = ws
    return ws


def amax_slot(device):
    """A zero-initialised device scalar (1-element view) for a kernel's `amax` side output.  Slots come from a
    zero-filled pool per (device, stream) and are written once, so there is no per-call memset; the view keeps its
    pool buffer alive."""
    key = (device, stream(device))
    pools = _amax_pool_capture if torch.cuda.is_current_stream_capturing() else _amax_pool
    ent = pools.get(key)
    if ent is None or ent[1] >= ent[0].numel():
        ent = [torch.zeros(4096, dtype=torch.float32, device=device), 0]
        pools[key] = ent
    slot = ent[0][ent[1] : ent[1] + 1]
    ent[1] += 1
    return slot


def amax(t):
    """Device scalar holding max|t|: one streaming pass (operands whose producer has no amax side output)."""
    slot = amax_slot(t.device)
    call("trid_amax_f32", _p(t), t.numel(), _p(slot), stream())
    return slot


def gemm(A, B, C, M, N, K, lda, ldb, ldc, a_mode=A_KC, b_mode=B_KC, alpha=1.0, accumulate=False, bias=None,
         stats=None, batch=1, strideA=0, strideB=0, strideC=0, splits=1, strideSplit=0, conv=None, a_off=0, b_off=0,
         c_off=0, strideBias=0, bias_off=0, residual=None, ldres=0, relu=False, precision=None, a_amax=None, b_amax=None):
    """Raw descriptor call.  a_off/b_off/c_off are element offsets into A/B/C.  precision=16 (fp16-split
    arithmetic) takes the operands' largest magnitudes as device scalars; a missing one is computed here over the
    WHOLE tensor object passed (a superset of the operand is a valid, merely looser, scale)."""
    if (a_mode == A_KC and b_mode in (B_KC, B_NC) and stats is None and splits == 1 and residual is None and not relu and conv is None
            and a_off == 0 and b_off == 0 and c_off == 0 and bias_off == 0 and batch <= 65535
            and _skinny_ok(M, N, K, lda, ldb if b_mode == B_KC else None, precision, SKINNY_MIN_K_BATCHED) and A.data_ptr() % 16 == 0
            and (b_mode == B_NC or B.data_ptr() % 16 == 0) and strideA % 4 == 0 and (b_mode == B_NC or strideB % 4 == 0)
            and (batch == 1 or M >= SKINNY_MIN_M_BATCHED)):
        return skinny_gemm(A, lda, B, ldb, b_mode, C, ldc, M, N, K, bias=bias, alpha=alpha, accumulate=accumulate, batch=batch,
                           strideA=strideA, strideB=strideB, strideC=strideC, strideBias=strideBias)
    prec = GEMM_PRECISION if precision is None else precision
    if prec == 16 and _uses_split(M, N, K, a_mode, conv, 16):
        if a_amax is None:
            a_amax = amax(A)
        if b_amax is None:
            b_amax = amax(B)
    d = GemmDesc()
    d.A = _p(A) + 4 * a_off
    d.B = _p(B) + 4 * b_off
    d.C = _p(C) + 4 * c_off
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.strideA, d.strideB, d.strideC = strideA, strideB, strideC
    d.batch, d.splits, d.strideSplit = batch, splits, strideSplit
    d.a_mode, d.b_mode = a_mode, b_mode
    d.alpha = alpha
    d.accumulate = 1 if accumulate else 0
    d.bias = (_p(bias) + 4 * bias_off) if bias is not None else None
    d.strideBias = strideBias
    d.stats = _p(stats)
    if conv is not None:
        d.H, d.W, d.Cin = conv
    d.precision = prec
    d.a_amax = _p(a_amax)
    d.b_amax = _p(b_amax)
    d.residual = _p(residual)
    d.ldres = ldres
    d.relu = 1 if relu else 0
    prof = PROFILE
    if prof is not None:
        split = _uses_split(M, N, K, a_mode, conv, prec)
        key = (a_mode, b_mode) + ((128, 64 if N <= 64 else 128) if split else _gemm_tile(M, N, stats is not None)) + (split,)
        label = prof["match"](key)
        if label:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call("trid_gemm_f32", ctypes.addressof(d), stream())
            e1.record()
            prof["events"].append((label, 2.0 * M * N * K * batch, e0, e1))
            return
    call("trid_gemm_f32", ctypes.addressof(d), stream())


# --------------------------------------------------------------------------- P16 (pre-split) operands
class P16:
    """A GEMM operand stored pre-split (csrc/gemm_p16.hip): `data` holds, per row and 32-wide K group, 32 fp16 high parts
    then 32 fp16 low parts of x * 2^s (4 bytes per element, same shape / dtype as the fp32 tensor it replaces so it
    travels through the allocator unchanged); `amax` is the device scalar the scale s was derived from."""

    __slots__ = ("data", "amax", "fmt", "tmax")

    def __init__(self, data, amax, fmt=1, tmax=None):
        """fmt 1: P16 proper (fp32-class).  fmt 2: a plain bf16 tensor (configs[3]'s arithmetic: the residual blocks'
        convolutions read bf16 operands) - same producers / consumers, no scale, half the bytes.
        tmax (eval-mode flow): device scalar holding the TRUE max|x| where `amax` is only a bound of it."""
        self.data, self.amax, self.fmt = data, amax, fmt
        self.tmax = tmax if tmax is not None else amax

    @property
    def shape(self):
        return self.data.shape

    def unpack(self):
        if self.fmt == 2:
            return self.data.float()  # (taps / masks of the parity tests)
        K = self.data.shape[-1]
        out = torch.empty_like(self.data)
        call("trid_p16_unpack_f32", _p(self.data), self.data.numel() // K, K, _p(self.amax), _p(out), stream())
        return out


P16_VARIANT = int(__import__("os").environ.get("TRID_P16_VARIANT", "-1"))  # tile shape of trid_gemm_p16; < 0: the library's choice per shape


class Partials:
    """BatchNorm partials of a convolution epilogue with the number of rows each of them covers (the tile height of the
    kernel that wrote them: 128, 96 or 64 rows, or whole image-row bands) - bn_finalize_minmax needs both."""

    __slots__ = ("data", "rows")

    def __init__(self, data, rows):
        self.data, self.rows = data, int(rows)

    @property
    def shape(self):
        return self.data.shape

    @property
    def rows_per_part(self):
        return self.rows


def p16_empty(shape, like, fmt):
    return torch.empty(shape, dtype=torch.float32 if fmt == 1 else torch.bfloat16, device=like.device)


def p16_pack(x, amax_=None, fmt=1):
    """fp32 [..., K] (K % 32 == 0, rows contiguous) -> P16 (fmt 1) / bf16 (fmt 2) with the same shape."""
    K = x.shape[-1]
    if amax_ is None and fmt == 1:
        amax_ = amax(x)
    out = p16_empty(x.shape, x, fmt)
    call("trid_p16_pack_f32", _p(x), x.numel() // K, K, K, _p(amax_) if fmt == 1 else None, _p(out), fmt, stream())
    return P16(out, amax_ if fmt == 1 else None, fmt)


def p16_pack_wt(w, N, T, C, flip, amax_):
    """w [N][T][C] fp32 -> P16 [C][T*N] (taps reversed when flip): the data-gradient operand of a conv."""
    out = empty((C, T * N), w)
    call("trid_p16_pack_wt_f32", _p(w), N, T, C, 1 if flip else 0, _p(amax_), _p(out), stream())
    return P16(out, amax_)


class BnBwdSums:
    """The BatchNorm-backward sums of a gradient tensor, written by the data-gradient GEMM that produces it (gemm_p16(bn_bwd=...),
    csrc/gemm_p16.hip BnBwdFuse) and consumed by bn_bwd_p16(presummed=...) instead of its reduce pass over g and y."""

    __slots__ = ("y", "st", "relu", "ws", "ws2", "M", "mask")

    def __init__(self, y, st, relu=True, mask=None):
        """mask: the ReLU bit mask of the BLOCK OUTPUT the layer feeds (bn_apply_p16(want_mask=True)) - a residual block's bn3,
        whose own sign says nothing about the ReLU that follows the identity add (bn_bwd_p16's mask mode 3)."""
        C = y.shape[-1]
        self.y, self.st, self.relu, self.mask = y, st, relu, mask
        self.M = y.numel() // C
        n = _query("trid_bn_bwd_fused_ws_floats", int(self.M), int(C))
        both = empty((2 * n,), y)
        self.ws, self.ws2 = both[:n], both[n:]


USE_BNB_FUSE = os.environ.get("TRID_BNB_FUSE", "1") != "0"  # BatchNorm-backward sums from the producing GEMM's epilogue (0: A/B runs)


def bn_bwd_fusable(y, M, N, fmt=1):
    """Can the data-gradient GEMM that writes g [M, N] also form the BatchNorm-backward sums against the saved conv output y?
    (the tile kernel's staged store path: P16 operands, fp32 g and y, whole 64- / 128-column tiles; the streaming short-K kernel
    and the ring-of-rows kernel do not carry the epilogue)"""
    return (USE_BNB_FUSE and fmt == 1 and y.dtype == torch.float32 and y.numel() == M * N
            and _query("trid_gemm_p16_bnb_ok", int(M), int(N)) == 1)  # (the library's own precondition, not a mirror of it)


def gemm_p16(A, B, C, M, N, K, ldc, conv=None, alpha=1.0, accumulate=False, bias=None, stats=None, residual=None, ldres=0,
             relu=False, splits=1, strideSplit=0, variant=None, minmax=False, cmask=None, bn_bwd=None):
    """C[M,N] = alpha * A . B^T with both operands P16: A [M][K] (or an NHWC image [B,H,W,Cin] with conv=(H,W,Cin),
    K = 9*Cin), B [N][K].  minmax: the BatchNorm partials `stats` are [tiles][N][4] = (mean, M2, min, max).
    cmask (with accumulate, ldc == N): a relu_mask of bn_apply_p16 over C; C = A . B^T + (bit ? C : 0).
    bn_bwd: a BnBwdSums - C is a gradient that feeds that BatchNorm layer's backward; its sums come out of the epilogue."""
    if (USE_STREAM and conv is None and A.fmt == 1 and B.fmt == 1 and C.dtype == torch.float32 and stats is None and alpha == 1.0
            and bias is None and residual is None and not relu and splits == 1 and bn_bwd is None and gemm_p16_stream_rows(M, N, K, accumulate)
            and (cmask is None or N % 256 == 0)):
        return gemm_p16_stream(A, B, C, M, N, K, ldc, accumulate=accumulate, cmask=cmask)  # short-K data gradients (conv1 of a block)
    d = GemmDesc()
    d.A, d.B, d.C = _p(A.data), _p(B.data), _p(C)
    d.M, d.N, d.K = M, N, K
    d.lda = conv[2] if conv is not None else K
    d.ldb, d.ldc = K, ldc
    d.batch, d.splits, d.strideSplit = 1, splits, strideSplit
    d.a_mode, d.b_mode = (A_CONV if conv is not None else A_KC), B_KC
    d.alpha = alpha
    d.accumulate = 1 if accumulate else 0
    d.bias = _p(bias)
    d.stats = _p(stats)
    if conv is not None:
        d.H, d.W, d.Cin = conv
    assert A.fmt == B.fmt
    d.precision = 16 if A.fmt == 1 else 1
    d.a_amax, d.b_amax = _p(A.amax), _p(B.amax)
    d.residual = _p(residual)
    d.ldres = ldres
    d.relu = 1 if relu else 0
    d.c_mask = _p(cmask)
    d.stats_minmax = 1 if minmax else 0
    d.c_format = 2 if C.dtype == torch.bfloat16 else 0  # (data gradients of the bf16 mode are bf16 tensors)
    if bn_bwd is not None:
        st = bn_bwd.st
        d.bnb_y, d.bnb_mean, d.bnb_invstd, d.bnb_scale, d.bnb_shift = _p(bn_bwd.y), _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift)
        d.bnb_ws, d.bnb_ws2 = _p(bn_bwd.ws), _p(bn_bwd.ws2)
        d.bnb_relu, d.bnb_mask = (2, _p(bn_bwd.mask)) if bn_bwd.mask is not None else ((1 if bn_bwd.relu else 0), None)
    v = P16_VARIANT if variant is None else variant
    if stats is not None:
        rows = gemm_p16_rows(M, N, A.fmt, v)
        if stats.shape[0] != (M + rows - 1) // rows:
            raise RuntimeError("gemm_p16: the partials buffer must hold one entry per %d rows (gemm_p16_rows): %d parts for M=%d" % (rows, stats.shape[0], M))
    prof = PROFILE
    if prof is not None and N > 64:
        label = prof["match"]((A_CONV if conv is not None else A_KC, B_KC, 128, 128, True))
        if label:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call("trid_gemm_p16", ctypes.addressof(d), v, stream())
            e1.record()
            prof["events"].append((label, 2.0 * M * N * K, e0, e1))
            return
    call("trid_gemm_p16", ctypes.addressof(d), v, stream())


USE_STREAM = __import__("os").environ.get("TRID_STREAM_1X1", "1") != "0"  # short-K 1x1 convolutions on csrc/gemm_stream.hip (0: A/B runs)


@__import__("functools").lru_cache(maxsize=None)
def _query(name, *ints):
    """A pure shape query of the library (cached: the same few shapes come back every step, and a ctypes call costs ~2 us of the
    host time an eager step has to stay under)."""
    return int(getattr(L.load(), name)(*ints))


def gemm_p16_rows(M, N, fmt=1, variant=None):
    """Rows per BatchNorm partial (= tile height) of trid_gemm_p16 for this shape: 128, or 96 where the library's dispatch picks
    its 96-row tiles (an M x N grid that leaves the last round of resident workgroups partly empty)."""
    return _query("trid_gemm_p16_rows", int(M), int(N), 16 if fmt == 1 else 1, P16_VARIANT if variant is None else int(variant))


def gemm_p16_stream_rows(M, N, K, accumulate=False):
    """Rows per step (= per BatchNorm partial) of the streaming short-K kernel for this shape; 0: not covered."""
    return _query("trid_gemm_p16_stream_rows", int(M), int(N), int(K), 1 if accumulate else 0)


def gemm_p16_stream(A, B, C, M, N, K, ldc, accumulate=False, stats=None, cmask=None):
    """C[M,N] (+)= A . B^T on the streaming kernel (A P16 [M][K], B P16 [N][K], K in {64, 128, 256}); cmask: see gemm_p16."""
    prof = PROFILE
    if prof is not None and prof["match"]("stream1x1"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("trid_gemm_p16_stream", _p(A.data), _p(A.amax), _p(B.data), _p(B.amax), _p(C), ldc, _p(stats), M, N, K, 1 if accumulate else 0, _p(cmask), stream())
        e1.record()
        # (events carry ALGORITHMIC BYTES for this HBM-bound kernel: A once, C once - twice when accumulating -, the filter)
        prof["events"].append(("stream1x1", 4.0 * (M * K + M * N * (2 if accumulate else 1) + N * K), e0, e1))
        return
    call("trid_gemm_p16_stream", _p(A.data), _p(A.amax), _p(B.data), _p(B.amax), _p(C), ldc, _p(stats), M, N, K, 1 if accumulate else 0, _p(cmask), stream())


USE_P16 = __import__("os").environ.get("TRID_P16", "1") != "0"  # residual blocks on pre-split operands (csrc/gemm_p16.hip)
USE_P16_STEM = __import__("os").environ.get("TRID_P16_STEM", "1") != "0"  # ... and the stem on csrc/stem_conv.hip (0: A/B runs)


USE_HALO_BLOCKS = __import__("os").environ.get("TRID_HALO_BLOCKS", "1") != "0"  # layer1's 3x3 convolutions on csrc/stem_conv.hip (0: A/B runs)


def conv_p16(x, w, conv3=False, stats=True):
    """x: P16 [B,H,W,C] (or [M,C]); w: P16 [N, K] -> raw conv output y fp32 [.., N] (+ Partials: (mean, M2, min, max) per
    column and row group, (mean, M2) for bf16 operands)."""
    C = x.shape[-1]
    M = x.data.numel() // C
    N = w.shape[0]
    y = empty(tuple(x.shape[:-1]) + (N,), x.data, dtype=torch.bfloat16 if (x.fmt == 2 and BF16_Y) else torch.float32)
    mm = stats and x.fmt == 1  # bf16 operands need no scale, hence no extremes
    if not conv3 and x.fmt == 1 and USE_STREAM:
        rows = gemm_p16_stream_rows(M, N, C)  # short reductions (K = 64 / 128 / 256): the streaming kernel
        if rows:
            st = empty(((M + rows - 1) // rows, N, 4), x.data) if stats else None
            gemm_p16_stream(x, w, y, M, N, C, N, stats=st)
            return (y, Partials(st, rows)) if stats else y  # (64-row partials for K = 256)
    if conv3 and x.fmt == 1 and USE_HALO_BLOCKS and y.dtype == torch.float32 and conv3x3_halo_rows(x.shape[1], x.shape[2], C, N):
        if not stats:  # 64-channel 3x3 convolutions at large maps (layer1's conv2): the stem's ring-of-rows kernel
            return conv3x3_halo_p16(x, w, stats=False)
        y, st, rows = conv3x3_halo_p16(x, w)
        return y, Partials(st, rows)
    rows = gemm_p16_rows(M, N, x.fmt)
    st = empty(((M + rows - 1) // rows, N, 4 if mm else 2), x.data) if stats else None
    if conv3:
        gemm_p16(x, w, y, M, N, 9 * C, N, conv=(x.shape[1], x.shape[2], C), stats=st, minmax=mm)
    else:
        gemm_p16(x, w, y, M, N, C, N, stats=st, minmax=mm)
    return (y, Partials(st, rows)) if stats else y


def conv3x3_halo_rows(H, W, Cin, Cout):
    """Image rows per step of the stem's ring-of-rows convolution kernel for this geometry (csrc/stem_conv.hip), 0 when
    the kernel does not cover it (channel counts other than 32 / 64, a width that does not tile 128 / 256 pixels)."""
    return _query("trid_conv3x3_halo_rows", int(H), int(W), int(Cin), int(Cout))


def conv3x3_halo_p16(x, w, stats=True, chunks_per_image=0):
    """3x3 / stride 1 / pad 1 convolution of a P16 NHWC image x [B,H,W,Cin] with P16 filters w [Cout, 9*Cin] (Cin, Cout in
    {32, 64}: the stem) -> raw output y fp32 [B,H,W,Cout]; with stats also the (mean, M2, min, max) BatchNorm partials
    [steps][Cout][4] and the rows each of them covers (the `rows_per_part` of bn_finalize_minmax)."""
    Bi, H, W, C = x.shape
    N = w.shape[0]
    th = conv3x3_halo_rows(H, W, C, N)
    if th == 0 or x.fmt != 1 or w.fmt != 1:
        raise RuntimeError("conv3x3_halo_p16: geometry H=%d W=%d Cin=%d Cout=%d (or a non-P16 operand) is not covered" % (H, W, C, N))
    y = empty((Bi, H, W, N), x.data)
    st = empty((Bi * H // th, N, 4), x.data) if stats else None
    call("trid_conv3x3_halo_p16", _p(x.data), _p(x.amax), _p(w.data), _p(w.amax), _p(y), _p(st), Bi, H, W, C, N, int(chunks_per_image), stream())
    return (y, st, th * W) if stats else y


def stem_conv1(images, w, stats=True):
    """The stem's first convolution (3 -> 32 channels, 3x3, stride 2, pad 1) straight from the NCHW image batch: y fp32
    [B,Ho,Wo,32] (+ per-128-row (mean, M2, min, max) partials)."""
    Bi, Cin, Hi, Wi = images.shape
    if Cin != 3 or w.shape[0] != 32 or not w.is_contiguous() or not images.is_contiguous():
        raise RuntimeError("stem_conv1: a contiguous [B,3,H,W] image batch and contiguous [32,3,3,3] filters are needed")
    Ho, Wo = (Hi + 1) // 2, (Wi + 1) // 2
    y = empty((Bi, Ho, Wo, 32), images)
    M = Bi * Ho * Wo
    st = empty(((M + STATS_ROWS - 1) // STATS_ROWS, 32, 4), images) if stats else None
    call("trid_stem_conv1_f32", _p(images), _p(w), _p(y), _p(st), Bi, Hi, Wi, stream())
    return (y, st) if stats else y


USE_HALO_WGRAD = __import__("os").environ.get("TRID_HALO_WGRAD", "1") != "0"  # 3x3 weight gradients with 32 / 64 channels on the ring-of-rows kernel (0: A/B runs)


def conv3x3_wgrad_halo_rows(H, W, Cin, Cout):
    return _query("trid_conv3x3_wgrad_halo_rows", int(H), int(W), int(Cin), int(Cout)) if USE_HALO_WGRAD else 0


def conv3x3_wgrad_halo_p16(dy, x):
    """Weight gradient of a 3x3 / stride 1 / pad 1 convolution with 32 / 64 channels: dy P16 [B,H,W,N], x P16 [B,H,W,C] -> dw fp32
    [N, 9*C] (column = tap * C + c, as wgrad_p16(conv=...)); every pixel staged once (csrc/stem_conv.hip)."""
    Bi, H, W, C = x.shape
    N = dy.shape[-1]
    if dy.fmt != 1 or x.fmt != 1 or not conv3x3_wgrad_halo_rows(H, W, C, N):
        raise RuntimeError("conv3x3_wgrad_halo_p16: geometry H=%d W=%d Cin=%d Cout=%d (or a non-P16 operand) is not covered" % (H, W, C, N))
    dw = empty((N, 9 * C), x.data)
    slabs = empty((_query("trid_conv3x3_wgrad_halo_slabs"), N * 9 * C), x.data)
    call("trid_conv3x3_wgrad_halo_p16", _p(dy.data), _p(dy.amax), _p(x.data), _p(x.amax), _p(dw), _p(slabs), Bi, H, W, C, N, stream())
    return dw


def stem_conv1_wgrad(images, dy):
    """Weight gradient of the stem's first convolution: dw [32, 3, 3, 3] from the NCHW image batch and dy fp32 [B,Ho,Wo,32]
    (exact fp32 MFMA, no im2col tensor)."""
    Bi, Cin, Hi, Wi = images.shape
    if Cin != 3 or dy.shape[-1] != 32 or dy.dtype != torch.float32 or not dy.is_contiguous() or not images.is_contiguous():
        raise RuntimeError("stem_conv1_wgrad: a contiguous [B,3,H,W] image batch and a contiguous fp32 [B,Ho,Wo,32] gradient are needed")
    dw = empty((32, 3, 3, 3), images)
    slabs = empty((_query("trid_stem_conv1_wgrad_slabs"), 32 * 27), images)
    call("trid_stem_conv1_wgrad_f32", _p(images), _p(dy), _p(dw), _p(slabs), Bi, Hi, Wi, stream())
    return dw


def bn_finalize_minmax(partials, M, gamma, beta, running_mean, running_var, relu, bound, momentum=BN_MOMENTUM, eps=BN_EPS, rows_per_part=STATS_ROWS):
    """bn_finalize on (mean, M2, min, max) partials (a Partials record, or a raw [parts][C][4] tensor with `rows_per_part`);
    `bound` (a zeroed amax_slot) receives max|act(BatchNorm(y))|."""
    C = gamma.numel()
    st = BNState(C, gamma)
    if isinstance(partials, Partials):
        partials, rows_per_part = partials.data, partials.rows
    call("trid_bn_finalize_minmax_f32", _p(partials), partials.shape[0], rows_per_part, M, C, _p(gamma), _p(beta),
         _p(running_mean), _p(running_var), momentum, eps, _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift),
         1 if relu else 0, _p(bound), _p(bn_finalize_ws(partials.device)), stream())
    if running_mean is not None:
        note_parameter_write()
    return st


def bn_apply_p16(y, st, bound, relu=True, res=None, res_st=None, bound_res=None, want_mask=False, fmt=1):
    """act(bn(y) (+ res | bn(res))) written as a P16 tensor.  bound: device scalar >= max|bn(y)| (bn_finalize_minmax);
    with a residual, bound_res bounds the residual term and the output's amax scalar is their sum.  res: raw conv
    output (with res_st; fp32, or bf16 in the bf16 mode - like y) or a P16 tensor (identity)."""
    C = y.shape[-1]
    M = y.numel() // C
    out = p16_empty(y.shape, y, fmt)
    mask = torch.empty(((M * C // 4 + 63) // 64) * 4, dtype=torch.int64, device=y.device) if want_mask else None
    res_p16 = isinstance(res, P16)
    osum = amax_slot(y.device) if (res is not None and fmt == 1) else None
    res_fmt = res.fmt if res_p16 else (2 if (res is not None and res.dtype == torch.bfloat16) else 0)
    call("trid_bn_apply_p16_f32", _p(y), 2 if y.dtype == torch.bfloat16 else 0, _p(st.scale), _p(st.shift),
         _p(res.data if res_p16 else res), _p(res_st.scale) if res_st else None, _p(res_st.shift) if res_st else None, res_fmt,
         _p(res.amax) if res_p16 else None, _p(out), fmt, M, C, 1 if relu else 0, _p(mask), _p(bound), _p(bound_res), _p(osum),
         stream())
    o = P16(out, (osum if res is not None else bound) if fmt == 1 else None, fmt)
    return (o, mask) if want_mask else o


# --------------------------------------------------------------------------- eval mode: fused epilogues writing P16
USE_EVAL_P16 = os.environ.get("TRID_EVAL_P16", "1") != "0"  # eval-mode image encoder on the P16 kernels (0: the folded on-the-fly-split path, A/B runs)


def eval_bound_coefs(entries, device):
    """[(w, scale, shift)] per convolution -> device tensor [n, 2]: (max_n |scale_n| ||w_n||_1, max_n |shift_n|), the output
    bound coefficients of the eval-mode epilogues (csrc/gemm_common.h EvalBound).  w: the fp32 filter parameter (any layout
    that keeps an output channel's weights contiguous), scale / shift: the running-statistics BatchNorm coefficients."""
    rows = [[w.data_ptr(), w.shape[0], w.numel() // w.shape[0], sc.data_ptr(), sh.data_ptr()] for (w, sc, sh) in entries]
    table = torch.tensor(rows, dtype=torch.int64, device=device)
    coef = torch.zeros(len(rows), 2, dtype=torch.float32, device=device)
    call("trid_eval_bound_coefs_f32", _p(table), len(rows), _p(coef), stream())
    return coef


def bn_eval_bound(partials, st, relu):
    """Device scalar max|act(bn(y))| of a conv output from its epilogue's (mean, M2, min, max) Partials and the eval-mode
    coefficients `st` (exact: an affine map takes extremes to extremes)."""
    bound = amax_slot(partials.data.device)
    call("trid_bn_eval_bound_f32", _p(partials.data), partials.data.shape[0], partials.data.shape[1], _p(st.scale), _p(st.shift),
         1 if relu else 0, _p(bound), stream())
    return bound


def conv3x3_halo_eval_pool_ok(H, W, Cin, Cout):
    return bool(_query("trid_conv3x3_halo_eval_pool_ok", int(H), int(W), int(Cin), int(Cout)))


def conv3x3_halo_eval_p16(x, w, st, coef, relu=True, pool=False):
    """conv_eval_p16 for the 32 / 64-channel 3x3 convolutions at large maps (the stem's conv2 / conv3, layer1's conv2) on the
    ring-of-rows kernel (csrc/stem_conv.hip): x P16 [B,H,W,Cin], w P16 [Cout, 9*Cin] -> act(bn(conv)) P16 [B,H,W,Cout]; pool:
    followed by the 2x2 average, written pooled ([B,H/2,W/2,Cout]: the stem's conv3; conv3x3_halo_eval_pool_ok)."""
    Bi, H, W, C = x.shape
    N = w.shape[0]
    out = p16_empty((Bi, H // 2, W // 2, N) if pool else (Bi, H, W, N), x.data, 1)
    bound, tmax = amax_slot(x.data.device), amax_slot(x.data.device)
    call("trid_conv3x3_halo_eval_p16", _p(x.data), _p(x.amax), _p(w.data), _p(w.amax), _p(st.scale), _p(st.shift), _p(out), _p(coef), _p(x.tmax),
         _p(bound), _p(tmax), Bi, H, W, C, N, 1 if relu else 0, 1 if pool else 0, stream())
    return P16(out, bound, 1, tmax)


def stem_conv1_eval_p16(images, w, st, coef, img_amax, relu=True):
    """Eval-mode stem conv1 (3 -> 32, stride 2) + BatchNorm + ReLU straight from the NCHW batch -> P16 [B,Ho,Wo,32]; img_amax: device
    scalar >= max|images| (ops.amax)."""
    Bi, Cin, Hi, Wi = images.shape
    if Cin != 3 or w.shape[0] != 32 or not w.is_contiguous() or not images.is_contiguous():
        raise RuntimeError("stem_conv1_eval_p16: a contiguous [B,3,H,W] image batch and contiguous [32,3,3,3] filters are needed")
    out = p16_empty((Bi, (Hi + 1) // 2, (Wi + 1) // 2, 32), images, 1)
    bound, tmax = amax_slot(images.device), amax_slot(images.device)
    call("trid_stem_conv1_eval_p16", _p(images), _p(w), _p(st.scale), _p(st.shift), _p(out), _p(coef), _p(img_amax), _p(bound), _p(tmax),
         Bi, Hi, Wi, 1 if relu else 0, stream())
    return P16(out, bound, 1, tmax)


def conv_eval_pool_ok(H, W, N):
    """Can the tile kernel's eval epilogue write the 2x2 average pool of a 3x3 convolution's activated output directly?"""
    return W % 2 == 0 and H % 2 == 0 and 128 % W == 0 and (128 // W) % 2 == 0 and (H * W) % 128 == 0 and N > 64


def conv_eval_p16(x, w, st, coef, relu=True, res=None, conv3=False, pool=False):
    """Eval-mode conv + BatchNorm(running statistics) (+ residual) (+ ReLU) in ONE kernel, P16 in -> P16 out:
    act(st.scale * conv(x, w) + st.shift (+ res)).  x: P16 [B,H,W,C] / [M,C] with its true maximum `x.tmax`; w: P16 [N, K];
    coef: this convolution's row of eval_bound_coefs; res: P16 [.., N] or None.  The output's scale comes from the bound
    coef[0] * max|x| + coef[1] (+ max|res|) - no pass over the output - and its true maximum is folded into `.tmax`."""
    C = x.shape[-1]
    M = x.data.numel() // C
    N = w.shape[0]
    K = 9 * C if conv3 else C
    shape = tuple(x.shape[:-1]) + (N,)
    if pool:  # (conv3 form only: the output is AvgPool2d(2) of the activated convolution)
        assert conv3 and res is None and conv_eval_pool_ok(x.shape[1], x.shape[2], N)
        shape = (x.shape[0], x.shape[1] // 2, x.shape[2] // 2, N)
    out = p16_empty(shape, x.data, 1)
    dev = x.data.device
    if conv3 and res is None and not pool and USE_HALO_BLOCKS and conv3x3_halo_rows(x.shape[1], x.shape[2], C, N):
        return conv3x3_halo_eval_p16(x, w, st, coef, relu)
    bound, tmax = amax_slot(dev), amax_slot(dev)
    if not conv3 and USE_STREAM and K in (64, 128, 256) and _query("trid_conv1x1_bn_res_p16_ok", int(M), int(N), int(K)):
        call("trid_conv1x1_eval_p16", _p(x.data), _p(x.amax), _p(w.data), _p(w.amax), _p(st.scale), _p(st.shift),
             _p(res.data) if res is not None else None, _p(res.amax) if res is not None else None, _p(out), _p(coef), _p(x.tmax),
             _p(res.tmax) if res is not None else None, _p(bound), _p(tmax), M, N, K, 1 if relu else 0, stream())
        return P16(out, bound, 1, tmax)
    d = GemmDesc()
    d.A, d.B, d.C = _p(x.data), _p(w.data), _p(out)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = C, K, N
    d.batch, d.splits = 1, 1
    d.a_mode, d.b_mode = (A_CONV if conv3 else A_KC), B_KC
    d.alpha = 1.0
    if conv3:
        d.H, d.W, d.Cin = x.shape[1], x.shape[2], C
    d.precision = 16
    d.a_amax, d.b_amax = _p(x.amax), _p(w.amax)
    d.relu = 1 if relu else 0
    d.c_format = 1
    d.col_scale, d.bias = _p(st.scale), _p(st.shift)
    if res is not None:
        d.res_p16, d.res_amax, d.eval_tres = _p(res.data), _p(res.amax), _p(res.tmax)
    d.eval_coef, d.eval_tin = _p(coef), _p(x.tmax)
    d.out_bound, d.out_tmax = _p(bound), _p(tmax)
    d.eval_pool_w = x.shape[2] if pool else 0
    call("trid_gemm_p16", ctypes.addressof(d), -1, stream())
    return P16(out, bound, 1, tmax)


USE_FUSED_EXPAND = __import__("os").environ.get("TRID_FUSED_EXPAND", "1") != "0"  # identity blocks of layer1 / layer2: conv3 + bn3 + residual in one pass (0: A/B runs)


def conv1x1_bn_res_ok(M, N, K):
    """Does the fused conv3 + bn3 + identity pass pay for this shape?  (K = 256 - layer3 - is built and tested but measured
    no faster than GEMM + apply: 104 vs 103 us, its statistics-only pass is compute-bound; tools/exp/fused_bench.py)"""
    return USE_FUSED_EXPAND and USE_STREAM and K <= 128 and bool(_query("trid_conv1x1_bn_res_p16_ok", int(M), int(N), int(K)))


def conv1x1_stats_p16(x, w):
    """BatchNorm partials of the 1x1 convolution y = x . w^T WITHOUT storing y (the first of the two passes of
    conv1x1_bn_res_p16): [steps][N][4] = (mean, M2, min, max), `rows_per_part` set."""
    C = x.shape[-1]
    M = x.data.numel() // C
    N = w.shape[0]
    rows = _query("trid_gemm_p16_stream_stats_rows", int(M), int(N), int(C))
    st = empty(((M + rows - 1) // rows, N, 4), x.data)
    call("trid_gemm_p16_stream", _p(x.data), _p(x.amax), _p(w.data), _p(w.amax), None, N, _p(st), M, N, C, 0, None, stream())
    return Partials(st, rows)


def conv1x1_bn_res_p16(x, w, st, bound, res, relu=True, want_mask=False, keep_y=False):
    """relu(bn(x . w^T) + res) as a P16 tensor, the convolution recomputed inside the pass (csrc/gemm_stream.hip, FUSE):
    x P16 [..., K], w P16 [N, K], st / bound from bn_finalize_minmax on conv1x1_stats_p16's partials, res P16 [..., N] (the
    identity branch).  Returns (out P16, relu mask or None, y fp32 or None); bit-identical to conv_p16 + bn_apply_p16."""
    C = x.shape[-1]
    M = x.data.numel() // C
    N = w.shape[0]
    shape = tuple(x.shape[:-1]) + (N,)
    out = p16_empty(shape, x.data, 1)
    y = empty(shape, x.data) if keep_y else None
    mask = torch.empty(((M * N // 4 + 63) // 64) * 4, dtype=torch.int64, device=x.data.device) if want_mask else None
    osum = amax_slot(x.data.device)
    call("trid_conv1x1_bn_res_p16", _p(x.data), _p(x.amax), _p(w.data), _p(w.amax), _p(y), _p(st.scale), _p(st.shift), _p(res.data), _p(res.amax),
         _p(out), _p(bound), _p(res.amax), _p(osum), _p(mask), M, N, C, 1 if relu else 0, stream())
    return P16(out, osum, 1), mask, y


def bn_apply_pool2_p16(y, st, bound, relu=True, fmt=1):
    """avgpool2(act(bn(y))) -> P16; y: raw conv output fp32 (st given) or a P16 tensor (st None: plain pooling, the
    output keeps the input's scale)."""
    src = y.data if isinstance(y, P16) else y
    Bi, H, W, C = src.shape
    out = p16_empty((Bi, H // 2, W // 2, C), src, fmt)
    call("trid_bn_apply_pool2_p16_f32", _p(src), _p(st.scale) if st else None, _p(st.shift) if st else None,
         y.fmt if isinstance(y, P16) else (2 if y.dtype == torch.bfloat16 else 0), _p(y.amax) if isinstance(y, P16) else None, _p(out), fmt, Bi, H, W, C,
         1 if (relu and st is not None) else 0, _p(bound), stream())
    # (plain pooling of a P16 tensor: an average never exceeds the maximum - the input's true maximum still bounds the output)
    return P16(out, bound if fmt == 1 else None, fmt, y.tmax if (isinstance(y, P16) and st is None and fmt == 1) else None)


def bn_bwd_p16(g, y, st, mask_mode, act=None, pooled=False, want_dres=False, fmt=1, presummed=None):
    """BatchNorm backward with dy written as a P16 tensor: the reduce pass also bounds max|dy| (triangle inequality
    over per-channel maxima), the apply pass scales by that bound.  Returns (dy P16, dgamma, dbeta, dres).
    g: fp32, or (fmt 2) a bf16 tensor - dres, its masked copy, has g's dtype.
    presummed: the BnBwdSums the GEMM that produced g filled - the reduce pass over g and y is skipped, only its fold runs."""
    Bi, H, W, C = y.shape
    dg = empty((2, C), y)
    dgamma, dbeta = dg[0], dg[1]
    bound = None
    g_fmt = 2 if g.dtype == torch.bfloat16 else 0
    y_fmt = 2 if y.dtype == torch.bfloat16 else 0
    if presummed is not None:
        assert fmt == 1 and g_fmt == 0 and y_fmt == 0 and not pooled and presummed.y is y
        assert (mask_mode == 3 and presummed.mask is act) if presummed.mask is not None else mask_mode == (1 if presummed.relu else 0)
        bound = amax_slot(y.device)
        call("trid_bn_bwd_final_f32", _p(presummed.ws), _p(presummed.ws2), presummed.M, C, _p(st.scale), _p(dgamma), _p(dbeta), _p(bound), stream())
    elif fmt == 1:
        ws = _bn_ws(C, y)
        assert g_fmt == 0 and y_fmt == 0
        bound = amax_slot(y.device)
        call("trid_bn_bwd_reduce_bound_f32", _p(g), _p(y), _p(act), _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift),
             mask_mode, 1 if pooled else 0, Bi, H, W, C, _p(dgamma), _p(dbeta), _p(ws), _p(bound), stream())
    else:
        ws = _bn_ws(C, y)
        call("trid_bn_bwd_reduce_g_f32", _p(g), g_fmt, _p(y), y_fmt, _p(act), _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift),
             mask_mode, 1 if pooled else 0, Bi, H, W, C, _p(dgamma), _p(dbeta), _p(ws), stream())
    dy = p16_empty(y.shape, y, fmt)
    dres = torch.empty(y.shape, dtype=g.dtype, device=y.device) if want_dres else None
    call("trid_bn_bwd_apply_p16_f32", _p(g), g_fmt, _p(y), y_fmt, _p(act), _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift),
         _p(dgamma), _p(dbeta), mask_mode, 1 if pooled else 0, Bi, H, W, C, _p(dy), fmt, _p(dres), _p(bound), stream())
    return P16(dy, bound, fmt), dgamma, dbeta, dres


USE_BN_DUAL = os.environ.get("TRID_BN_DUAL", "1") != "0"  # bn3 + downsample BatchNorm backward in one reduce / one apply pass (0: A/B runs)


def bn_bwd_dual_ok(g, y1, y2, fmt=1):
    C = y1.shape[-1]
    CQ = C // 4
    return (USE_BN_DUAL and fmt == 1 and g.dtype == torch.float32 and y1.dtype == torch.float32 and y2.dtype == torch.float32 and g.is_contiguous()
            and tuple(y1.shape) == tuple(y2.shape) == tuple(g.shape) and C % 32 == 0 and (256 % CQ == 0 or CQ % 256 == 0))


def bn_bwd_dual_p16(g, bits, y1, st1, y2, st2):
    """The backward of the two BatchNorm layers a downsample block's output gradient feeds (bn3 and the downsample branch's), both
    behind the block's ReLU mask `bits`: one reduce pass and one apply pass over (g, y1, y2).  Returns
    (dy1 P16, dgamma1, dbeta, dy2 P16, dgamma2, dbeta2) - what two bn_bwd_p16(..., 3, act=bits) calls return."""
    C = y1.shape[-1]
    M = y1.numel() // C
    dg = empty((4, C), y1)
    key = ("dual", C, y1.device, stream(y1.device))
    ws = _ws_cache.get(key)
    if ws is None:
        ws = empty((2 * L.load().trid_bn_bwd_ws_floats(C),), y1)
        _ws_cache[key] = ws
    b1, b2 = amax_slot(y1.device), amax_slot(y1.device)
    call("trid_bn_bwd_dual_reduce_bound_f32", _p(g), _p(bits), _p(y1), _p(y2), _p(st1.mean), _p(st1.invstd), _p(st1.scale), _p(st2.mean),
         _p(st2.invstd), _p(st2.scale), M, C, _p(dg[0]), _p(dg[1]), _p(dg[2]), _p(dg[3]), _p(ws), _p(b1), _p(b2), stream())
    dy1, dy2 = p16_empty(y1.shape, y1, 1), p16_empty(y2.shape, y2, 1)
    call("trid_bn_bwd_dual_apply_p16_f32", _p(g), _p(bits), _p(y1), _p(y2), _p(st1.mean), _p(st1.invstd), _p(st1.scale), _p(st2.mean),
         _p(st2.invstd), _p(st2.scale), _p(dg[0]), _p(dg[1]), _p(dg[2]), M, C, _p(dy1), _p(dy2), _p(b1), _p(b2), stream())
    return P16(dy1, b1, 1), dg[0], dg[2], P16(dy2, b2, 1), dg[1], dg[3]


def wgrad_p16(dy, x, conv=None, alpha=1.0):
    """Weight gradient on P16 operands: dW [N, J] = dy[M,N]^T @ X[M,J] with X = x [M, C] (1x1) or the 3x3 gather of the
    NHWC image x [B,H,W,C] (conv=(H,W,C), J = 9*C).  Split over the pixels, slabs folded by trid_slab_reduce_f32."""
    N = dy.shape[-1]
    M = dy.data.numel() // N
    C = x.shape[-1]
    J = 9 * C if conv is not None else C
    out = empty((N, J), dy.data)
    tiles = ((N + 127) // 128) * ((J + 127) // 128)
    splits = _wgrad_splits(tiles, M, slots=768 if N <= 64 else 512)
    d = GemmDesc()
    d.A, d.B = _p(dy.data), _p(x.data)
    d.M, d.N, d.K = N, J, M
    d.lda, d.ldb, d.ldc = N, C, J
    d.batch, d.splits = 1, splits
    d.a_mode, d.b_mode = A_MC, (B_CONV if conv is not None else B_NC)
    d.alpha = alpha
    if conv is not None:
        d.H, d.W, d.Cin = conv
    d.precision = 16 if dy.fmt == 1 else 1
    d.a_amax, d.b_amax = _p(dy.amax), _p(x.amax)
    if splits == 1:
        d.C = _p(out)
        call("trid_gemm_p16_wgrad", ctypes.addressof(d), stream())
        return out
    slab = empty((splits, N, J), dy.data)
    d.C, d.strideSplit = _p(slab), N * J
    call("trid_gemm_p16_wgrad", ctypes.addressof(d), stream())
    call("trid_slab_reduce_f32", _p(slab), _p(out), N * J, splits, N * J, 0, stream())
    return out


USE_SKINNY = os.environ.get("TRID_SKINNY_GEMM", "1") != "0"  # M <= 128 linears on csrc/skinny_gemm.hip (0: A/B runs)


# raw gemm() calls (the attention pool's batched products): only the per-HEAD ones with all 128 batch rows and a long reduction
# (o = Z Wv^T, dq = dU Wk^T: 59 -> 34 us).  The per-IMAGE ones (32 heads x 196 tokens: M = 32, or K = 196) measured SLOWER on
# this kernel (S 68 -> 86 us, Z 60 -> 186 us, tools/exp/attn_bench.py): thousands of workgroups with half their waves idle
SKINNY_MIN_K_BATCHED = 512
SKINNY_MIN_M_BATCHED = 65


def _skinny_ok(M, N, K, lda, ldb_k_contig, prec, min_k=256):
    """The batch-sized GEMMs (attention-pool projections, embedding layers, their data gradients): one row of output tiles
    on a tiled kernel - they run on the reduction-split, weight-streaming kernel instead (exact fp32)."""
    return (USE_SKINNY and prec is None and M <= 128 and K >= min_k and K % 4 == 0 and N >= 32 and N <= 8192 and lda % 4 == 0
            and (ldb_k_contig is None or ldb_k_contig % 4 == 0))


def skinny_gemm(a, lda, b, ldb, b_mode, out, ldc, M, N, K, bias=None, alpha=1.0, accumulate=False, batch=1, strideA=0, strideB=0, strideC=0,
                strideBias=0):
    call("trid_skinny_gemm_f32", _p(a), lda, _p(b), ldb, b_mode, _p(out), ldc, _p(bias), M, N, K, alpha,
         1 if accumulate else 0, batch, strideA, strideB, strideC, strideBias, stream())


def linear(x, w, bias=None, out=None, alpha=1.0, accumulate=False, prec=None, aa=None, ba=None, relu=False):
    """y[M,N] = alpha * x[M,K] @ w[N,K]^T + bias.  x may be a strided row view.
    prec / aa / ba (here and below): GEMM arithmetic override and the two operands' amax device scalars."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = empty((M, N), x)
    if not relu and x.stride(1) == 1 and w.stride(1) == 1 and _skinny_ok(M, N, K, x.stride(0), w.stride(0), prec) and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0:
        skinny_gemm(x, x.stride(0), w, w.stride(0), B_KC, out, out.stride(0), M, N, K, bias=bias, alpha=alpha, accumulate=accumulate)
        return out
    gemm(x, w, out, M, N, K, x.stride(0), w.stride(0), out.stride(0), alpha=alpha, accumulate=accumulate, bias=bias,
         precision=prec, a_amax=aa, b_amax=ba, relu=relu)
    return out


def matmul_nn(a, b, out=None, alpha=1.0, accumulate=False, prec=None, aa=None, ba=None):
    """out[M,N] = a[M,K] @ b[K,N] (b rows N-contiguous).  Skinny outputs with a long
    K (loss gradients: [B,K_queue]x[K_queue,C]) are split over K into slabs so the
    launch fills the chip instead of running a serial K loop on a handful of CUs."""
    M, K = a.shape
    N = b.shape[1]
    if out is None:
        out = empty((M, N), a)
    if a.stride(1) == 1 and b.stride(1) == 1 and _skinny_ok(M, N, K, a.stride(0), None, prec) and a.data_ptr() % 16 == 0:
        skinny_gemm(a, a.stride(0), b, b.stride(0), B_NC, out, out.stride(0), M, N, K, alpha=alpha, accumulate=accumulate)
        return out
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    splits = min(K // 128, (256 + tiles - 1) // tiles) if (tiles < 64 and K >= 1024 and out.stride(0) == N) else 1
    if splits <= 1:
        gemm(a, b, out, M, N, K, a.stride(0), b.stride(0), out.stride(0), b_mode=B_NC, alpha=alpha,
             accumulate=accumulate, precision=prec, a_amax=aa, b_amax=ba)
        return out
    slab = empty((splits, M, N), a)
    gemm(a, b, slab, M, N, K, a.stride(0), b.stride(0), N, b_mode=B_NC, alpha=alpha, splits=splits, strideSplit=M * N,
         precision=prec, a_amax=aa, b_amax=ba)
    call("trid_slab_reduce_f32", _p(slab), _p(out), M * N, splits, M * N, 1 if accumulate else 0, stream())
    return out


_WGRAD_SPLITS_OVERRIDE = None  # kernel experiments


def _wgrad_splits(tiles, K, slots=512):
    """Split-K factor for a weight-gradient GEMM (K = pixels of the batch): ONE full round of resident workgroups
    (`slots` = 256 CUs x 2) when the tile count allows a >= 90 % full one, else two rounds; >= 512 reduction rows per
    split.  Counts >= 8 are multiples of 8: the P16 kernels give every XCD whole splits (the tiles of a split then share
    one L2), which balances only then.  Fewer, longer splits also mean fewer slabs to fold.  (Sweep on the RN50 layer
    shapes at B=128: tools/exp/wgrad_splits.py, profiles/r03i_wgrad_split_sweep.txt.)"""
    if _WGRAD_SPLITS_OVERRIDE is not None:
        return _WGRAD_SPLITS_OVERRIDE
    tiles = max(tiles, 1)
    cap = max(1, K // 512)

    def fit(n):
        s = min(cap, n // tiles)
        return s - s % 8 if s >= 8 else s

    s = fit(slots)
    if s * tiles < 0.9 * slots:
        s = max(s, fit(2 * slots))
    return max(1, s)


def matmul_tn(a, b, out=None, alpha=1.0, prec=None, aa=None, ba=None):
    """out[Ma,Nb] = a[K,Ma]^T @ b[K,Nb]  (weight-gradient form, split-K over K)."""
    K, Ma = a.shape
    Nb = b.shape[1]
    if out is None:
        out = empty((Ma, Nb), a)
    tiles = ((Ma + 127) // 128) * ((Nb + 127) // 128)
    splits = _wgrad_splits(tiles, K)
    if splits == 1 or out.stride(0) != Nb:
        gemm(a, b, out, Ma, Nb, K, a.stride(0), b.stride(0), out.stride(0), a_mode=A_MC, b_mode=B_NC, alpha=alpha,
             precision=prec, a_amax=aa, b_amax=ba)
        return out
    slab = empty((splits, Ma, Nb), a)
    gemm(a, b, slab, Ma, Nb, K, a.stride(0), b.stride(0), Nb, a_mode=A_MC, b_mode=B_NC, alpha=alpha, splits=splits,
         strideSplit=Ma * Nb, precision=prec, a_amax=aa, b_amax=ba)
    call("trid_slab_reduce_f32", _p(slab), _p(out), Ma * Nb, splits, Ma * Nb, 0, stream())
    return out


def stats_buffer(M, N, like):
    return empty(((M + STATS_ROWS - 1) // STATS_ROWS, N, 2), like)


def conv1x1(x, w, stats=False, bias=None, relu=False, residual=None, prec=None, aa=None, ba=None):
    """x [B,H,W,C] NHWC (or [M,C]), w [N,C] -> y [.., N]; optional BN partials (training) or the
    fused eval epilogue: + bias[N] (+ residual [.., N]) then ReLU."""
    C = x.shape[-1]
    M = x.numel() // C
    N = w.shape[0]
    y = empty(x.shape[:-1] + (N,), x)
    st = stats_buffer(M, N, x) if stats else None
    gemm(x, w, y, M, N, C, C, w.stride(0), N, stats=st, bias=bias, relu=relu, residual=residual, ldres=N,
         precision=prec, a_amax=aa, b_amax=ba)
    return (y, st) if stats else y


def conv3x3(x, w, stats=False, bias=None, relu=False, prec=None, aa=None, ba=None):
    """x [B,H,W,C] NHWC, w [N, 9*C] (tap-major, channel-minor = OHWI) -> y [B,H,W,N]."""
    Bi, H, W, C = x.shape
    N = w.shape[0]
    M = Bi * H * W
    y = empty((Bi, H, W, N), x)
    st = stats_buffer(M, N, x) if stats else None
    gemm(x, w, y, M, N, 9 * C, C, 9 * C, N, a_mode=A_CONV, stats=st, conv=(H, W, C), bias=bias, relu=relu,
         precision=prec, a_amax=aa, b_amax=ba)
    return (y, st) if stats else y


def fold_bn(w2d, st):
    """Eval-mode BatchNorm folded into the preceding conv: (w * scale[:,None], shift) so that
    relu(bn(conv(x, w))) == relu(conv(x, w') + shift), one GEMM with a bias + ReLU epilogue."""
    return rowscale_add(st.scale, w2d), st.shift


def conv1x1_wgrad(dy, x, prec=None, aa=None, ba=None):
    """dW [N,C] = dy[M,N]^T @ x[M,C]."""
    N, C = dy.shape[-1], x.shape[-1]
    return matmul_tn(dy.reshape(-1, N), x.reshape(-1, C), prec=prec, aa=aa, ba=ba)


def conv3x3_wgrad(dy, x, prec=None, aa=None, ba=None):
    """dW [N, 9*C] for the 3x3/s1/p1 conv; dy [B,H,W,N], x [B,H,W,C]."""
    Bi, H, W, C = x.shape
    N = dy.shape[-1]
    M = Bi * H * W
    J = 9 * C
    out = empty((N, J), x)
    tiles = ((N + 127) // 128) * ((J + 127) // 128)
    splits = _wgrad_splits(tiles, M)
    if splits == 1:
        gemm(dy, x, out, N, J, M, N, C, J, a_mode=A_MC, b_mode=B_CONV, conv=(H, W, C), precision=prec, a_amax=aa, b_amax=ba)
        return out
    slab = empty((splits, N, J), x)
    gemm(dy, x, slab, N, J, M, N, C, J, a_mode=A_MC, b_mode=B_CONV, conv=(H, W, C), splits=splits, strideSplit=N * J,
         precision=prec, a_amax=aa, b_amax=ba)
    call("trid_slab_reduce_f32", _p(slab), _p(out), N * J, splits, N * J, 0, stream())
    return out


def weight_transpose(w, N, T, C, flip):
    """w [N][T][C] -> [C][T][N] (taps reversed when flip): dgrad weights."""
    wt = empty((C, T * N), w)
    call("trid_weight_transpose_f32", _p(w), _p(wt), N, T, C, 1 if flip else 0, stream())
    return wt


def stem_im2col(img, ldcol=28):
    Bi, Cin, H, W = img.shape
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    col = empty((Bi * Ho * Wo, ldcol), img)
    call("trid_stem_im2col_f32", _p(img), _p(col), Bi, Cin, H, W, Ho, Wo, ldcol, stream())
    return col, Ho, Wo


# --------------------------------------------------------------------------- BatchNorm
class BNState:
    """Per-layer BatchNorm coefficients produced by the forward pass."""

    __slots__ = ("mean", "invstd", "scale", "shift")

    def __init__(self, C, like):
        buf = empty((4, C), like)
        self.mean, self.invstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]


def bn_finalize(partials, M, gamma, beta, running_mean, running_var, momentum=BN_MOMENTUM, eps=BN_EPS):
    C = gamma.numel()
    st = BNState(C, gamma)
    rows = STATS_ROWS
    if isinstance(partials, Partials):  # (bf16 mode: conv_p16's (mean, M2) partials with their tile height)
        partials, rows = partials.data, partials.rows
    call("trid_bn_finalize_f32", _p(partials), partials.shape[0], rows, M, C, _p(gamma), _p(beta),
         _p(running_mean), _p(running_var), momentum, eps, _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift),
         _p(bn_finalize_ws(partials.device)), stream())
    if running_mean is not None:
        note_parameter_write()
    return st


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps=BN_EPS):
    C = gamma.numel()
    st = BNState(C, gamma)
    call("trid_bn_eval_coeffs_f32", _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, _p(st.scale),
         _p(st.shift), C, stream())
    return st


def bn_apply(y, st, relu=True, res=None, res_st=None, out=None, want_mask=False, amax=None):
    """out = act(bn(y) (+ res | bn(res))).  want_mask: also return the 1-bit-per-element ReLU mask
    (int64 words) the backward pass uses instead of re-reading `out` (bn_bwd mask_mode 3).
    amax: a zeroed device scalar (amax_slot) that receives max|out|."""
    C = y.shape[-1]
    M = y.numel() // C
    if out is None:
        out = torch.empty_like(y)
    mask = torch.empty(((M * C // 4 + 63) // 64) * 4, dtype=torch.int64, device=y.device) if want_mask else None
    call("trid_bn_apply_f32", _p(y), _p(st.scale), _p(st.shift), _p(res), _p(res_st.scale) if res_st else None,
         _p(res_st.shift) if res_st else None, _p(out), M, C, 1 if relu else 0, _p(mask), _p(amax), stream())
    return (out, mask) if want_mask else out


def bn_apply_pool2(y, st, relu=True, amax=None):
    """avgpool2(act(bn(y))) ; st=None -> plain 2x2 average pooling."""
    Bi, H, W, C = y.shape
    out = empty((Bi, H // 2, W // 2, C), y)
    call("trid_bn_apply_pool2_f32", _p(y), _p(st.scale) if st else None, _p(st.shift) if st else None, _p(out), Bi, H, W,
         C, 1 if (relu and st is not None) else 0, _p(amax), stream())
    return out


def avgpool2_bwd(g, dx=None, accumulate=False):
    """g fp32 or bf16 (dx in the same dtype)."""
    Bi, Ho, Wo, C = g.shape
    if dx is None:
        dx = empty((Bi, Ho * 2, Wo * 2, C), g, dtype=g.dtype)
    assert dx.dtype == g.dtype
    call("trid_avgpool2_bwd_f32", _p(g), _p(dx), Bi, Ho * 2, Wo * 2, C, 1 if accumulate else 0,
         2 if g.dtype == torch.bfloat16 else 0, stream())
    return dx


_ws_cache = {}


def _bn_ws(C, like):
    # one workspace per (channels, device, STREAM): two backward passes on different streams must not share it
    key = (C, like.device, stream(like.device))
    ws = _ws_cache.get(key)
    if ws is None:
        ws = empty((L.load().trid_bn_bwd_ws_floats(C),), like)
        _ws_cache[key] = ws
    return ws


def bn_bwd(g, y, st, gamma_like, mask_mode, act=None, pooled=False, want_dres=False, amax=None):
    """BatchNorm backward.  g: dL/d(out) ([B,H/2,W/2,C] when pooled).  Returns
    (dy, dgamma, dbeta, dres).  amax: zeroed device scalar that receives max|dy|."""
    Bi, H, W, C = y.shape
    dg = empty((2, C), y)
    dgamma, dbeta = dg[0], dg[1]
    ws = _bn_ws(C, y)
    call("trid_bn_bwd_reduce_f32", _p(g), _p(y), _p(act), _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift),
         mask_mode, 1 if pooled else 0, Bi, H, W, C, _p(dgamma), _p(dbeta), _p(ws), stream())
    dy = torch.empty_like(y)
    dres = torch.empty_like(y) if want_dres else None
    call("trid_bn_bwd_apply_f32", _p(g), _p(y), _p(act), _p(st.mean), _p(st.invstd), _p(st.scale), _p(st.shift),
         _p(dgamma), _p(dbeta), mask_mode, 1 if pooled else 0, Bi, H, W, C, _p(dy), _p(dres), _p(amax), stream())
    return dy, dgamma, dbeta, dres


# --------------------------------------------------------------------------- small ops
def colsum(x2d, out=None, accumulate=False):
    M, N = x2d.shape
    if out is None:
        out = empty((N,), x2d)
    call("trid_colsum_f32", _p(x2d), _p(out), M, N, x2d.stride(0), 1 if accumulate else 0, stream())
    return out


def softmax_rows_(s, n):
    rows, ld = s.numel() // s.shape[-1], s.shape[-1]
    call("trid_softmax_rows_f32", _p(s), rows, n, ld, stream())
    return s


def softmax_rows_bwd(p, dp, n, out=None):
    rows, ld = p.numel() // p.shape[-1], p.shape[-1]
    if out is None:
        out = torch.empty_like(p)
    call("trid_softmax_rows_bwd_f32", _p(p), _p(dp), _p(out), rows, n, ld, stream())
    return out


def l2norm_rows(x, eps=1e-12):
    rows, C = x.shape
    y = torch.empty_like(x)
    inv = empty((rows,), x)
    call("trid_l2norm_rows_f32", _p(x), _p(y), _p(inv), rows, C, eps, stream())
    return y, inv


def l2norm_rows_bwd(dy, y, inv, dx=None, accumulate=False):
    rows, C = y.shape
    if dx is None:
        dx = torch.empty_like(y)
    call("trid_l2norm_rows_bwd_f32", _p(dy), _p(y), _p(inv), _p(dx), rows, C, 1 if accumulate else 0, stream())
    return dx


def rowdot(x, y):
    rows, C = x.shape
    out = empty((rows,), x)
    call("trid_rowdot_f32", _p(x), _p(y), _p(out), rows, C, stream())
    return out


def rowscale_add(s, y, dx=None, accumulate=False):
    rows, C = y.shape
    if dx is None:
        dx = torch.empty_like(y)
    call("trid_rowscale_add_f32", _p(s), _p(y), _p(dx), rows, C, 1 if accumulate else 0, stream())
    return dx


def sum_to(x, out, scale=1.0, accumulate=False):
    call("trid_sum_f32", _p(x), _p(out), x.numel(), scale, 1 if accumulate else 0, stream())
    return out
