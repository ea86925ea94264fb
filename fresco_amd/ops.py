"""Tensor-level entry points over the C ABI (include/fresco_hip.h).

PyTorch is used for device memory, the current HIP stream and dtype casts only; every arithmetic op
below is one call into libfresco_hip.so.  All functions require HIP ("cuda") tensors and raise on
CPU tensors -- there is no fallback path.
"""
import math

import torch

from . import _lib
from ._lib import FrescoHipError


try:  # raw handle of the current HIP stream without building a torch.cuda.Stream object (~10x cheaper per call)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover
    _raw_stream = None


def _stream():
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise FrescoHipError("fresco_amd operators run on the GPU only (got a %s tensor); "
                                 "there is no CPU fallback" % t.device)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _f32c(t):
    return t.to(torch.float32).contiguous()


def _rows(t):
    """(tensor, row stride in elements, rows) for a (..., C) tensor whose rows are uniformly strided
    (dense, or a column slice of a dense (..., k*C) tensor such as a fused projection output);
    anything else is made dense first."""
    C = t.shape[-1]
    if t.is_contiguous() and C % 8 == 0 and t.dim() >= 2 and t.data_ptr() % 16 == 0:  # the common case
        return t, C, t.numel() // C
    ok = t.stride(-1) == 1 and t.dim() >= 2
    if ok:
        ld = t.stride(-2)
        exp = ld
        for d in range(t.dim() - 2, -1, -1):  # leading dims must nest without gaps
            if t.shape[d] != 1 and t.stride(d) != exp:
                ok = False
                break
            exp *= t.shape[d]
        ok = ok and ld >= C and ld % 8 == 0 and t.data_ptr() % 16 == 0
    if not ok:
        t = t.contiguous()
        ld = C
    return t, ld, t.numel() // C


class Workspace:
    """Grow-only device scratch buffer (the C ABI never allocates)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        return self.buf


class OptContext:
    """A fresco_ctx (include/fresco_hip.h): the side stream + events the two-pipeline form of fresco_opt_run_ctx uses.  Owned
    by the caller -- the library keeps no process-wide stream table; one context per host thread and device."""

    def __init__(self):
        import ctypes
        p = ctypes.c_void_p()
        _lib.check(_lib.load().fresco_ctx_create(ctypes.byref(p)), "fresco_ctx_create")
        self.ptr = p

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                _lib.load().fresco_ctx_destroy(self.ptr)
                self.ptr = None
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


class _PerThreadContexts:
    def __init__(self):
        import threading
        self._tls = threading.local()

    def get(self, device):
        table = getattr(self._tls, "table", None)
        if table is None:
            table = self._tls.table = {}
        ctx = table.get(device)
        if ctx is None:
            ctx = table[device] = OptContext()
        return ctx


_opt_contexts = _PerThreadContexts()


class _PerStreamWorkspace:
    """Default scratch when the caller passes none: one grow-only buffer per (thread, device, stream), so
    that concurrent callers on different streams or threads never share scratch memory."""

    def __init__(self):
        import threading

        self._tls = threading.local()

    def get(self, nbytes, device):
        table = getattr(self._tls, "table", None)
        if table is None:
            table = self._tls.table = {}
        key = (device, _stream())
        ws = table.get(key)
        if ws is None:
            ws = table[key] = Workspace()
        return ws.get(nbytes, device)


_default_ws = _PerStreamWorkspace()


# ---------------------------------------------------------------------------------------------
# fused linear projections
# ---------------------------------------------------------------------------------------------
def linear_supported(K, N, dtype):
    """shapes / dtype fresco_linear takes (SD-1.5 up_blocks.2/3 attention widths); others keep torch's GEMM"""
    return dtype == torch.float16 and K in (320, 640) and N % 64 == 0


def linear(x, weights, biases=None, outs=None, x_rows=None, x_rows_trusted=False):
    """out_j = x W_j^T (+ b_j) for the 1..3 weight matrices in `weights` (each (N, K) fp16, read where it lives --
    nothing is stacked or cached), x (..., K) fp16 read once; `biases`: None or a list of (N,) fp16 / None.
    Returns len(weights) tensors shaped x.shape[:-1] + (N,); `outs` may supply them (dense in the last dim,
    uniformly strided rows -- e.g. the two halves of a fused K|V buffer).
    x_rows (int32, (M',)): gathered form -- output row m is the projection of x's flat row x_rows[m]; the outputs are
    (M', N).  The table is bounds-checked once (a host synchronisation; cached with the tensor) unless the caller vouches
    for it (`x_rows_trusted`: tables derived from a mask's non-zero positions)."""
    if torch.is_tensor(weights):
        weights = [weights]
    weights = list(weights)
    nw = len(weights)
    if biases is None:
        biases = [None] * nw
    elif torch.is_tensor(biases):
        biases = [biases]
    _need_gpu(x, *weights, *[b for b in biases if b is not None])
    K = x.shape[-1]
    N = weights[0].shape[0]
    if not 1 <= nw <= 3 or len(biases) != nw:
        raise ValueError("linear: 1..3 weights and as many biases (or None)")
    for W in weights:
        if W.dtype != torch.float16 or x.dtype != torch.float16 or tuple(W.shape) != (N, K) or not W.is_contiguous():
            raise ValueError("linear: x (...,K) and every W (N,K) must be contiguous fp16 with matching shapes")
    for b in biases:
        if b is not None and (b.dtype != torch.float16 or tuple(b.shape) != (N,) or not b.is_contiguous()):
            raise ValueError("linear: a bias must be a contiguous (N,) fp16 tensor")
    x2, x_ld, M = _rows(x)
    if x_rows is not None:
        _need_gpu(x_rows)
        if x_rows.dtype != torch.int32 or x_rows.dim() != 1 or not x_rows.is_contiguous():
            raise TypeError("linear: x_rows must be a contiguous 1-D int32 tensor")
        n_x = M

        def run(x_rows=x_rows, n_x=n_x):
            lo, hi = int(x_rows.min()), int(x_rows.max())
            if lo < 0 or hi >= n_x:
                raise ValueError("linear: x_rows address rows %d..%d but x has %d rows" % (lo, hi, n_x))
        if not x_rows_trusted:
            _check_once(("xrows", x_rows.data_ptr(), x_rows.numel(), x_rows._version, n_x), x_rows, run)
        M = x_rows.numel()
    if outs is None:
        shape = (M, N) if x_rows is not None else x.shape[:-1] + (N,)
        outs = [torch.empty(shape, dtype=torch.float16, device=x.device) for _ in range(nw)]
    ptrs, lds = [], []
    for t in outs:
        t2, ld, rows = _rows(t)
        if t2 is not t or rows != M or t.shape[-1] != N or t.dtype != torch.float16:
            raise ValueError("linear: every output must be a (M,N) fp16 tensor with uniformly strided rows")
        ptrs.append(t.data_ptr())
        lds.append(ld)
    wp = [W.data_ptr() for W in weights] + [None] * (3 - nw)
    bp = [_ptr(b) for b in biases] + [None] * (3 - nw)
    while len(ptrs) < 3:
        ptrs.append(None)
        lds.append(0)
    if x_rows is not None:
        rc = _lib.load().fresco_linear_rows(x2.data_ptr(), x_ld, x_rows.data_ptr(), wp[0], wp[1], wp[2], bp[0], bp[1],
                                            bp[2], ptrs[0], ptrs[1], ptrs[2], lds[0], lds[1], lds[2], nw, M, N, K,
                                            _stream())
    else:
        rc = _lib.load().fresco_linear(x2.data_ptr(), x_ld, wp[0], wp[1], wp[2], bp[0], bp[1], bp[2], ptrs[0], ptrs[1],
                                       ptrs[2], lds[0], lds[1], lds[2], nw, M, N, K, _stream())
    _lib.check(rc, "fresco_linear(M=%d,N=%d,K=%d,nw=%d)" % (M, N, K, nw))
    return outs


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------
def attention(q, k, v, heads, scale, *, kv_rows=None, n_groups=None, M=None, group_rows=None,
              diag_bias=0.0, workspace=None):
    """softmax(scale * q k^T + diag_bias*I) v with grouped keys (fresco_attn_fwd).

    q: (B, Lq, C) fp16.  k, v: (..., C) fp16 whose leading dims flatten to rows.
    Default grouping: one key group per batch element (plain attention): k, v are (B, Lk, C).
    """
    _need_gpu(q, k, v, kv_rows)
    if q.dtype != torch.float16 or k.dtype != torch.float16 or v.dtype != torch.float16:
        raise TypeError("fresco_amd.attention: fp16 tensors required (got %s/%s/%s); the SD-1.5 FRESCO "
                        "pipeline runs its UNet in fp16" % (q.dtype, k.dtype, v.dtype))
    B, Lq, C = q.shape
    D = C // heads
    if D * heads != C:
        raise ValueError("channels %d not divisible by heads %d" % (C, heads))
    q, q_ld, _ = _rows(q)
    k, kv_ld, k_rows = _rows(k)
    v, v_ld, v_rows = _rows(v)
    if v_ld != kv_ld:
        v = v.contiguous()
        k = k.contiguous()
        kv_ld = C
    if n_groups is None:
        if k_rows != v_rows or k_rows % B != 0:
            raise ValueError("k and v must have the same number of rows, a multiple of the batch")
        n_groups = B
        group_rows = k_rows // B
        M = group_rows
    if kv_rows is not None:
        if kv_rows.dtype != torch.int32:
            raise TypeError("kv_rows must be int32")
        kv_rows = kv_rows.contiguous()
        M = kv_rows.numel()
        limit = min(k_rows, v_rows)

        def run(kv_rows=kv_rows, limit=limit):
            lo, hi = int(kv_rows.min()), int(kv_rows.max())
            if lo < 0 or hi + (n_groups - 1) * group_rows >= limit:
                raise ValueError("kv_rows address rows %d..%d (+%d groups of %d) but k/v have %d rows"
                                 % (lo, hi, n_groups, group_rows, limit))
        _check_once(("rows", kv_rows.data_ptr(), M, kv_rows._version, n_groups, int(group_rows), limit), kv_rows, run)
    if kv_rows is None and ((n_groups - 1) * group_rows + M > min(k_rows, v_rows)):
        raise ValueError("k/v have %d/%d rows, grouping needs %d" % (k_rows, v_rows, (n_groups - 1) * group_rows + M))
    lib = _lib.load()
    ws_bytes = lib.fresco_attn_workspace_bytes(n_groups, heads, M, D)
    ws = (workspace or _default_ws).get(ws_bytes, q.device)
    out = torch.empty((B, Lq, C), dtype=q.dtype, device=q.device)
    rc = lib.fresco_attn_fwd_ld(q.data_ptr(), k.data_ptr(), v.data_ptr(), _ptr(kv_rows), out.data_ptr(),
                                ws.data_ptr(), ws.numel(), B, heads, Lq, D, n_groups, M, group_rows,
                                float(scale), float(diag_bias), q_ld, kv_ld, _stream())
    _lib.check(rc, "fresco_attn_fwd(B=%d,H=%d,Lq=%d,D=%d,groups=%d,M=%d)" % (B, heads, Lq, D, n_groups, M))
    return out


def attention_kvproj_supported(heads, head_dim, in_features):
    return bool(_lib.load().fresco_attn_kvproj_supported(int(heads), int(head_dim), int(in_features)))


def attention_kvproj(q, hidden, x_rows, w_k, w_v, heads, scale, n_groups, M, workspace=None):
    """Cross-frame pass whose K | V projection of the selected rows is fused into the key pack (fresco_attn_fwd_kvproj).

    q (B, Lq, C) fp16; hidden (..., K_in) fp16 whose leading dims flatten to rows; x_rows int32 (n_groups * M): key m of
    group g = row x_rows[g * M + m] of `hidden` (in range: the caller's table, checked once where it is built);
    w_k, w_v (C, K_in) fp16 contiguous -- the live weights of bias-free nn.Linear modules.  Returns (B, Lq, C)."""
    _need_gpu(q, hidden, x_rows, w_k, w_v)
    if any(t.dtype != torch.float16 for t in (q, hidden, w_k, w_v)) or x_rows.dtype != torch.int32:
        raise TypeError("fresco_amd.attention_kvproj: fp16 tensors and an int32 row table required")
    B, Lq, C = q.shape
    D = C // heads
    K_in = hidden.shape[-1]
    if w_k.shape != (C, K_in) or w_v.shape != (C, K_in) or not (w_k.is_contiguous() and w_v.is_contiguous()):
        raise ValueError("attention_kvproj: contiguous (%d, %d) weights expected" % (C, K_in))
    if x_rows.numel() != n_groups * M or not x_rows.is_contiguous():
        raise ValueError("attention_kvproj: x_rows must hold n_groups * M = %d entries" % (n_groups * M))
    q, q_ld, _ = _rows(q)
    hidden, x_ld, _ = _rows(hidden)
    lib = _lib.load()
    ws_bytes = lib.fresco_attn_workspace_bytes(n_groups, heads, M, D)
    ws = (workspace or _default_ws).get(ws_bytes, q.device)
    out = torch.empty((B, Lq, C), dtype=q.dtype, device=q.device)
    rc = lib.fresco_attn_fwd_kvproj(q.data_ptr(), hidden.data_ptr(), x_ld, x_rows.data_ptr(), w_k.data_ptr(), w_v.data_ptr(),
                                    out.data_ptr(), ws.data_ptr(), ws.numel(), B, heads, Lq, D, n_groups, M, K_in,
                                    float(scale), q_ld, _stream())
    _lib.check(rc, "fresco_attn_fwd_kvproj(B=%d,H=%d,Lq=%d,D=%d,groups=%d,M=%d,K=%d)" % (B, heads, Lq, D, n_groups, M, K_in))
    return out


def attention_f32(q, k, v, scale):
    """softmax(scale * q k^T) v at fp32 accuracy (fresco_attn_f32_guarded): q (B,Lq,D), k (B,Lk,D), v (B,Lk,Dv) ->
    (B,Lq,Dv).  One head; batch entries are independent problems (windows).  When several 128-query workgroups share a
    key set (Lq >= 256) K and V are converted to the kernel's operand images once per launch, into a workspace taken from
    the caching allocator PER CALL (stream-safe: two streams running the flow network never share it; nothing is kept).
    No range limit: the split-fp16 kernels need |q scale log2 e|, |k|, |v| < 1000; a range pass on the device raises a
    flag otherwise and the exact-fp32 MFMA kernel recomputes the launch (no host sync either way)."""
    _need_gpu(q, k, v)
    q, k, v = _f32c(q), _f32c(k), _f32c(v)
    B, Lq, D = q.shape
    Lk, Dv = k.shape[1], v.shape[2]
    if k.shape != (B, Lk, D) or v.shape[:2] != (B, Lk):
        raise ValueError("attention_f32: q (B,Lq,D), k (B,Lk,D), v (B,Lk,Dv) expected")
    out = torch.empty(B, Lq, Dv, dtype=torch.float32, device=q.device)
    lib = _lib.load()
    need = lib.fresco_attn_f32_workspace_bytes(B, Lk, D, Dv) if Lq >= 256 else 0
    if need:
        # workspace form: the range tests ride in the split pass and the attention prologue; the flag is a word of a
        # zero-filled pool, handed out once (no memset, no range pass: 20 us of short launches per call, round 6)
        ws = torch.empty(need, dtype=torch.uint8, device=q.device)
        rc = lib.fresco_attn_f32_guarded_ws(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), ws.data_ptr(), need,
                                            _a32_zero_flag(q.device), B, Lq, Lk, D, Dv, float(scale), _stream())
        _lib.check(rc, "fresco_attn_f32_guarded_ws(B=%d,Lq=%d,Lk=%d,D=%d,Dv=%d)" % (B, Lq, Lk, D, Dv))
        return out
    flag = torch.empty(1, dtype=torch.int32, device=q.device)
    rc = lib.fresco_attn_f32_guarded(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), None, 0,
                                     flag.data_ptr(), B, Lq, Lk, D, Dv, float(scale), _stream())
    _lib.check(rc, "fresco_attn_f32_guarded(B=%d,Lq=%d,Lk=%d,D=%d,Dv=%d)" % (B, Lq, Lk, D, Dv))
    return out


_A32_POOL_WORDS = 4096
_a32_pools = None  # a threading.local with a dict keyed by (device index, stream), made on first use


def _a32_zero_flag(device):
    """address of an int32 that is zero in the current stream's order and has never been handed out: word i of a pool that
    torch.zeros created ON THIS STREAM (one fill per 4096 calls).  A pool that runs out is dropped: its memory goes back to
    the caching allocator, which re-issues it in this stream's order (the kernels that still read its words come first).
    Under stream capture a word is handed out once per CAPTURE and read by every replay: a replay whose operands left the
    range leaves it set, and later replays of that graph take the exact-fp32 kernel as well (correct, slower) -- the flow
    network is not captured anywhere in this package."""
    global _a32_pools
    if _a32_pools is None:
        _a32_pools = _threading.local()
    pools = getattr(_a32_pools, "d", None)
    if pools is None:
        pools = _a32_pools.d = {}
    key = (device.index, _stream())
    ent = pools.get(key)
    if ent is None or ent[1] >= _A32_POOL_WORDS:
        ent = pools[key] = [torch.zeros(_A32_POOL_WORDS, dtype=torch.int32, device=device), 0]
    ptr = ent[0].data_ptr() + 4 * ent[1]
    ent[1] += 1
    return ptr


# ---- the flow network's dense layers (csrc/flownet.hip): fp32 tensors in NHWC / token layout, products on the fp16 matrix
# pipe from (hi, lo) fp16 planes.  A "split" below is a pair of fp16 tensors (M, ld) with x * scale = hi + lo (scale: FN_A_SCALE
# for activations, FN_W_SCALE for weights -- powers of two that keep the lo pieces normal fp16 numbers, csrc/flownet.hip).
FN_A_SCALE, FN_W_SCALE = 64.0, 1024.0
_fn_zero_page = {}  # per device: 128 bytes of zeros, the LDS-DMA source of rows outside a GEMM (padding, tails)

# Range guard of the split planes: |x * scale| beyond 65000 saturates -- finite, wrong.  Every producer ORs 1
# into the int32 word of the innermost `fn_range_guard` of the calling thread; whoever opened the guard reads the word once
# (one host sync per flow-network forward) and recomputes with library ops when it is set.
import threading as _threading

_fn_guard = _threading.local()


class fn_range_guard:
    """with fn_range_guard(device) as g: ... producers ...;  g.tripped() -> bool (synchronises)"""

    def __init__(self, device):
        self.flag = torch.zeros(1, dtype=torch.int32, device=device)

    def __enter__(self):
        self._outer = getattr(_fn_guard, "cur", None)
        _fn_guard.cur = self
        return self

    def __exit__(self, *exc):
        _fn_guard.cur = self._outer
        return False

    def tripped(self):
        return bool(int(self.flag.item()) != 0)


def _fn_flag_ptr(device):
    g = getattr(_fn_guard, "cur", None)
    if g is None:
        return None
    if g.flag.device != device:
        raise ValueError("fn_range_guard on %s, operands on %s" % (g.flag.device, device))
    return g.flag.data_ptr()


def _fn_f32_operand(t, name, like):
    """optional fp32 side operand (bias, gamma, mean, residual ...) of a flownet.hip kernel: the kernels read raw fp32 words,
    so a .half() module or a strided view must not get through as it is"""
    if t is None:
        return None
    if not t.is_cuda or t.device != like.device:
        raise ValueError("fresco_amd: %s on %s, operands on %s" % (name, t.device, like.device))
    return _f32c(t)


def _fn_rows_table(t, name, like, n=None):
    if t is None:
        return None
    if t.dtype != torch.int32 or not t.is_contiguous() or t.device != like.device or (n is not None and t.numel() != n):
        raise ValueError("fresco_amd: %s must be a contiguous int32 table%s on the operands' device"
                         % (name, "" if n is None else " of %d rows" % n))
    return t


def fn_prep(x, mean=None, rstd=None, residual=None, rows_per_img=0, relu_a=False, relu_b=False, want_f32=False,
            want_split=True, ld=None, out_split=None, scale=FN_A_SCALE):
    """y = relu_b?(relu_a?((x - mean) * rstd) + residual) on (M, C) fp32 rows (fresco_fn_prep).  Returns (y or None,
    (hi, lo) or None); the planes have row stride ld >= C with channels C .. ld-1 zeroed.  out_split: planes to write into
    (views with a row stride are fine: (hi, lo, ld))."""
    _need_gpu(x)
    x = _f32c(x)
    M, C = x.shape
    mean, rstd, residual = (_fn_f32_operand(t, n_, x) for t, n_ in ((mean, "mean"), (rstd, "rstd"), (residual, "residual")))
    y = torch.empty_like(x) if want_f32 else None
    hi = lo = None
    ldo = C
    if out_split is not None:
        hi, lo, ldo = out_split
    elif want_split:
        ldo = ld or C
        hi = torch.empty(M, ldo, dtype=torch.float16, device=x.device)
        lo = torch.empty(M, ldo, dtype=torch.float16, device=x.device)
    rc = _lib.load().fresco_fn_prep(x.data_ptr(), _ptr(mean), _ptr(rstd), _ptr(residual), _ptr(y), _ptr(hi), _ptr(lo), M, C,
                                    int(ldo), int(rows_per_img), int(relu_a), int(relu_b), float(scale),
                                    _fn_flag_ptr(x.device), _stream())
    _lib.check(rc, "fresco_fn_prep(M=%d,C=%d,ld=%d)" % (M, C, ldo))
    return y, ((hi, lo) if hi is not None else None)


def fn_colstats(x, n_img, eps=1e-5):
    """InstanceNorm2d statistics of x (n_img * rows, C) fp32 -> (mean, rstd), (n_img, C) each (fresco_fn_colstats)"""
    _need_gpu(x)
    M, C = x.shape
    rows = M // n_img
    lib = _lib.load()
    nbytes = lib.fresco_fn_colstats_workspace_bytes(n_img, rows, C)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    mean = torch.empty(n_img, C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(n_img, C, dtype=torch.float32, device=x.device)
    rc = lib.fresco_fn_colstats(x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), nbytes, n_img, rows, C,
                                float(eps), _stream())
    _lib.check(rc, "fresco_fn_colstats(n=%d,rows=%d,C=%d)" % (n_img, rows, C))
    return mean, rstd


def fn_gemm(a, w, N, K, bias=None, act=0, conv=None, M=None, want_f32=True, want_split=False, out_split=None,
            instance_norm_eps=None, a_rows=None, out_rows=None, out_f32=None, out_blocks=0):
    """act(A W^T + bias) (fresco_fn_gemm).  a = (hi, lo) planes, (rows, lda); w = (hi, lo) planes (N, K).
    conv = (n_img, H, W, kh, kw, stride, pad): implicit im2col of the NHWC tensor behind `a` (K = kh kw cin).
    a_rows / out_rows (linear layers): int32 (M) tables -- problem row m reads input row a_rows[m], writes output row
    out_rows[m] (the rows of `out` not named by the table keep whatever they held: pass a full permutation).
    instance_norm_eps (convolutions): also return InstanceNorm2d statistics (mean, rstd) of the result, from partial sums
    the kernel's epilogue leaves behind (no second pass over the output).
    Returns (out fp32 (M, N) or None, (hi, lo) (M, N) or None[, (mean, rstd)])."""
    ah, al = a
    wh, wl = w
    _need_gpu(ah, wh)
    lda = ah.stride(0)
    if conv is None:
        M = (ah.shape[0] if a_rows is None else a_rows.numel()) if M is None else M
        cargs = (0, 0, 0, 0, 0, 1, 0)
    else:
        n_img, H, W, kh, kw, stride, pad = conv
        OH, OW = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
        M = n_img * OH * OW
        cargs = (n_img, H, W, kh, kw, stride, pad)
    dev = ah.device
    bias = _fn_f32_operand(bias, "bias", ah)
    a_rows = _fn_rows_table(a_rows, "a_rows", ah, M)
    out_rows = _fn_rows_table(out_rows, "out_rows", ah, M)
    # out_blocks = n > 1: W holds n stacked projections; the result is (n, M, N / n) fp32 -- every projection's rows contiguous
    ocb, obs, ldc = 0, 0, N
    if out_blocks and out_blocks > 1:
        if N % out_blocks or out_f32 is not None or want_split or out_split is not None or conv is not None:
            raise ValueError("fn_gemm: out_blocks needs a plain fp32 linear product whose N divides into the blocks")
        ocb = N // out_blocks
        out = torch.empty(out_blocks, M, ocb, dtype=torch.float32, device=dev)
        obs, ldc = M * ocb, ocb
    else:
        out = out_f32 if out_f32 is not None else (torch.empty(M, N, dtype=torch.float32, device=dev) if want_f32 else None)
    oh = ol = None
    ldo = N
    if out_split is not None:
        oh, ol, ldo = out_split
    elif want_split:
        oh = torch.empty(M, N, dtype=torch.float16, device=dev)
        ol = torch.empty(M, N, dtype=torch.float16, device=dev)
    stats = None
    fused = instance_norm_eps is not None and conv is not None and (M // conv[0]) % 256 == 0
    if fused:
        stats = torch.empty((M // 64) * N * 2, dtype=torch.float64, device=dev)
    zeros = _fn_zero_page.get(dev)
    if zeros is None:
        zeros = _fn_zero_page[dev] = torch.zeros(64, dtype=torch.float16, device=dev)
    lib = _lib.load()
    rc = lib.fresco_fn_gemm(ah.data_ptr(), al.data_ptr(), lda, wh.data_ptr(), wl.data_ptr(), _ptr(bias), _ptr(out),
                            _ptr(oh), _ptr(ol), int(ldc), int(ldo), M, N, K, int(act), 1.0 / (FN_A_SCALE * FN_W_SCALE),
                            FN_A_SCALE, *cargs, _ptr(stats), zeros.data_ptr(), _ptr(a_rows), _ptr(out_rows),
                            _fn_flag_ptr(dev), int(ocb), int(obs), _stream())
    _lib.check(rc, "fresco_fn_gemm(M=%d,N=%d,K=%d,conv=%s)" % (M, N, K, conv))
    res = (out, ((oh, ol) if oh is not None else None))
    if instance_norm_eps is None:
        return res
    if not fused:
        return res + (fn_colstats(out, conv[0], instance_norm_eps),)
    mean = torch.empty(conv[0], N, dtype=torch.float32, device=dev)
    rstd = torch.empty(conv[0], N, dtype=torch.float32, device=dev)
    rc = lib.fresco_fn_colstats_finish(stats.data_ptr(), mean.data_ptr(), rstd.data_ptr(), conv[0], M // conv[0], N,
                                       float(instance_norm_eps), _stream())
    _lib.check(rc, "fresco_fn_colstats_finish")
    return res + ((mean, rstd),)


def fn_layernorm(x, gamma, beta, residual=None, eps=1e-5, want_f32=True, want_split=False, out_split=None):
    """residual + LayerNorm(x) over the 128 channels of (M, 128) fp32 rows (fresco_fn_layernorm) -> (y or None, split or None)"""
    _need_gpu(x)
    x = _f32c(x)
    M, C = x.shape
    gamma, beta, residual = (_fn_f32_operand(t, n_, x) for t, n_ in ((gamma, "gamma"), (beta, "beta"), (residual, "residual")))
    y = torch.empty_like(x) if want_f32 else None
    oh = ol = None
    ldo = C
    if out_split is not None:
        oh, ol, ldo = out_split
    elif want_split:
        oh = torch.empty(M, C, dtype=torch.float16, device=x.device)
        ol = torch.empty(M, C, dtype=torch.float16, device=x.device)
    rc = _lib.load().fresco_fn_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(residual), _ptr(y), _ptr(oh),
                                         _ptr(ol), C, int(ldo), M, C, float(eps), FN_A_SCALE, _fn_flag_ptr(x.device),
                                         _stream())
    _lib.check(rc, "fresco_fn_layernorm(M=%d,C=%d)" % (M, C))
    return y, ((oh, ol) if oh is not None else None)


def fn_conv7_weight(weight):
    """(hi, lo) planes of the stem's weight as fresco_fn_conv7_rgb reads it: (64, 3, 7, 7) -> W'[cout][32 ky + 3 kx + ci],
    the 11 surplus positions of every kernel row zero"""
    w = weight.detach().float().permute(0, 2, 3, 1).reshape(64, 7, 21)
    w = torch.nn.functional.pad(w, (0, 11)).reshape(64, 224).contiguous()
    _, planes = fn_prep(w, scale=FN_W_SCALE)
    return planes


def fn_conv7_rgb(x_nhwc, w_split, instance_norm_eps=None):
    """Conv2d(3, 64, 7, stride 2, padding 3, bias=False) on (n, H, W, 3) fp32 NHWC, w_split = fn_conv7_weight(weight)
    -> (n, OH, OW, 64); instance_norm_eps: also the (mean, rstd) of the InstanceNorm2d that follows (partial sums out of
    the epilogue where the map allows it, a second pass otherwise)"""
    _need_gpu(x_nhwc, *w_split)
    x_nhwc = _f32c(x_nhwc)
    wh, wl = w_split
    if tuple(wh.shape) != (64, 224) or tuple(wl.shape) != (64, 224) or wh.dtype != torch.float16 or wl.dtype != torch.float16 \
            or not wh.is_contiguous() or not wl.is_contiguous():
        raise ValueError("fn_conv7_rgb: w_split must be the (64, 224) fp16 planes of fn_conv7_weight")
    n, H, W, c = x_nhwc.shape
    if c != 3:
        raise ValueError("fn_conv7_rgb: x must be (n, H, W, 3)")
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dev = x_nhwc.device
    out = torch.empty(n, OH, OW, 64, dtype=torch.float32, device=dev)
    fused = instance_norm_eps is not None and OW % 64 == 0 and (OH * OW) % 256 == 0
    stats = torch.empty((n * OH * OW // 64) * 64 * 2, dtype=torch.float64, device=dev) if fused else None
    lib = _lib.load()
    rc = lib.fresco_fn_conv7_rgb(x_nhwc.data_ptr(), wh.data_ptr(), wl.data_ptr(), out.data_ptr(), _ptr(stats), n, H, W,
                                 _fn_flag_ptr(dev), _stream())
    _lib.check(rc, "fresco_fn_conv7_rgb(n=%d,H=%d,W=%d)" % (n, H, W))
    if instance_norm_eps is None:
        return out
    if not fused:
        return out, fn_colstats(out.view(n * OH * OW, 64), n, instance_norm_eps)
    mean = torch.empty(n, 64, dtype=torch.float32, device=dev)
    rstd = torch.empty(n, 64, dtype=torch.float32, device=dev)
    rc = lib.fresco_fn_colstats_finish(stats.data_ptr(), mean.data_ptr(), rstd.data_ptr(), n, OH * OW, 64,
                                       float(instance_norm_eps), _stream())
    _lib.check(rc, "fresco_fn_colstats_finish")
    return out, (mean, rstd)


def fn_convex_upsample(logits, flow_tok, B, h, w):
    """convex upsampling by 8 (gmflow.py:75-90): logits (B h w, 576) fp32 rows of the mask head, flow_tok (B, h w, 2) ->
    (B, 2, 8 h, 8 w) (fresco_fn_convex_upsample)"""
    _need_gpu(logits, flow_tok)
    logits, flow_tok = _f32c(logits), _f32c(flow_tok)
    out = torch.empty(B, 2, 8 * h, 8 * w, dtype=torch.float32, device=logits.device)
    rc = _lib.load().fresco_fn_convex_upsample(logits.data_ptr(), flow_tok.data_ptr(), out.data_ptr(), B, h, w, _stream())
    _lib.check(rc, "fresco_fn_convex_upsample(B=%d,h=%d,w=%d)" % (B, h, w))
    return out


_checked_tables = {}


def _check_once(key, tensor, fn):
    """Run the (synchronising) validity check `fn` once per index table; the verdict is cached with the table
    (address, size, version; a weak reference guards against a recycled address)."""
    import weakref
    hit = _checked_tables.get(key)
    if hit is not None and hit() is tensor:
        return
    if len(_checked_tables) > 64:
        _checked_tables.clear()
    fn()
    _checked_tables[key] = weakref.ref(tensor)


def _check_permutations(owner, fwd_map, HW):
    """every row of fwd_map (N, HW) must be a permutation of 0..HW-1: the temporal kernel writes
    out[f][fwd_map[f][p]], so anything else leaves rows of the result unwritten (the reference's
    get_mapping_ind always produces permutations, src/flow_utils.py:99-101, 134-135).  `owner` is the tensor
    object the caller holds (the verdict is cached with IT: views made here are new objects every call)."""
    def run():
        srt = torch.sort(fwd_map, dim=1).values
        ok = bool((srt == torch.arange(HW, device=fwd_map.device, dtype=fwd_map.dtype)).all())
        if not ok:
            raise ValueError("fresco_amd.temporal_attention: fwd_mapping rows must be permutations of 0..%d" % (HW - 1))
    _check_once(("perm", owner.data_ptr(), tuple(owner.shape), owner._version), owner, run)


def _prep_maps(fwd_map, mask, N, HW):
    if fwd_map is not None:
        fwd_map = fwd_map.reshape(N, HW)
        if fwd_map.dtype != torch.int64:
            fwd_map = fwd_map.to(torch.int64)
        fwd_map = fwd_map.contiguous()
    mask = mask.reshape(-1, N, N)
    if mask.dtype == torch.bool:
        mask = mask.contiguous().view(torch.uint8)
    elif mask.dtype != torch.uint8:
        mask = (mask != 0).contiguous().view(torch.uint8)
    return fwd_map, mask.contiguous()


def temporal_attention(q, k, v, fwd_map, mask, heads, scale, chunk):
    """fresco_temporal_attn: q, k, v (chunk*N, HW, C) fp16; fwd_map (N,HW) int64 (a permutation per frame,
    checked once per table); mask (HW,N,N) bool."""
    _need_gpu(q, k, v, fwd_map, mask)
    if q.dtype != torch.float16 or k.dtype != torch.float16 or v.dtype != torch.float16:
        raise TypeError("fresco_amd.temporal_attention: fp16 tensors required")
    Bt, HW, C = q.shape
    N = Bt // chunk
    D = C // heads
    q, q_ld, _ = _rows(q)
    k, k_ld, _ = _rows(k)
    v, v_ld, _ = _rows(v)
    owner = fwd_map
    fwd_map, mask = _prep_maps(fwd_map, mask, N, HW)
    _check_permutations(owner, fwd_map, HW)
    out = torch.empty((Bt, HW, C), dtype=q.dtype, device=q.device)
    rc = _lib.load().fresco_temporal_attn_ld(q.data_ptr(), k.data_ptr(), v.data_ptr(), fwd_map.data_ptr(),
                                             mask.data_ptr(), out.data_ptr(), chunk, N, HW, heads, D,
                                             float(scale), q_ld, k_ld, v_ld, _stream())
    _lib.check(rc, "fresco_temporal_attn(chunk=%d,N=%d,HW=%d,H=%d,D=%d)" % (chunk, N, HW, heads, D))
    return out


def temporal_pack(q, k, v, fwd_map, chunk, n_loc, f0, world):
    """Multi-GPU, way out (fresco_temporal_pack): this rank's frames [f0, f0+n_loc) of q, k, v (chunk*n_loc, HW, C)
    gathered along the trajectories into per-destination ranges: returns (world, n_loc, chunk, HW/world, 3C)."""
    _need_gpu(q, k, v, fwd_map)
    Bl, HW, C = q.shape
    q, q_ld, _ = _rows(q)
    k, k_ld, _ = _rows(k)
    v, v_ld, _ = _rows(v)
    buf = torch.empty((world, n_loc, chunk, HW // world, 3 * C), dtype=torch.float16, device=q.device)
    rc = _lib.load().fresco_temporal_pack(q.data_ptr(), k.data_ptr(), v.data_ptr(), fwd_map.data_ptr(), buf.data_ptr(),
                                          chunk, n_loc, f0, HW, C, world, q_ld, k_ld, v_ld, _stream())
    _lib.check(rc, "fresco_temporal_pack(chunk=%d,n_loc=%d,HW=%d,C=%d,world=%d)" % (chunk, n_loc, HW, C, world))
    return buf


def temporal_attention_packed(qkv, mask, heads, scale, chunk):
    """fresco_temporal_attn_packed: qkv (N, chunk, P, 3C) fp16 = q | k | v rows already in trajectory order for a
    range of P trajectories, mask (P, N, N) for that range -> (N, chunk, P, C)."""
    _need_gpu(qkv, mask)
    N, ch, P, C3 = qkv.shape
    C = C3 // 3
    D = C // heads
    if ch != chunk or qkv.dtype != torch.float16 or not qkv.is_contiguous():
        raise ValueError("temporal_attention_packed: contiguous fp16 (N, chunk, P, 3C) expected")
    _, mask = _prep_maps(None, mask, N, 1)
    if mask.shape[0] != P:
        raise ValueError("temporal_attention_packed: mask must cover the %d trajectories of the range" % P)
    out = torch.empty((N, chunk, P, C), dtype=torch.float16, device=qkv.device)
    rc = _lib.load().fresco_temporal_attn_packed(qkv.data_ptr(), mask.data_ptr(), out.data_ptr(), chunk, N, P, heads,
                                                 D, float(scale), _stream())
    _lib.check(rc, "fresco_temporal_attn_packed(chunk=%d,N=%d,P=%d,H=%d,D=%d)" % (chunk, N, P, heads, D))
    return out


def temporal_unpack(buf, fwd_map, chunk, n_loc, f0, world):
    """Multi-GPU, way back (fresco_temporal_unpack): buf (world, n_loc, chunk, HW/world, C) = this rank's frames'
    result rows per trajectory range -> (chunk*n_loc, HW, C) in frame order."""
    _need_gpu(buf, fwd_map)
    w, nl, ch, Pw, C = buf.shape
    HW = Pw * world
    out = torch.empty((chunk * n_loc, HW, C), dtype=torch.float16, device=buf.device)
    rc = _lib.load().fresco_temporal_unpack(buf.contiguous().data_ptr(), fwd_map.data_ptr(), out.data_ptr(), chunk,
                                            n_loc, f0, HW, C, world, _stream())
    _lib.check(rc, "fresco_temporal_unpack(chunk=%d,n_loc=%d,HW=%d,C=%d,world=%d)" % (chunk, n_loc, HW, C, world))
    return out


# ---------------------------------------------------------------------------------------------
# warp family (fp32)
# ---------------------------------------------------------------------------------------------
def flow_warp(x, flow):
    """(B,C,h,w) fp32 sampled at pixel + flow[b % Bf] (bilinear, zeros, align_corners=True)."""
    _need_gpu(x, flow)
    x, flow = _f32c(x), _f32c(flow)
    B, C, h, w = x.shape
    Bf = flow.shape[0]
    if flow.shape[1:] != (2, h, w) or B % Bf != 0:
        raise ValueError("flow %s does not match feature %s" % (tuple(flow.shape), tuple(x.shape)))
    out = torch.empty_like(x)
    rc = _lib.load().fresco_flow_warp(x.data_ptr(), flow.data_ptr(), out.data_ptr(), B, C, h, w, Bf, _stream())
    _lib.check(rc, "fresco_flow_warp")
    return out


def resize_bilinear(x, scale_factor, mul=1.0):
    """F.interpolate(x * mul, scale_factor=s, mode='bilinear') for (B,C,H,W)."""
    _need_gpu(x)
    x = _f32c(x)
    B, C, H, W = x.shape
    ho, wo = int(math.floor(H * scale_factor)), int(math.floor(W * scale_factor))
    if ho <= 0 or wo <= 0:
        raise ValueError("resize to empty output")
    out = torch.empty(B, C, ho, wo, dtype=torch.float32, device=x.device)
    r = 1.0 / scale_factor
    rc = _lib.load().fresco_resize_bilinear(x.data_ptr(), out.data_ptr(), B * C, H, W, ho, wo, r, r,
                                            float(mul), _stream())
    _lib.check(rc, "fresco_resize_bilinear")
    return out


def max_pool(x, k):
    """F.max_pool2d(x, kernel_size=k) for (B,C,H,W)."""
    _need_gpu(x)
    x = _f32c(x)
    B, C, H, W = x.shape
    out = torch.empty(B, C, H // k, W // k, dtype=torch.float32, device=x.device)
    rc = _lib.load().fresco_max_pool(x.data_ptr(), out.data_ptr(), B * C, H, W, k, _stream())
    _lib.check(rc, "fresco_max_pool")
    return out


def dilate(x, k):
    _need_gpu(x)
    x = _f32c(x)
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    rc = _lib.load().fresco_dilate(x.data_ptr(), out.data_ptr(), B * C, H, W, k, _stream())
    _lib.check(rc, "fresco_dilate")
    return out


def flow_occlusion(fwd_flow, bwd_flow, images=None, alpha=0.01, beta=0.5, color_thr=255 * 0.25):
    """(fwd_occ, bwd_occ) fp32 (N,H,W) for the frame pairs (n, n+1 mod N); images (N,C,H,W) in 0..255 adds
    the colour-difference test, None = forward_backward_consistency_check alone."""
    _need_gpu(fwd_flow)
    fwd_flow, bwd_flow = _f32c(fwd_flow), _f32c(bwd_flow)
    N, two, H, W = fwd_flow.shape
    if two != 2 or bwd_flow.shape != fwd_flow.shape:
        raise ValueError("flows must be (N,2,H,W) and of equal shape")
    C, ip = 0, None
    if images is not None:
        images = _f32c(images)
        if images.shape[0] != N or tuple(images.shape[2:]) != (H, W):
            raise ValueError("images must be (N,C,H,W) matching the flows")
        C, ip = images.shape[1], images.data_ptr()
    fo = torch.empty(N, H, W, dtype=torch.float32, device=fwd_flow.device)
    bo = torch.empty_like(fo)
    rc = _lib.load().fresco_flow_occlusion(ip, fwd_flow.data_ptr(), bwd_flow.data_ptr(), fo.data_ptr(), bo.data_ptr(),
                                           N, C, H, W, alpha, beta, color_thr, _stream())
    _lib.check(rc, "fresco_flow_occlusion")
    return fo, bo


def warp_fuse_chain(lat, bwd_flow, fwd_flow, bwd_occ, fwd_occ, sal, warp_sal, warp_sal_last, chunk):
    """In-place frame chain of warp_tensor on lat (chunk*N, C, h, w) fp32 contiguous."""
    _need_gpu(lat)
    assert lat.dtype == torch.float32 and lat.is_contiguous()
    Bt, C, h, w = lat.shape
    N = Bt // chunk
    args = [_f32c(t) for t in (bwd_flow, fwd_flow, bwd_occ, fwd_occ, sal, warp_sal, warp_sal_last)]
    rc = _lib.load().fresco_warp_fuse_chain(lat.data_ptr(), *[t.data_ptr() for t in args], chunk, N, C, h, w,
                                            _stream())
    _lib.check(rc, "fresco_warp_fuse_chain(N=%d)" % N)
    return lat


def adain(content, style, eps_content=1e-5, eps_style=1.0):
    _need_gpu(content, style)
    if content.shape != style.shape:
        raise ValueError("AdaIN: content %s vs style %s" % (tuple(content.shape), tuple(style.shape)))
    dt = torch.result_type(content, style)
    if dt not in (torch.float16, torch.float32):
        dt = torch.float32
    c, s = content.to(dt).contiguous(), style.to(dt).contiguous()
    rows = c.shape[0] * c.shape[1]
    L = c.numel() // rows
    out = torch.empty_like(c)
    rc = _lib.load().fresco_adain(c.data_ptr(), s.data_ptr(), out.data_ptr(), rows, L, float(eps_content),
                                  float(eps_style), _lib.F16 if dt == torch.float16 else _lib.F32, _stream())
    _lib.check(rc, "fresco_adain")
    return out


def chan_mean_std(feat, eps=1e-5):
    """-> (mean, std) fp32 of shape (N*C,): per-plane mean and sqrt(unbiased variance + eps)  (src/utils.py:58-67)."""
    _need_gpu(feat)
    dt = feat.dtype if feat.dtype in (torch.float16, torch.float32) else torch.float32
    x = feat.to(dt).contiguous()
    rows = x.shape[0] * x.shape[1]
    L = x.numel() // rows
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    std = torch.empty(rows, dtype=torch.float32, device=x.device)
    rc = _lib.load().fresco_chan_mean_std(x.data_ptr(), mean.data_ptr(), std.data_ptr(), rows, L, float(eps),
                                          _lib.F16 if dt == torch.float16 else _lib.F32, _stream())
    _lib.check(rc, "fresco_chan_mean_std")
    return mean, std


# ---------------------------------------------------------------------------------------------
# feature optimisation (fp32)
# ---------------------------------------------------------------------------------------------
def _opt_args(cs, prep, target, chunk):
    Bt, C, h, w = cs.shape
    N = Bt // chunk
    if prep is not None:
        fwd_flow, bwd_flow, fwd_occ, bwd_occ = [_f32c(t) for t in prep]
        assert fwd_flow.shape == (N, 2, h, w) and bwd_flow.shape == (N, 2, h, w), (fwd_flow.shape, (N, 2, h, w))
        assert fwd_occ.numel() == N * h * w and bwd_occ.numel() == N * h * w
        keep = (fwd_flow, bwd_flow, fwd_occ, bwd_occ)
    else:
        keep = (None, None, None, None)
    if target is not None:
        target = _f32c(target)
        assert target.shape == (Bt, h * w, h * w), (target.shape, (Bt, h * w, h * w))
    return N, C, h, w, keep, target


def opt_run(cs, prep, target, intra_weight, iters, chunk, lr=0.2, betas=(0.9, 0.999), eps=1e-8,
            workspace=None, context=None):
    """`iters` Adam steps on cs (chunk*N, C, h, w) fp32 contiguous, in place (fresco_opt_run).
    prep = (fwd_flow, bwd_flow, fwd_occ, bwd_occ) at feature resolution, or None."""
    _need_gpu(cs)
    assert cs.dtype == torch.float32 and cs.is_contiguous()
    N, C, h, w, keep, target = _opt_args(cs, prep, target, chunk)
    lib = _lib.load()
    has_t, has_s = int(prep is not None), int(target is not None and intra_weight > 0)
    nbytes = lib.fresco_opt_workspace_bytes(chunk, N, C, h, w, has_t, has_s)
    ws = (workspace or _default_ws).get(nbytes, cs.device)
    ctx = context or _opt_contexts.get(cs.device)
    rc = lib.fresco_opt_run_ctx(ctx.ptr, cs.data_ptr(), _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), _ptr(keep[3]),
                            _ptr(target), ws.data_ptr(), ws.numel(), chunk, N, C, h, w, float(intra_weight),
                            int(iters), float(lr), float(betas[0]), float(betas[1]), float(eps), _stream())
    _lib.check(rc, "fresco_opt_run(chunk=%d,N=%d,C=%d,h=%d,w=%d)" % (chunk, N, C, h, w))
    return cs


def opt_run_sharded(cs, prep_pairs, target, intra_weight, iters, chunk, N_total, exchange, lr=0.2,
                    betas=(0.9, 0.999), eps=1e-8, workspace=None):
    """Frame-sharded optimize_feature loop (fresco_opt_sharded_begin / _step_part).

    cs: local (chunk*n_loc, C, h, w) fp32 contiguous, updated in place.
    prep_pairs = (fwd_flow, bwd_flow, fwd_occ, bwd_occ) for the n_loc+1 local frame pairs, or None.
    exchange: where the inter-GPU traffic happens.  Either an object with `halo_start(cs) -> handle` and
    `halo_finish(handle) -> (halo_l, halo_r)` (fresco_amd.dist.FrameShard): the exchange of the frames Adam(it-1) left
    behind is started, the launches of step `it` that read no halo frame run (part 1), then the halos are waited for and
    the boundary pairs' signs + Adam follow (part 2).  Or a plain callable `exchange(cs) -> (halo_l, halo_r)` (blocking,
    called before every undivided step).  halo_l / halo_r: the current frame before / after the owned range,
    (chunk, C, h, w) each.  Both forms give identical bits."""
    _need_gpu(cs)
    assert cs.dtype == torch.float32 and cs.is_contiguous()
    Bt, C, h, w = cs.shape
    n_loc = Bt // chunk
    keep = (None, None, None, None)
    if prep_pairs is not None:
        keep = tuple(_f32c(t) for t in prep_pairs)
        assert keep[0].shape == (n_loc + 1, 2, h, w) and keep[2].numel() == (n_loc + 1) * h * w
    if target is not None:
        target = _f32c(target)
        assert target.shape == (Bt, h * w, h * w)
    lib = _lib.load()
    has_t, has_s = int(prep_pairs is not None), int(target is not None and intra_weight > 0)
    nbytes = lib.fresco_opt_sharded_workspace_bytes(chunk, n_loc, C, h, w, has_t, has_s)
    ws = (workspace or _default_ws).get(nbytes, cs.device)
    rc = lib.fresco_opt_sharded_begin(_ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), _ptr(keep[3]), ws.data_ptr(),
                                      ws.numel(), chunk, n_loc, N_total, C, h, w, has_s, _stream())
    _lib.check(rc, "fresco_opt_sharded_begin")
    overlapped = hasattr(exchange, "halo_start") and hasattr(exchange, "halo_finish")

    def step(it, halo_l, halo_r, part):
        rc = lib.fresco_opt_sharded_step_part(cs.data_ptr(), _ptr(halo_l), _ptr(halo_r), _ptr(keep[0]), _ptr(keep[1]),
                                              _ptr(keep[2]), _ptr(keep[3]), _ptr(target), ws.data_ptr(), ws.numel(),
                                              chunk, n_loc, N_total, C, h, w, float(intra_weight), it, float(lr),
                                              float(betas[0]), float(betas[1]), float(eps), part, _stream())
        _lib.check(rc, "fresco_opt_sharded_step_part(it=%d, part=%d)" % (it, part))

    for it in range(1, iters + 1):
        if not has_t:
            step(it, None, None, 3)
        elif overlapped:
            handle = exchange.halo_start(cs)           # asynchronous: the frames Adam(it-1) produced
            step(it, None, None, 1)                    # normalise, interior signs, Gram, S V: under the transfer
            halo_l, halo_r = exchange.halo_finish(handle)
            halo_l, halo_r = _f32c(halo_l), _f32c(halo_r)
            step(it, halo_l, halo_r, 2)                # boundary signs + Adam
        else:
            halo_l, halo_r = exchange(cs)
            halo_l, halo_r = _f32c(halo_l), _f32c(halo_r)
            step(it, halo_l, halo_r, 3)
    return cs


def opt_loss_grad(cs, prep, target, intra_weight, chunk, workspace=None):
    """One closure evaluation: returns (loss_temporal, loss_spatial) device tensor (2,) and grad."""
    _need_gpu(cs)
    cs = _f32c(cs)
    N, C, h, w, keep, target = _opt_args(cs, prep, target, chunk)
    lib = _lib.load()
    has_t, has_s = int(prep is not None), int(target is not None and intra_weight > 0)
    nbytes = lib.fresco_opt_workspace_bytes(chunk, N, C, h, w, has_t, has_s)
    ws = (workspace or _default_ws).get(nbytes, cs.device)
    grad = torch.empty_like(cs)
    loss = torch.zeros(2, dtype=torch.float32, device=cs.device)
    rc = lib.fresco_opt_loss_grad(cs.data_ptr(), _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), _ptr(keep[3]),
                                  _ptr(target), grad.data_ptr(), loss.data_ptr(), ws.data_ptr(), ws.numel(),
                                  chunk, N, C, h, w, float(intra_weight), _stream())
    _lib.check(rc, "fresco_opt_loss_grad")
    return loss, grad


def gram_target(x, workspace=None):
    """Cosine Gram matrix of (B,C,h,w) features -> (B,hw,hw) fp32 (fresco_gram_target)."""
    _need_gpu(x)
    x = _f32c(x)
    B, C, h, w = x.shape
    hw = h * w
    out = torch.empty(B, hw, hw, dtype=torch.float32, device=x.device)
    nbytes = (B * C * hw * 4 + 255) // 256 * 256 + 33 * ((B * hw * 4 + 255) // 256 * 256) + 512
    ws = (workspace or _default_ws).get(nbytes, x.device)
    rc = _lib.load().fresco_gram_target(x.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), B, C, hw,
                                        _stream())
    _lib.check(rc, "fresco_gram_target")
    return out
