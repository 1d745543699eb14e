"""Tensor-level wrappers over the C ABI (include/openclip_hip.h).

PyTorch is plumbing here: it owns device memory (caching allocator) and the stream.  Every wrapper checks
device / dtype / contiguity, then enqueues the HIP kernel on ``torch.cuda.current_stream()``.  There is no
CPU or ATen fallback: a CPU tensor or a missing library raises.
"""
import torch

from . import _lib

EPI_BF16, EPI_BIAS_GELU, EPI_BIAS_RESID_F32, EPI_DGELU, EPI_F32 = 0, 1, 2, 3, 4
EPI_BIAS_QUICKGELU = 7  # EPI_BIAS_GELU with x * sigmoid(1.702 x) (layers.py:29-32); its backward is the same EPI_DGELU
EPI_BIAS_RESID_BF16 = 8  # out bf16 = bf16(resid bf16 + bf16(acc + bias)): the residual add on the image tower's bf16 stream (as the reference's autocast)
BF16, F32 = torch.bfloat16, torch.float32


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, dtype, name):
    if t is None:
        return 0
    if not t.is_cuda:
        raise RuntimeError(f"open_clip_amd: '{name}' must live on the MI355X (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"open_clip_amd: '{name}' must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise RuntimeError(f"open_clip_amd: '{name}' must be contiguous")
    return t.data_ptr()


def _chk2d(t, dtype, name):
    """2-D, unit inner stride, arbitrary row stride -> (ptr, ld)"""
    if not t.is_cuda:
        raise RuntimeError(f"open_clip_amd: '{name}' must live on the MI355X (got {t.device}); there is no CPU path")
    if t.dtype != dtype or t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError(f"open_clip_amd: '{name}' must be a 2-D {dtype} matrix with unit inner stride")
    return t.data_ptr(), t.stride(0)


def empty(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


# ---- GEMMs ------------------------------------------------------------------------------------------
def gemm_nt(epi, a, b, out, bias=None, resid=None, aux=None, alpha=1.0):
    """out[M,N] = a[M,K] @ b[N,K]^T (+ epilogue); a, b bf16; out bf16 or fp32 depending on ``epi``.  ``aux`` (uint8 [M,N]) holds the
    GELU derivative in 8-bit fixed point (written by EPI_BIAS_GELU, read by EPI_DGELU; see ``dgelu_encode`` / ``dgelu_decode``)."""
    pa, lda = _chk2d(a, BF16, "a")
    pb, ldb = _chk2d(b, BF16, "b")
    odt = F32 if epi in (EPI_BIAS_RESID_F32, EPI_F32) else BF16
    rdt = BF16 if epi == EPI_BIAS_RESID_BF16 else F32
    po, ldc = _chk2d(out, odt, "out")
    M, K = a.shape
    N = b.shape[0]
    if b.shape[1] != K or out.shape[0] != M or out.shape[1] != N:
        raise RuntimeError(f"gemm_nt: shape mismatch a{tuple(a.shape)} b{tuple(b.shape)} out{tuple(out.shape)}")
    if resid is not None and (resid.shape != out.shape or resid.stride(0) != ldc):
        raise RuntimeError("gemm_nt: resid must match out")
    if aux is not None and (aux.shape != out.shape or aux.stride(0) != ldc):
        raise RuntimeError("gemm_nt: aux must match out")
    _lib.call("ocn_gemm_nt", epi, pa, lda, pb, ldb, po, ldc, M, N, K, _chk(bias, F32, "bias"),
              0 if resid is None else _chk2d(resid, rdt, "resid")[0], 0 if aux is None else _chk2d(aux, torch.uint8, "aux")[0],
              float(alpha), _stream())
    return out


def set_tile_rescue(on: bool):
    """``ocn_set_tile_rescue``: the persistent GEMMs' multi-GPU form (finishing workgroups hand out the shares of workgroups whose CU is held by a
    collective's kernel; include/openclip_hip.h).  Process-wide; same results (NT bit-identical, wgrad up to the order of its fp32 atomics)."""
    _lib.call("ocn_set_tile_rescue", int(bool(on)))


def tile_rescue() -> bool:
    return bool(_lib.load().ocn_get_tile_rescue())


def multi_gpu_defaults(world_size: int):
    """What a data-parallel run switches on in the kernels (called by the distributed losses and by ``NativeGradSync`` when world_size > 1, i.e. wherever
    collectives' kernels will hold CUs while GEMMs are launched): the GEMMs' rescue form.  ``set_tile_rescue(False)`` afterwards switches it off again;
    the environment variable OCN_TILE_RESCUE=0 keeps it off."""
    import os
    if int(world_size) > 1 and torch.cuda.is_available() and os.environ.get("OCN_TILE_RESCUE", "1") != "0":
        set_tile_rescue(True)


def gemm_nt_splitk_plan(M, N, K):
    """K-slices ``gemm_nt_splitk`` should use for out[M,N] = a[M,K] @ b[N,K]^T (1: the product fills the chip as it is -- use ``gemm_nt``)"""
    return int(_lib.load().ocn_gemm_nt_splitk_plan(int(M), int(N), int(K)))


def gemm_nt_splitk(a, b, out, ksplit, rowscale=None, sub_rows=None, sub_alpha=0.0, scale=None):
    """out[M,N] fp32 = scale * (rowscale[:, None] * (a[M,K] @ b[N,K]^T) - sub_alpha * sub_rows) with K cut into ``ksplit`` slices that run as tiles of their own
    (few output tiles, long K: ocn_gemm_nt_splitk); ``rowscale`` fp32 [M], ``sub_rows`` bf16 [M,N], ``scale`` a 1-element fp32 device tensor, each optional"""
    pa, lda = _chk2d(a, BF16, "a")
    pb, ldb = _chk2d(b, BF16, "b")
    po, ldc = _chk2d(out, F32, "out")
    M, K = a.shape
    N = b.shape[0]
    if b.shape[1] != K or tuple(out.shape) != (M, N):
        raise RuntimeError(f"gemm_nt_splitk: shape mismatch a{tuple(a.shape)} b{tuple(b.shape)} out{tuple(out.shape)}")
    ws = torch.empty(int(ksplit) * M * ldc, dtype=F32, device=a.device)
    ps, lds = (0, 0) if sub_rows is None else _chk2d(sub_rows, BF16, "sub_rows")
    _lib.call("ocn_gemm_nt_splitk", pa, lda, pb, ldb, po, ldc, M, N, K, int(ksplit), ws.data_ptr(), _chk(rowscale, F32, "rowscale"), ps, lds, float(sub_alpha),
              _chk(scale, F32, "scale"), _stream())
    return out


def scale_rows_bf16(x16, scale):
    """bf16(scale[r] * x16[r, :])"""
    px, ldx = _chk2d(x16, BF16, "x16")
    out = empty(x16.shape, BF16, x16)
    _lib.call("ocn_scale_rows_bf16", px, ldx, _chk(scale, F32, "scale"), out.data_ptr(), out.stride(0), x16.shape[0], x16.shape[1], _stream())
    return out


def sub_scaled_rows(out, x16, scale, alpha):
    """out[r, :] -= (alpha / scale[r]) * x16[r, :]   (out fp32 [R, E], a row-contiguous view is fine)"""
    po, ldo = _chk2d(out, F32, "out")
    px, ldx = _chk2d(x16, BF16, "x16")
    _lib.call("ocn_sub_scaled_rows", po, ldo, px, ldx, _chk(scale, F32, "scale"), float(alpha), x16.shape[0], x16.shape[1], _stream())
    return out


DGELU_SCALE, DGELU_OFFSET = 200.0, 0.13  # csrc/ocn_common.h: q = round((gelu' + 0.13) * 200), |error| <= 0.0025


def dgelu_decode(q):
    """the values the backward epilogue multiplies with, from the stored bytes (test / tooling helper; torch arithmetic)"""
    return q.float() / DGELU_SCALE - DGELU_OFFSET


def dgelu_encode(d):
    """bytes for given derivative values in [-0.13, 1.145] (test / tooling helper)"""
    return torch.round((d.float() + DGELU_OFFSET) * DGELU_SCALE).clamp_(0, 255).to(torch.uint8)


def gemm_tn_accum(a, b, dw, dbias=None, alpha=1.0, deterministic=False):
    """dw[N,K] += alpha * a[M,N]^T @ b[M,K]; dbias[N] += alpha * colsum(a).  a, b bf16; dw, dbias fp32.
    ``deterministic``: the reproducible form (ocn_gemm_tn_accum_det: per-split slabs summed in a fixed order instead of fp32 atomics)"""
    pa, lda = _chk2d(a, BF16, "a")
    pb, ldb = _chk2d(b, BF16, "b")
    pw, ldw = _chk2d(dw, F32, "dw")
    M, N = a.shape
    K = b.shape[1]
    if b.shape[0] != M or dw.shape[0] != N or dw.shape[1] != K:
        raise RuntimeError(f"gemm_tn_accum: shape mismatch a{tuple(a.shape)} b{tuple(b.shape)} dw{tuple(dw.shape)}")
    if deterministic:
        need = _lib.load().ocn_gemm_tn_det_workspace_bytes(M, N, K)
        ws = torch.empty(need, dtype=torch.uint8, device=a.device) if need > 0 else None
        _lib.call("ocn_gemm_tn_accum_det", pa, lda, pb, ldb, pw, ldw, M, N, K, _chk(dbias, F32, "dbias"), float(alpha),
                  0 if ws is None else ws.data_ptr(), need, _stream())
        return dw
    _lib.call("ocn_gemm_tn_accum", pa, lda, pb, ldb, pw, ldw, M, N, K, _chk(dbias, F32, "dbias"), float(alpha), _stream())
    return dw


def gemm_tn_accum2(a1, b1, dw1, dbias1, a2, b2, dw2, dbias2, alpha=1.0):
    """two weight gradients over the same rows and the same K in ONE launch (ocn_gemm_tn_accum2):
    dw1 += alpha * a1^T @ b1, dw2 += alpha * a2^T @ b2 (+ both bias gradients or neither)"""
    M, K = b1.shape
    if a1.shape[0] != M or a2.shape[0] != M or tuple(b2.shape) != (M, K) or tuple(dw1.shape) != (a1.shape[1], K) or tuple(dw2.shape) != (a2.shape[1], K):
        raise RuntimeError(f"gemm_tn_accum2: shape mismatch a1{tuple(a1.shape)} b1{tuple(b1.shape)} dw1{tuple(dw1.shape)} "
                           f"a2{tuple(a2.shape)} b2{tuple(b2.shape)} dw2{tuple(dw2.shape)}")
    if (dbias1 is None) != (dbias2 is None):
        raise RuntimeError("gemm_tn_accum2: both bias gradients or neither")
    pa1, lda1 = _chk2d(a1, BF16, "a1")
    pb1, ldb1 = _chk2d(b1, BF16, "b1")
    pw1, ldw1 = _chk2d(dw1, F32, "dw1")
    pa2, lda2 = _chk2d(a2, BF16, "a2")
    pb2, ldb2 = _chk2d(b2, BF16, "b2")
    pw2, ldw2 = _chk2d(dw2, F32, "dw2")
    _lib.call("ocn_gemm_tn_accum2", pa1, lda1, pb1, ldb1, pw1, ldw1, _chk(dbias1, F32, "dbias1"), a1.shape[1],
              pa2, lda2, pb2, ldb2, pw2, ldw2, _chk(dbias2, F32, "dbias2"), a2.shape[1], M, K, float(alpha), _stream())
    return dw1, dw2


# ---- casts --------------------------------------------------------------------------------------------
def cast_bf16(src, out=None):
    out = empty(src.shape, BF16, src) if out is None else out
    _lib.call("ocn_cast_f32_bf16", _chk(src, F32, "src"), _chk(out, BF16, "out"), src.numel(), _stream())
    return out


def cast_bf16_scaled(src, scale, out=None):
    """bf16(src * scale) with ``scale`` a 1-element fp32 DEVICE tensor (no host read of its value)"""
    out = empty(src.shape, BF16, src) if out is None else out
    _lib.call("ocn_cast_f32_bf16_scaled", _chk(src, F32, "src"), _chk(out, BF16, "out"), src.numel(), _chk(scale, F32, "scale"), _stream())
    return out


def cast_transpose_bf16(src, out=None):
    R, C = src.shape
    out = empty((C, R), BF16, src) if out is None else out
    _lib.call("ocn_cast_transpose_f32_bf16", _chk(src, F32, "src"), _chk(out, BF16, "out"), R, C, _stream())
    return out


# ---- LayerNorm -----------------------------------------------------------------------------------------
def layernorm_fwd(x, w, b, want_bf16=True, want_f32=False, eps=1e-5):
    """``x`` fp32 or bf16 (the image tower's bf16 residual stream); statistics and arithmetic are fp32 either way"""
    M, C = x.shape
    y16 = empty((M, C), BF16, x) if want_bf16 else None
    y32 = empty((M, C), F32, x) if want_f32 else None
    mean, rstd = empty((M,), F32, x), empty((M,), F32, x)
    x16 = x.dtype == BF16
    _lib.call("ocn_layernorm_fwd", _chk(x, BF16 if x16 else F32, "x"), int(x16), _chk(w, F32, "w"), _chk(b, F32, "b"), _chk(y16, BF16, "y16"),
              _chk(y32, F32, "y32"), _chk(mean, F32, "mean"), _chk(rstd, F32, "rstd"), M, C, float(eps), _stream())
    return y16, y32, mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd, dw, db, dres=None, want_f32=True, want_bf16=False, dcol=None, deterministic=False):
    """(dx fp32 | None, dx bf16 | None); ``dres`` = the residual branch's fp32 gradient, added in; dw / db are accumulated into;
    ``dcol`` (fp32 [C], accumulated into): column sums of dx in fp32 = the bias gradient of the linear in front of this LayerNorm's input;
    ``deterministic``: dw / db / dcol through per-workgroup slabs added in a fixed order instead of fp32 atomics.
    ``x`` fp32 or bf16; with a bf16 ``x`` (image tower, bf16 residual stream) ``dy`` is bf16 and ``dres`` bf16 or fp32"""
    M, C = x.shape
    is32 = dy.dtype == F32
    x16 = x.dtype == BF16
    dr16 = dres is not None and dres.dtype == BF16
    dx32 = empty((M, C), F32, x) if want_f32 else None
    dx16 = empty((M, C), BF16, x) if want_bf16 else None
    ws = empty((_lib.load().ocn_layernorm_bwd_det_workspace_floats(M, C),), F32, x) if deterministic else None
    _lib.call("ocn_layernorm_bwd", _chk(dy, F32 if is32 else BF16, "dy"), int(is32), _chk(x, BF16 if x16 else F32, "x"), int(x16), _chk(w, F32, "w"),
              _chk(mean, F32, "mean"), _chk(rstd, F32, "rstd"), _chk(dres, BF16 if dr16 else F32, "dres"), int(dr16), _chk(dx32, F32, "dx32"), _chk(dx16, BF16, "dx16"),
              _chk(dw, F32, "dw"), _chk(db, F32, "db"), _chk(dcol, F32, "dcol"), _chk(ws, F32, "det_workspace"), M, C, _stream())
    return dx32, dx16


def colsum_f32(x, out, deterministic=False):
    """out[C] += column sums of x [R, C] (fp32); ``deterministic``: a single, fixed order of additions"""
    R, C = x.shape
    _lib.call("ocn_colsum_f32", _chk(x, F32, "x"), _chk(out, F32, "out"), R, C, int(deterministic), _stream())
    return out


# ---- attention -------------------------------------------------------------------------------------------
class SeqLayout:
    """layout of a packed ("varlen") batch as the attention launches take it: ``seq_off`` (int32 [B+1], device) and, optionally, the
    buckets of ocn_seq_bucket_plan -- ``order`` (int32 [B], device) + ``counts`` (host ints, one per 32-row block count)"""
    __slots__ = ("seq_off", "order", "counts", "_c")

    def __init__(self, seq_off, order=None, counts=None):
        import ctypes
        self.seq_off, self.order, self.counts = seq_off, order, (list(counts) if counts is not None else None)
        self._c = (ctypes.c_int32 * len(self.counts))(*self.counts) if self.counts is not None else None

    def args(self, B, L):
        so = _chk(self.seq_off, torch.int32, "seq_off")
        if self.order is None:
            return so, 0, 0
        if self.order.numel() != B or len(self.counts) != (L + 31) // 32 or sum(self.counts) != B:
            raise RuntimeError(f"SeqLayout: order of {self.order.numel()} ids / counts {self.counts} do not describe B={B}, L={L}")
        return so, _chk(self.order, torch.int32, "order"), self._c


def _layout(seq_off):
    return seq_off if isinstance(seq_off, SeqLayout) else SeqLayout(seq_off)


def attn_fwd(qkv, B, L, H, causal, scale, head_dim=64, seq_off=None):
    """``seq_off`` (int32 [B+1] on the device, or a SeqLayout): packed batch -- sequence b owns rows seq_off[b]..seq_off[b+1] of qkv
    (head_dim 64 only)"""
    C = H * head_dim
    if seq_off is not None:
        lay = _layout(seq_off)
        if head_dim != 64 or qkv.shape[1] != 3 * C or lay.seq_off.numel() != B + 1:
            raise RuntimeError(f"attn_fwd(varlen): needs head_dim 64 and seq_off of B+1 entries (qkv {tuple(qkv.shape)}, head_dim {head_dim})")
        out = empty((qkv.shape[0], C), BF16, qkv)
        lse = empty((B * H * L,), F32, qkv)
        so, order, counts = lay.args(B, L)
        _lib.call("ocn_attn_fwd_varlen", _chk(qkv, BF16, "qkv"), _chk(out, BF16, "out"), _chk(lse, F32, "lse"), so, order, counts,
                  B, L, H, int(causal), float(scale), _stream())
        return out, lse
    if qkv.shape != (B * L, 3 * C):
        raise RuntimeError(f"attn_fwd: qkv shape {tuple(qkv.shape)} != {(B * L, 3 * C)} (H={H}, head_dim={head_dim})")
    out = empty((B * L, C), BF16, qkv)
    lse = empty((B * H * L,), F32, qkv)
    _lib.call("ocn_attn_fwd_hd", _chk(qkv, BF16, "qkv"), _chk(out, BF16, "out"), _chk(lse, F32, "lse"), B, L, H, head_dim, int(causal),
              float(scale), _stream())
    return out, lse


def attn_bwd(qkv, out, dout, lse, B, L, H, causal, scale, head_dim=64, seq_off=None):
    dqkv = empty(qkv.shape, BF16, qkv)
    if seq_off is not None:
        so, order, counts = _layout(seq_off).args(B, L)
        _lib.call("ocn_attn_bwd_varlen", _chk(qkv, BF16, "qkv"), _chk(out, BF16, "out"), _chk(dout, BF16, "dout"), _chk(lse, F32, "lse"),
                  _chk(dqkv, BF16, "dqkv"), so, order, counts, B, L, H, int(causal), float(scale), _stream())
        return dqkv
    delta = empty((B * H * L,), F32, qkv)  # workspace of the generic path (exchanged between its two launches)
    _lib.call("ocn_attn_bwd_hd", _chk(qkv, BF16, "qkv"), _chk(out, BF16, "out"), _chk(dout, BF16, "dout"), _chk(lse, F32, "lse"),
              _chk(dqkv, BF16, "dqkv"), _chk(delta, F32, "delta"), B, L, H, head_dim, int(causal), float(scale), _stream())
    return dqkv


def attn_pooled_fwd(q, kv, rows, B, L, H, causal, scale, seq_off=None):
    """single-query attention of a tower's last block (head_dim 64): q [B, C] = the pooled rows' queries, kv [M, 2C] = K | V of every row;
    ``rows`` int32 [B] = absolute row of each sequence's pooled token; ``seq_off`` as in attn_fwd (packed rows).  -> out [B, C], lse [B*H]"""
    C = H * 64
    if q.shape != (B, C) or kv.shape[1] != 2 * C:
        raise RuntimeError(f"attn_pooled_fwd: q {tuple(q.shape)} / kv {tuple(kv.shape)} do not match B={B}, H={H}, head_dim 64")
    so = _chk(_layout(seq_off).seq_off, torch.int32, "seq_off") if seq_off is not None else 0
    out = empty((B, C), BF16, q)
    lse = empty((B * H,), F32, q)
    _lib.call("ocn_attn_pooled_fwd", _chk(q, BF16, "q"), _chk(kv, BF16, "kv"), _chk(out, BF16, "out"), _chk(lse, F32, "lse"), so,
              _chk(rows, torch.int32, "rows"), B, L, H, int(causal), float(scale), _stream())
    return out, lse


def attn_pooled_bwd(q, kv, out, dout, lse, rows, B, L, H, causal, scale, seq_off=None):
    """-> dq [B, C] bf16, dkv [M, 2C] bf16.  The kernel writes every key row of every sequence; rows of ``kv`` that belong to NO sequence can only
    exist with a packed layout whose last offset is below M -- the model never builds one (_TextPack allocates M = seq_off[B] rows, read from the
    same plan) -- and the dense layout must cover kv exactly, which is checked here: dkv feeds the K,V weight-gradient GEMM, an unwritten row
    would be uninitialised memory."""
    so = _chk(_layout(seq_off).seq_off, torch.int32, "seq_off") if seq_off is not None else 0
    C = H * 64
    if q.shape != (B, C) or kv.shape[1] != 2 * C or (seq_off is None and kv.shape[0] != B * L):
        raise RuntimeError(f"attn_pooled_bwd: q {tuple(q.shape)} / kv {tuple(kv.shape)} do not match B={B}, L={L}, H={H}, head_dim 64")
    dq = empty(q.shape, BF16, q)
    dkv = empty(kv.shape, BF16, kv)  # packed: M == seq_off[B] (model._TextPack takes M from the plan that wrote seq_off), so no row stays unwritten
    _lib.call("ocn_attn_pooled_bwd", _chk(q, BF16, "q"), _chk(kv, BF16, "kv"), _chk(out, BF16, "out"), _chk(dout, BF16, "dout"), _chk(lse, F32, "lse"),
              _chk(dq, BF16, "dq"), _chk(dkv, BF16, "dkv"), so, _chk(rows, torch.int32, "rows"), B, L, H, int(causal), float(scale), _stream())
    return dq, dkv


# ---- embeddings / pooling ----------------------------------------------------------------------------------
def patchify(image, P, Kpad):
    B, Cin, H, W = image.shape
    if Cin != 3:
        raise RuntimeError("patchify: images must have 3 channels")
    is16 = image.dtype == BF16
    out = empty((B * (H // P) * (W // P), Kpad), BF16, image)
    _lib.call("ocn_patchify", _chk(image, BF16 if is16 else F32, "image"), int(is16), _chk(out, BF16, "patches"), B, H, W, P, Kpad, _stream())
    return out


def patchify_u8(image, P, Kpad, mean, std, hwc):
    """uint8 pixels -> normalised bf16 patch matrix (ToTensor + Normalize fused; see ocn_patchify_u8)"""
    import ctypes
    if image.dtype != torch.uint8 or image.dim() != 4:
        raise RuntimeError("patchify_u8: expected a 4-D uint8 image batch")
    B = image.shape[0]
    H, W = (image.shape[1], image.shape[2]) if hwc else (image.shape[2], image.shape[3])
    if (image.shape[3] if hwc else image.shape[1]) != 3:
        raise RuntimeError("patchify_u8: images must have 3 channels")
    out = empty((B * (H // P) * (W // P), Kpad), BF16, image)
    m3, s3 = (ctypes.c_float * 3)(*[float(v) for v in mean]), (ctypes.c_float * 3)(*[float(v) for v in std])
    _lib.call("ocn_patchify_u8", _chk(image, torch.uint8, "image"), int(hwc), m3, s3, _chk(out, BF16, "patches"), B, H, W, P, Kpad, _stream())
    return out


def embed_assemble_fwd(patch_out, cls, pos, B, G, C):
    emb = empty((B * (G + 1), C), F32, patch_out)
    _lib.call("ocn_embed_assemble_fwd", _chk(patch_out, F32, "patch_out"), _chk(cls, F32, "cls"), _chk(pos, F32, "pos"),
              _chk(emb, F32, "emb"), B, G, C, _stream())
    return emb


def embed_assemble_bwd(demb, dpos, dcls, B, G, C, deterministic=False):
    dpatch = empty((B * G, C), BF16, demb)
    _lib.call("ocn_embed_assemble_bwd", _chk(demb, F32, "demb"), _chk(dpatch, BF16, "dpatch"), _chk(dpos, F32, "dpos"),
              _chk(dcls, F32, "dcls"), B, G, C, int(deterministic), _stream())
    return dpatch


def token_embed_fwd(text, table, pos):
    B, L = text.shape
    vocab, C = table.shape
    x = empty((B * L, C), F32, table)
    _lib.call("ocn_token_embed_fwd", _chk(text, torch.int64, "text"), _chk(table, F32, "table"), _chk(pos, F32, "pos"),
              _chk(x, F32, "x"), B, L, C, vocab, _stream())
    return x


def token_embed_bwd(text, dx, dtable, dpos):
    B, L = text.shape
    vocab, C = dtable.shape
    _lib.call("ocn_token_embed_bwd", _chk(text, torch.int64, "text"), _chk(dx, F32, "dx"), _chk(dtable, F32, "dtable"),
              _chk(dpos, F32, "dpos"), B, L, C, vocab, _stream())


def token_embed_bwd_sorted(text, dx, dtable, dpos, deterministic=False):
    """segment-reduce form of token_embed_bwd: ``dtable`` must be zero on entry.  The device sort of the B*L ids is index plumbing
    (torch.sort); all arithmetic is in ocn_token_embed_bwd_sorted."""
    B, L = text.shape
    vocab, C = dtable.shape
    keys, order = torch.sort(text.reshape(-1), stable=bool(deterministic))  # the reproducible form sums a run in sorted order: equal ids must keep their order
    is16 = dx.dtype == BF16
    _lib.call("ocn_token_embed_bwd_sorted", _chk(keys, torch.int64, "sorted_tokens"), _chk(order, torch.int64, "order"),
              _chk(dx, BF16 if is16 else F32, "dx"), int(is16), _chk(dtable, F32, "dtable"), _chk(dpos, F32, "dpos"), B, L, C, vocab,
              int(deterministic), _stream())


def seq_pack_plan(text, vocab=None, buckets=False):
    """(eot [B], plan, last_row [B], order) int32 on the device: the packed layout of a text batch.  plan[0..B] = seq_off
    (ocn_seq_pack_plan; plan[B] = the packed row count M), plan[B+1] = the number of ids outside [0, vocab) when ``vocab`` is given
    (ocn_token_range_check), plan[B+2 ..] = the ceil(L / 32) bucket counts when ``buckets`` (ocn_seq_bucket_plan; ``order`` = the
    sequence ids grouped by block count, else None)"""
    B, L = text.shape
    nbk = (L + 31) // 32 if buckets else 0
    eot, plan, last_row = empty((B,), torch.int32, text), empty((B + 2 + nbk,), torch.int32, text), empty((B,), torch.int32, text)
    _lib.call("ocn_seq_pack_plan", _chk(text, torch.int64, "text"), _chk(eot, torch.int32, "eot"), _chk(plan, torch.int32, "seq_off"),
              _chk(last_row, torch.int32, "last_row"), B, L, _stream())
    if vocab is not None:
        _lib.call("ocn_token_range_check", text.data_ptr(), B * L, int(vocab), plan.data_ptr() + 4 * (B + 1), _stream())
    else:
        plan[B + 1:B + 2].zero_()
    order = None
    if buckets:
        order = empty((B,), torch.int32, text)
        _lib.call("ocn_seq_bucket_plan", plan.data_ptr(), _chk(order, torch.int32, "order"), plan.data_ptr() + 4 * (B + 2), B, L, _stream())
    return eot, plan, last_row, order


def token_range_check(text, vocab):
    """int32 [1] on the device: how many ids of ``text`` lie outside [0, vocab)"""
    bad = empty((1,), torch.int32, text)
    _lib.call("ocn_token_range_check", _chk(text, torch.int64, "text"), text.numel(), int(vocab), _chk(bad, torch.int32, "bad"), _stream())
    return bad


def seq_pack_rows(text, seq_off, M):
    B, L = text.shape
    tokens, posidx = empty((M,), torch.int64, text), empty((M,), torch.int32, text)
    _lib.call("ocn_seq_pack_rows", _chk(text, torch.int64, "text"), _chk(seq_off, torch.int32, "seq_off"), _chk(tokens, torch.int64, "tokens"),
              _chk(posidx, torch.int32, "posidx"), B, L, _stream())
    return tokens, posidx


def token_embed_fwd_rows(tokens, posidx, table, pos):
    M = tokens.numel()
    vocab, C = table.shape
    x = empty((M, C), F32, table)
    _lib.call("ocn_token_embed_fwd_rows", _chk(tokens, torch.int64, "tokens"), _chk(posidx, torch.int32, "posidx"), _chk(table, F32, "table"),
              _chk(pos, F32, "pos"), _chk(x, F32, "x"), M, C, vocab, _stream())
    return x


def token_embed_bwd_sorted_varlen(tokens, seq_off, B, L, dx, dtable, dpos, deterministic=False):
    """packed-row form of token_embed_bwd_sorted (``tokens`` = the M packed ids; ``dtable`` zero on entry)"""
    vocab, C = dtable.shape
    M = tokens.numel()
    keys, order = torch.sort(tokens, stable=bool(deterministic))
    is16 = dx.dtype == BF16
    _lib.call("ocn_token_embed_bwd_sorted_varlen", _chk(keys, torch.int64, "sorted_tokens"), _chk(order, torch.int64, "order"),
              _chk(dx, BF16 if is16 else F32, "dx"), int(is16), _chk(dtable, F32, "dtable"), _chk(dpos, F32, "dpos"),
              _chk(seq_off, torch.int32, "seq_off"), B, L, M, C, vocab, int(deterministic), _stream())


def argmax_rows(text):
    B, L = text.shape
    idx = empty((B,), torch.int32, text)
    _lib.call("ocn_argmax_rows", _chk(text, torch.int64, "text"), _chk(idx, torch.int32, "idx"), B, L, _stream())
    return idx


def gather_rows(x, idx, B, L):
    """fp32 [B, C] rows of ``x`` (fp32, or bf16: the image tower's bf16 residual stream)"""
    C = x.shape[1]
    out = empty((B, C), F32, x)
    x16 = x.dtype == BF16
    _lib.call("ocn_gather_rows", _chk(x, BF16 if x16 else F32, "x"), int(x16), _chk(idx, torch.int32, "idx"), _chk(out, F32, "out"), B, L, C, _stream())
    return out


def gather_rows_bf16(x, idx, B, L):
    C = x.shape[1]
    out = empty((B, C), BF16, x)
    _lib.call("ocn_gather_rows_bf16", _chk(x, BF16, "x"), _chk(idx, torch.int32, "idx"), _chk(out, BF16, "out"), B, L, C, _stream())
    return out


def scatter_rows(d, idx, dx, B, L, dx16=None):
    C = d.shape[1]
    _lib.call("ocn_scatter_rows", _chk(d, F32, "d"), _chk(idx, torch.int32, "idx"), _chk(dx, F32, "dx"), _chk(dx16, BF16, "dx16"), B, L, C, _stream())
    return dx


def scatter_add_rows(d, idx, dx, B, L, dx16=None):
    """dx[row_b] += d[b]; dx16[row_b] = bf16(dx[row_b]); ``dx`` None (bf16 gradient stream): dx16[row_b] = bf16(dx16[row_b] + d[b])"""
    C = d.shape[1]
    _lib.call("ocn_scatter_add_rows", _chk(d, F32, "d"), _chk(idx, torch.int32, "idx"), _chk(dx, F32, "dx"), _chk(dx16, BF16, "dx16"), B, L, C, _stream())
    return dx


def l2norm_fwd(x, eps=1e-12):
    B, E = x.shape
    y, y16, inv = empty((B, E), F32, x), empty((B, E), BF16, x), empty((B,), F32, x)
    _lib.call("ocn_l2norm_fwd", _chk(x, F32, "x"), _chk(y, F32, "y"), _chk(y16, BF16, "y16"), _chk(inv, F32, "inv"), B, E, float(eps), _stream())
    return y, y16, inv


def l2norm_bwd(dy, y, inv):
    B, E = y.shape
    dx = empty((B, E), F32, y)
    _lib.call("ocn_l2norm_bwd", _chk(dy, F32, "dy"), _chk(y, F32, "y"), _chk(inv, F32, "inv"), _chk(dx, F32, "dx"), B, E, _stream())
    return dx


# ---- losses ---------------------------------------------------------------------------------------------------
def softmax_ce_rows(logits, G, N, label_offset, loss_scale, grad_scale, inv_logit_scale, loss_sum, dscale_sum, det_rows=None):
    """``det_rows`` (fp32 [R, 3]): the rows' contributions are written there instead of being added atomically (reproducible form)"""
    pl, ld = _chk2d(logits, F32, "logits")
    pg, ldg = _chk2d(G, BF16, "G")
    _lib.call("ocn_softmax_ce_rows", pl, ld, pg, ldg, logits.shape[0], N, int(label_offset), float(loss_scale), float(grad_scale),
              float(inv_logit_scale), _chk(loss_sum, F32, "loss_sum"), _chk(dscale_sum, F32, "dscale_sum"), _chk(det_rows, F32, "det_rows"), _stream())


def fused_logits_ce(xs16, y16, G, N, label_offset, loss_scale, grad_scale, loss_sum, dscale_sum):
    """cross-entropy of the rows of xs16 @ y16[:N]^T against arange + label_offset without materialising the logits, in ONE pass of the MFMA GEMM
    (ocn_fused_logits_ce): fills G bf16 [R, ldg] with exp(logit - shift_r) and returns ``rowscale`` fp32 [R] = grad_scale / sum_j G[r, j] -- the softmax part
    of the logit gradient is G * rowscale[:, None]; accumulates loss_sum and dscale_sum (= sum((softmax - onehot) * grad_scale * logits))"""
    px, ldx = _chk2d(xs16, BF16, "xs16")
    py, ldy = _chk2d(y16, BF16, "y16")
    pg, ldg = _chk2d(G, BF16, "G")
    R, E = xs16.shape
    ws = torch.empty(_lib.load().ocn_fused_logits_ce_workspace_floats(R, N), dtype=F32, device=xs16.device)
    rowscale = torch.empty(R, dtype=F32, device=xs16.device)
    _lib.call("ocn_fused_logits_ce", px, ldx, py, ldy, R, N, E, int(label_offset), float(loss_scale), float(grad_scale), pg, ldg,
              ws.data_ptr(), rowscale.data_ptr(), _chk(loss_sum, F32, "loss_sum"), _chk(dscale_sum, F32, "dscale_sum"), _stream())
    return rowscale


def fused_logits_ce_supported(R, N, E):
    return E % 128 == 0 and N % 8 == 0 and R >= 256


def siglip_rows(logits, G, N, label_offset, negative_only, bias, loss_scale, grad_scale, inv_logit_scale, loss_sum, dscale_sum, dbias_sum, det_rows=None):
    """``bias``: the bias the logits carry -- a python float, or a 1-element fp32 DEVICE tensor (no host read); dscale_sum += sum(g * (logits - bias))"""
    pl, ld = _chk2d(logits, F32, "logits")
    pg, ldg = _chk2d(G, BF16, "G")
    bias_dev = _chk(bias, F32, "bias") if torch.is_tensor(bias) else 0
    _lib.call("ocn_siglip_rows", pl, ld, pg, ldg, logits.shape[0], N, int(label_offset), int(negative_only), 0.0 if bias_dev else float(bias), bias_dev,
              float(loss_scale), float(grad_scale), float(inv_logit_scale), _chk(loss_sum, F32, "loss_sum"),
              _chk(dscale_sum, F32, "dscale_sum"), _chk(dbias_sum, F32, "dbias_sum"), _chk(det_rows, F32, "det_rows"), _stream())


# ---- optimizer ----------------------------------------------------------------------------------------------------
def sumsq_accum(x, out):
    _lib.call("ocn_sumsq_accum", _chk(x, F32, "x"), x.numel(), _chk(out, F32, "out"), _stream())


def adamw_step(w, g, m, v, lr, beta1, beta2, eps, weight_decay, step, w_bf16=None, clip_coef=None):
    _lib.call("ocn_adamw_step", _chk(w, F32, "w"), _chk(g, F32, "g"), _chk(m, F32, "m"), _chk(v, F32, "v"),
              _chk(w_bf16, BF16, "w_bf16"), w.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
              int(step), _chk(clip_coef, F32, "clip_coef"), _stream())
