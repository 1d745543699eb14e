"""Native CLIP model: the reference's ``CLIP`` object surface over the HIP hot path.

``NativeCLIP`` keeps the reference's attribute names, parameter names/shapes and ``state_dict`` layout
(reference ``src/open_clip/model.py:318-548``, ``transformer.py``) so that
``open_clip.task.CLIPTask`` / ``open_clip_train.train.train_one_epoch`` drive it unmodified and reference
checkpoints load with ``load_state_dict``.  Modules here are *parameter containers*; all arithmetic happens in
``torch.autograd.Function``s that enqueue hand-written gfx950 kernels through the C ABI (``ops.py``).

Precision policy (mirrors ``--precision amp_bf16``, precision.py:6-16): fp32 master weights, bf16 GEMM /
attention operands with fp32 accumulation, fp32 LayerNorm / softmax / loss statistics.  Residual stream: fp32 in the text tower (what autocast
leaves there: the token embedding is fp32, model.py:399-401); in the image tower fp32 (``image_stream="fp32"``, the stricter default) or bf16
(``image_stream="bf16"``) -- what the reference's own autocast produces there: conv1 runs under autocast (transformer.py:794 -> bf16), LayerNorm
casts back to its input dtype (layers.py:23-26) and `q_x + attention(...)` adds two bf16 tensors, so the reference's image stream AND its
gradient are bf16.
Autograd granularity is one Function per residual block, so gradients reach ``.grad`` (and DDP's bucket hooks)
block by block while the backward is still running, and block-granular recompute (``set_grad_checkpointing``)
is a flag of the same Function.
"""
import math
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .configs import get_model_config

F32, BF16 = torch.float32, torch.bfloat16


def _round_up(a, b):
    return (a + b - 1) // b * b


# ------------------------------------------------------------------------------------------------------
# bf16 operand copies of fp32 master weights, refreshed when the parameter's version counter moves
# ------------------------------------------------------------------------------------------------------
class _WeightCache:
    def __init__(self):
        self._d = {}
        self.pair_wgrad = True  # block backward: run the wgrad GEMMs on a side stream under the HBM-bound kernels (_Paired)
        self.deterministic = False  # weight / bias gradient GEMMs in their reproducible form (ocn_gemm_tn_accum_det)
        self.single_query = True  # pooled last block (head_dim 64): K, V projection only + single-query attention (_pooled_block_forward)
        self._side = _StreamMap()  # wgrad side streams of this tower, one per (device, main stream)
        self.twin_stats = {"hit": 0, "miss": 0}  # how often a block's backward found the bf16 twin of its incoming gradient (tests assert it does)
        # residual stream of THIS tower: "fp32" | "bf16" (stream and its gradient bf16: the reference's autocast in the image tower) |
        # "bf16-fp32grad" (bf16 stream; the gradient travels as bf16 with an fp32 companion for the residual path, see _publish_f32)
        self.stream = "fp32"

    def side_stream(self, dev):
        """the wgrad stream that belongs to the CURRENT stream (one per main stream: the towers may run on streams of their own)"""
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        st = self._side.get(key)
        if st is None:
            st = self._side[key] = torch.cuda.Stream(device=dev)
        return st

    def get(self, p: torch.Tensor, kind: str):
        """kind: 'n' = bf16 copy [rows, cols]; 't' = bf16 transpose [cols, rows] (2-D views of p)"""
        key = (p.data_ptr(), kind)
        ver = p._version
        hit = self._d.get(key)
        if hit is not None and hit[0] == ver and hit[1] == p.shape:
            return hit[2]
        w2 = p.detach().reshape(p.shape[0], -1)
        out = ops.cast_bf16(w2) if kind == "n" else ops.cast_transpose_bf16(w2)
        self._d[key] = (ver, p.shape, out)
        return out

    def peek(self, p, kind):
        """the cached copy of ``p`` (possibly stale) or None -- NativeAdamW rewrites these in its own launch"""
        hit = self._d.get((p.data_ptr(), kind))
        return hit[2] if hit is not None and hit[1] == p.shape else None

    def mark_fresh(self, p, n16, t16):
        """the optimizer has just rewritten these copies from the updated parameter: record its new version"""
        for kind, t in (("n", n16), ("t", t16)):
            hit = self._d.get((p.data_ptr(), kind))
            if t is not None and hit is not None and hit[2] is t:
                self._d[(p.data_ptr(), kind)] = (p._version, p.shape, t)

    def clear(self):
        self._d.clear()

    def __deepcopy__(self, memo):
        """a copied model (EMA twin, base_task.py:171) has new parameter addresses: none of the cached operand copies could ever hit
        there, so the copy starts empty instead of duplicating every bf16 weight"""
        new = _WeightCache()
        new.pair_wgrad, new.deterministic, new.single_query, new.stream = self.pair_wgrad, self.deterministic, self.single_query, self.stream
        return new


# ------------------------------------------------------------------------------------------------------
# bf16 twins of residual-stream gradients: the kernel that produces a block's input gradient (LayerNorm backward)
# also emits it in bf16, and the previous block's backward uses that as its GEMM operand instead of running a cast
# kernel over the fp32 gradient (saves one 4-byte read per element per block).  The twin TRAVELS WITH THE GRADIENT TENSOR
# (an attribute on the tensor object autograd hands from one Function's backward to the next, stamped with the tensor's
# version counter): nothing is keyed by address, nothing outlives the tensor, and a consumer that receives any other
# tensor (autograd summed two branches, a hook replaced the gradient, ...) finds no attribute and casts.
# ------------------------------------------------------------------------------------------------------
# ------------------------------------------------------------------------------------------------------

def _publish_twin(g32, g16, colsum=None):
    """``colsum`` (fp32 [C], optional): the column sums of g32 taken in fp32 by the kernel that produced it (LayerNorm backward's ``dcol``) -- the
    bias gradient of the linear whose output gradient g32 is (the consuming block's c_proj)"""
    g32._ocn_bf16_twin = (g16, g32._version)
    g32._ocn_colsum = (colsum, g32._version) if colsum is not None else None


def _take_colsum(g32):
    cs = getattr(g32, "_ocn_colsum", None)
    return cs[0] if (cs is not None and cs[1] == g32._version and cs[0].device == g32.device) else None


def _publish_f32(g16, g32, colsum=None):
    """bf16 residual stream (image tower): the OFFICIAL gradient between two blocks is the bf16 tensor (autograd hands a bf16 tensor's gradient on in
    bf16, and it is the dgrad / wgrad GEMMs' operand as it stands); ``image_stream="bf16-fp32grad"`` lets the fp32 value the LayerNorm backward had in
    registers travel beside it for the residual path -- the same attribute mechanism as the twin above, in the other direction"""
    g16._ocn_f32 = (g32, g16._version) if g32 is not None else None
    g16._ocn_colsum = (colsum, g16._version) if colsum is not None else None


def _take_f32(g16):
    c = getattr(g16, "_ocn_f32", None)
    return c[0] if (c is not None and c[1] == g16._version and c[0].shape == g16.shape and c[0].device == g16.device) else None


def _take_twin(g32, cache=None):
    tw = getattr(g32, "_ocn_bf16_twin", None)
    hit = tw is not None and tw[1] == g32._version and tw[0].shape == g32.shape and tw[0].device == g32.device
    if cache is not None:
        cache.twin_stats["hit" if hit else "miss"] += 1
    return tw[0] if hit else ops.cast_bf16(g32)


# ------------------------------------------------------------------------------------------------------
# weight-gradient GEMMs under the HBM-bound kernels of the backward.  In a block's backward the four wgrad GEMMs
# (MFMA-bound; 128 KiB of LDS and 2 waves per SIMD on every CU) feed nothing but .grad, while the chain that produces the
# next block's input gradient alternates NT GEMMs with HBM-bound kernels -- two LayerNorm backwards and the attention
# backward -- whose workgroups fit NEXT TO a wgrad workgroup on the same CU (18-25 KiB of LDS, one more wave per SIMD).
# So each HBM-bound kernel is paired with wgrad work on a second HIP stream:
#     main:  dGELU GEMM, dh2 GEMM | LN2 bwd     | da GEMM | attention bwd | dh1 GEMM | LN1 bwd
#     side:                       | wgrad proj  |         | wgrad fc      |          | wgrad out, wgrad qkv
# with a fork (side waits for main) before and a join (main waits for side) after every pairing: the persistent NT GEMMs
# never share the chip with a wgrad (two MFMA-bound kernels only take CUs from each other, and a persistent kernel whose
# workgroups start late ends late), and the block's gradients are complete on the main stream when they are handed to
# autograd (DDP hooks / the optimizer run there).  ``model.pair_wgrad = False`` keeps everything on one stream.
# ------------------------------------------------------------------------------------------------------
# The two towers are independent until the loss.  With ``tower_streams`` the IMAGE tower runs on a stream of its own (forward and,
# through autograd's per-node stream bookkeeping, backward) next to the text tower on the caller's stream: the tail of one tower's
# persistent GEMM (a last round that fills 37-43 % of the CUs at N = 768 / 512) and its HBM-bound kernels are filled by the other
# tower's launches.  197 -> 188 ms per step (profiles/r02_tower_streams.txt).  The image tower takes the side stream because autograd
# accumulates every parameter gradient on the stream the accumulator was created on -- under DDP the caller's stream -- behind a
# wait for the producing node: text-tower nodes (created last, run first in backward) sit on the caller's stream and wait for nothing,
# the waits for the image tower's blocks queue up behind them.  The per-block wgrad side streams (_Paired) are switched off in this
# mode (two MFMA-bound streams are enough; measured 191.5 with, 188.1 without).  The stream belongs to the model (``NativeCLIP._tower_side``).


class _StreamMap(dict):
    """device / (device, stream) -> torch.cuda.Stream, owned by a model or a tower cache; a deep copy of the owner (the EMA twin of
    base_task.py:171) starts with an empty map instead of trying to copy stream handles"""

    def __deepcopy__(self, memo):
        return _StreamMap()


class _AfterStream(torch.autograd.Function):
    """identity whose BACKWARD makes its own stream wait for ``other``.  ``tower_streams = "serial"`` (measurement mode of bench.py: every
    kernel alone on the chip, but each tower on the stream -- and therefore in the caching-allocator pool -- it uses when the towers
    overlap) puts it on the image features: the autograd engine runs the text tower's nodes first (created later = higher sequence
    numbers), all of them on the caller's stream, so when this node runs the whole text backward is enqueued there and the image
    tower's backward on the side stream starts behind it."""

    @staticmethod
    def forward(ctx, x, other):
        ctx.other = other
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        torch.cuda.current_stream(g.device).wait_stream(ctx.other)
        return g, None


class _Paired:
    """``with _Paired(dev, cache) as side: side(fn, ...)`` enqueues fn on the side stream after everything already on the main
    stream; leaving the block joins the side stream back into the main stream."""

    def __init__(self, dev, cache, enabled=True):
        self.side = cache.side_stream(dev) if enabled else None
        self.main = torch.cuda.current_stream(dev) if self.side is not None else None

    def __enter__(self):
        if self.side is not None:
            self.side.wait_stream(self.main)
        return self

    def __call__(self, fn, *args):
        if self.side is None:
            return fn(*args)
        with torch.cuda.stream(self.side):
            return fn(*args)

    def __exit__(self, *exc):
        if self.side is not None:
            self.main.wait_stream(self.side)
        return False


# ------------------------------------------------------------------------------------------------------
# residual block (transformer.py:319-330)
# ------------------------------------------------------------------------------------------------------
def _grad_arena(p):
    """one zeroed fp32 allocation carved into views shaped like the tensors of ``p`` (the wgrad / LayerNorm kernels accumulate into it)"""
    sizes = [q.numel() for q in p]
    arena = torch.zeros(sum(sizes), dtype=F32, device=p[0].device)
    grads, o = [], 0
    for q, n in zip(p, sizes):
        grads.append(arena[o:o + n].view(q.shape))
        o += n
    return grads


def _block_forward(x, p, cache, B, L, heads, causal, seq_off=None, act=ops.EPI_BIAS_GELU):
    (ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, wfc, bfc, wproj, bproj) = p
    M, C = x.shape
    h1, _, mean1, rstd1 = ops.layernorm_fwd(x, ln1w, ln1b)
    qkv = ops.gemm_nt(ops.EPI_BF16, h1, cache.get(wqkv, "n"), ops.empty((M, 3 * C), BF16, x), bias=bqkv)
    hd = C // heads
    a, lse = ops.attn_fwd(qkv, B, L, heads, causal, hd ** -0.5, hd, seq_off)
    # the residual stream keeps the dtype it arrives in: fp32, or bf16 (image tower, ``image_stream="bf16"``: LayerNorm reads bf16, the residual
    # epilogue adds in bf16 as the reference's autocast does -- ops.EPI_BIAS_RESID_BF16)
    epi_res = ops.EPI_BIAS_RESID_BF16 if x.dtype == BF16 else ops.EPI_BIAS_RESID_F32
    xmid = ops.gemm_nt(epi_res, a, cache.get(wo, "n"), ops.empty((M, C), x.dtype, x), bias=bo, resid=x)
    h2, _, mean2, rstd2 = ops.layernorm_fwd(xmid, ln2w, ln2b)
    Fd = wfc.shape[0]
    f = ops.empty((M, Fd), torch.uint8, x)  # gelu'(pre-activation) in 8-bit fixed point: all the backward needs of it
    g = ops.gemm_nt(act, h2, cache.get(wfc, "n"), ops.empty((M, Fd), BF16, x), bias=bfc, aux=f)  # act: erf GELU or QuickGELU epilogue
    y = ops.gemm_nt(epi_res, g, cache.get(wproj, "n"), ops.empty((M, C), x.dtype, x), bias=bproj, resid=xmid)
    return y, (mean1, rstd1, h1, qkv, a, lse, xmid, mean2, rstd2, h2, f, g)


class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, wfc, bfc, wproj, bproj, cache, B, L, heads, causal, recompute,
                seq_off=None, act=ops.EPI_BIAS_GELU):
        p = (ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, wfc, bfc, wproj, bproj)
        y, saved = _block_forward(x, p, cache, B, L, heads, causal, seq_off, act)
        ctx.meta = (cache, B, L, heads, causal, recompute, seq_off, act)
        if recompute:  # block-granular activation recompute (transformer.py:579-581): keep only the block input
            ctx.save_for_backward(x, *p)
        else:
            ctx.save_for_backward(x, *p, *saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        cache, B, L, heads, causal, recompute, seq_off, act = ctx.meta
        t = ctx.saved_tensors
        x, p = t[0], t[1:13]
        (ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, wfc, bfc, wproj, bproj) = p
        if recompute:
            _, saved = _block_forward(x, p, cache, B, L, heads, causal, seq_off, act)
        else:
            saved = t[13:]
        (mean1, rstd1, h1, qkv, a, lse, xmid, mean2, rstd2, h2, f, g) = saved
        M, C = x.shape
        Fd = wfc.shape[0]
        bf = x.dtype == BF16  # bf16 residual stream: dy IS the bf16 GEMM operand; the residual path adds it in bf16 (or from its fp32 companion)
        keep32 = bf and cache.stream == "bf16-fp32grad"
        dy_colsum = _take_colsum(dy)  # only the head publishes one (B rows: free): the LAST block's c_proj bias gradient as an fp32 column sum
        if bf:
            dy16 = dy.contiguous()
            dy = (_take_f32(dy) if keep32 else None)
            dy = dy16 if dy is None else dy
        else:
            dy16 = _take_twin(dy, cache)
            dy = dy.contiguous()
        # one zeroed fp32 arena for all of the block's parameter gradients (wgrad kernels accumulate atomically)
        grads = _grad_arena(p)
        (dln1w, dln1b, dwqkv, dbqkv, dwo, dbo, dln2w, dln2b, dwfc, dbfc, dwproj, dbproj) = grads
        # (Round 4 tried ALL c_proj / out_proj bias gradients as fp32 column sums taken inside the LayerNorm backward that produces dy / dxmid
        # (``dcol``): on all rows it costs that kernel 3-4 % (0.5 ms per step) and moves no digit of the parity report -- the bias gradients'
        # error at batch 4096 was a same-sign error of the summands, loss.py -- so the blocks keep the weight-gradient GEMM's own bias row.)
        if dy_colsum is not None:
            dbproj.copy_(dy_colsum)

        dev = x.device
        # a locked block (lock_image_tower / lock_text_tower with some groups left trainable above it) still has to pass the gradient
        # down to... nothing that trains: autograd only calls this backward when its INPUT needs a gradient, so the dgrad chain below is
        # always wanted; the four weight-gradient GEMMs are skipped when none of the block's parameters trains
        need_w = any(ctx.needs_input_grad[1:13])
        # ---- MLP branch: x_out = x_mid + c_proj(gelu(c_fc(ln_2(x_mid)))) ----
        df = ops.gemm_nt(ops.EPI_DGELU, dy16, cache.get(wproj, "t"), ops.empty((M, Fd), BF16, x), aux=f)
        dh2 = ops.gemm_nt(ops.EPI_BF16, df, cache.get(wfc, "t"), ops.empty((M, C), BF16, x))
        pair, det = cache.pair_wgrad, cache.deterministic
        with _Paired(dev, cache, pair) as side:
            if need_w:
                side(ops.gemm_tn_accum, dy16, g, dwproj, None if dy_colsum is not None else dbproj, 1.0, det)
            dxmid, dxmid16 = ops.layernorm_bwd(dh2, xmid, ln2w, mean2, rstd2, dln2w, dln2b, dres=dy, want_f32=(not bf or keep32), want_bf16=True, deterministic=cache.deterministic)
        # ---- attention branch: x_mid = x + out_proj(attn(in_proj(ln_1(x)))) ----
        da = ops.gemm_nt(ops.EPI_BF16, dxmid16, cache.get(wo, "t"), ops.empty((M, C), BF16, x))
        with _Paired(dev, cache, pair) as side:
            if need_w:
                side(ops.gemm_tn_accum, df, h2, dwfc, dbfc, 1.0, det)
            dqkv = ops.attn_bwd(qkv, a, da, lse, B, L, heads, causal, (C // heads) ** -0.5, C // heads, seq_off)
        dh1 = ops.gemm_nt(ops.EPI_BF16, dqkv, cache.get(wqkv, "t"), ops.empty((M, C), BF16, x))
        with _Paired(dev, cache, pair) as side:
            if need_w and det:
                side(ops.gemm_tn_accum, dxmid16, a, dwo, dbo, 1.0, True)
                side(ops.gemm_tn_accum, dqkv, h1, dwqkv, dbqkv, 1.0, True)
            elif need_w:  # out-proj and QKV wgrads share their rows and their K = C: one launch (36 tiles, 7 M-splits instead of 28 + 9)
                side(ops.gemm_tn_accum2, dxmid16, a, dwo, dbo, dqkv, h1, dwqkv, dbqkv)
            dx, dx16 = ops.layernorm_bwd(dh1, x, ln1w, mean1, rstd1, dln1w, dln1b, dres=(dxmid if dxmid is not None else dxmid16), want_f32=(not bf or keep32),
                                         want_bf16=True, deterministic=cache.deterministic)
        if bf:
            _publish_f32(dx16, dx)
            dx = dx16
        else:
            _publish_twin(dx, dx16)
        if not need_w:
            grads = [None] * 12
        return (dx, *grads, None, None, None, None, None, None, None, None)


# ------------------------------------------------------------------------------------------------------
# The LAST residual block of a tower, evaluated only where its output is consumed.
# Both poolers read ONE row per sequence of the last block's output -- `x[:, 0]` behind ln_post (transformer.py:829-831, 'tok') and
# `x[arange, text.argmax(-1)]` behind ln_final (:941-944) -- and a residual block is row-wise except inside the attention (where row
# i reads the K / V of the other rows).  So of the last block only  LN1 -> QKV -> attention  has to run on every row; the out-projection,
# both residual adds, LN2 and the whole MLP are needed on the B pooled rows alone (1 row in 50 / 43), forward and backward: the rows that
# are dropped reach neither the features nor any gradient.  Same results as the full block up to fp32 summation order
# (tests/test_model_gpu.py::test_pooled_last_block_equals_full_block); ``model.pooled_last_block = False`` (a constructor argument, too) runs
# the full block.  `rows` = absolute row of each sequence's pooled token (int32 [B]); the output is [B, C].
# ------------------------------------------------------------------------------------------------------
# Round 4: the attention itself is pruned the same way (head_dim 64; ``model.pooled_single_query = False`` keeps the round-3 form).  Only ONE
# query per sequence is ever read, so Q is projected for the B pooled rows alone (K and V for every row: in_proj_weight[C:]), the attention is
# the single-query kernel of csrc/attention_pooled.hip (2 L d flops per head instead of 4 L^2 d; dK / dV are rank-1 in every key row), the
# dgrad below it contracts over 2C instead of 3C, and the pooled rows' share of LayerNorm-1's backward (the query path + the residual) is
# evaluated on those B rows and enters the all-row LayerNorm backward as its residual-gradient input (LayerNorm's backward is linear in dy).
def _pooled_block_forward(x, p, rows, cache, B, L, heads, causal, seq_off=None, act=ops.EPI_BIAS_GELU):
    (ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, wfc, bfc, wproj, bproj) = p
    M, C = x.shape
    hd = C // heads
    single = cache.single_query and hd == 64
    h1, _, mean1, rstd1 = ops.layernorm_fwd(x, ln1w, ln1b)
    x_p = ops.gather_rows(x, rows, B, 0)  # [B, C]: from here on only the pooled rows
    if single:
        w_n = cache.get(wqkv, "n")  # [3C, C] bf16, rows [Wq; Wk; Wv] (transformer.py:93-95)
        kv = ops.gemm_nt(ops.EPI_BF16, h1, w_n[C:], ops.empty((M, 2 * C), BF16, x), bias=bqkv[C:])
        h1_p, _, mean1_p, rstd1_p = ops.layernorm_fwd(x_p, ln1w, ln1b)  # the pooled rows of h1 (row-wise op: the same values)
        q_p = ops.gemm_nt(ops.EPI_BF16, h1_p, w_n[:C], ops.empty((B, C), BF16, x), bias=bqkv[:C])
        a_p, lse = ops.attn_pooled_fwd(q_p, kv, rows, B, L, heads, causal, hd ** -0.5, seq_off)
        att = (kv, q_p, h1_p, mean1_p, rstd1_p)
    else:
        qkv = ops.gemm_nt(ops.EPI_BF16, h1, cache.get(wqkv, "n"), ops.empty((M, 3 * C), BF16, x), bias=bqkv)
        a, lse = ops.attn_fwd(qkv, B, L, heads, causal, hd ** -0.5, hd, seq_off)
        a_p = ops.gather_rows_bf16(a, rows, B, 0)
        att = (qkv, a)
    xmid_p = ops.gemm_nt(ops.EPI_BIAS_RESID_F32, a_p, cache.get(wo, "n"), ops.empty((B, C), F32, x), bias=bo, resid=x_p)
    h2_p, _, mean2, rstd2 = ops.layernorm_fwd(xmid_p, ln2w, ln2b)
    Fd = wfc.shape[0]
    f_p = ops.empty((B, Fd), torch.uint8, x)
    g_p = ops.gemm_nt(act, h2_p, cache.get(wfc, "n"), ops.empty((B, Fd), BF16, x), bias=bfc, aux=f_p)
    y_p = ops.gemm_nt(ops.EPI_BIAS_RESID_F32, g_p, cache.get(wproj, "n"), ops.empty((B, C), F32, x), bias=bproj, resid=xmid_p)
    return y_p, (mean1, rstd1, h1, lse, a_p, xmid_p, mean2, rstd2, h2_p, f_p, g_p, x_p, *att)


class _PooledBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, wfc, bfc, wproj, bproj, rows, cache, B, L, heads, causal, recompute,
                seq_off=None, act=ops.EPI_BIAS_GELU):
        p = (ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, wfc, bfc, wproj, bproj)
        y_p, saved = _pooled_block_forward(x, p, rows, cache, B, L, heads, causal, seq_off, act)
        ctx.meta = (cache, B, L, heads, causal, recompute, seq_off, act)
        if recompute:
            ctx.save_for_backward(x, *p, rows)
        else:
            ctx.save_for_backward(x, *p, rows, *saved)
        return y_p

    @staticmethod
    def backward(ctx, dy_p):
        cache, B, L, heads, causal, recompute, seq_off, act = ctx.meta
        t = ctx.saved_tensors
        x, p, rows = t[0], t[1:13], t[13]
        (ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, wfc, bfc, wproj, bproj) = p
        saved = _pooled_block_forward(x, p, rows, cache, B, L, heads, causal, seq_off, act)[1] if recompute else t[14:]
        (mean1, rstd1, h1, lse, a_p, xmid_p, mean2, rstd2, h2_p, f_p, g_p, x_p) = saved[:12]
        att = saved[12:]
        single = len(att) == 5  # (kv, q_p, h1_p, mean1_p, rstd1_p) of the single-query form, (qkv, a) otherwise
        M, C = x.shape
        Fd = wfc.shape[0]
        hd = C // heads
        dy16 = _take_twin(dy_p, cache)
        dy_colsum = _take_colsum(dy_p)
        dy_p = dy_p.contiguous()
        grads = _grad_arena(p)
        (dln1w, dln1b, dwqkv, dbqkv, dwo, dbo, dln2w, dln2b, dwfc, dbfc, dwproj, dbproj) = grads
        need_w = any(ctx.needs_input_grad[1:13])
        # ---- MLP branch and out-projection: the B pooled rows (the same arithmetic as the full block's, row for row) ----
        if dy_colsum is not None:
            dbproj.copy_(dy_colsum)
        else:
            ops.colsum_f32(dy_p, dbproj, cache.deterministic)
        df_p = ops.gemm_nt(ops.EPI_DGELU, dy16, cache.get(wproj, "t"), ops.empty((B, Fd), BF16, x), aux=f_p)
        dh2_p = ops.gemm_nt(ops.EPI_BF16, df_p, cache.get(wfc, "t"), ops.empty((B, C), BF16, x))
        dxmid_p, dxmid16_p = ops.layernorm_bwd(dh2_p, xmid_p, ln2w, mean2, rstd2, dln2w, dln2b, dres=dy_p, want_f32=True, want_bf16=True, deterministic=cache.deterministic)
        det = cache.deterministic
        if need_w:
            ops.gemm_tn_accum(dy16, g_p, dwproj, None, 1.0, det)
            ops.gemm_tn_accum(df_p, h2_p, dwfc, dbfc, 1.0, det)
            ops.gemm_tn_accum(dxmid16_p, a_p, dwo, dbo, 1.0, det)
        # what reaches x besides LayerNorm-1's all-row backward lives on the B pooled rows only: it is ADDED into those rows of dx behind that kernel
        # (ocn_scatter_add_rows) instead of travelling through a zero [M, C] fp32 residual-gradient matrix (1 GB of fills and reads per step)
        if single:
            kv, q_p, h1_p, mean1_p, rstd1_p = att
            w_t = cache.get(wqkv, "t")  # [C, 3C] bf16: columns [Wq^T | Wk^T | Wv^T]
            da_p = ops.gemm_nt(ops.EPI_BF16, dxmid16_p, cache.get(wo, "t"), ops.empty((B, C), BF16, x))
            dq_p, dkv = ops.attn_pooled_bwd(q_p, kv, a_p, da_p, lse, rows, B, L, heads, causal, hd ** -0.5, seq_off)
            dh1 = ops.gemm_nt(ops.EPI_BF16, dkv, w_t[:, C:], ops.empty((M, C), BF16, x))       # every row: through K and V
            dh1q_p = ops.gemm_nt(ops.EPI_F32, dq_p, w_t[:, :C], ops.empty((B, C), F32, x))     # the pooled rows: through their query
            if need_w:
                ops.gemm_tn_accum(dkv, h1, dwqkv[C:], dbqkv[C:], 1.0, det)
                ops.gemm_tn_accum(dq_p, h1_p, dwqkv[:C], dbqkv[:C], 1.0, det)
            # pooled rows: LayerNorm-1 backward of the query path + the residual x -> xmid
            dxp, _ = ops.layernorm_bwd(dh1q_p, x_p, ln1w, mean1_p, rstd1_p, dln1w, dln1b, dres=dxmid_p, want_f32=True, want_bf16=False, deterministic=cache.deterministic)
        else:
            qkv, a = att
            # ---- attention and everything below it: every row (the pooled rows' queries read all keys / values) ----
            da_p = ops.gemm_nt(ops.EPI_F32, dxmid16_p, cache.get(wo, "t"), ops.empty((B, C), F32, x))
            da = torch.zeros((M, C), dtype=BF16, device=x.device)
            ops.scatter_rows(da_p, rows, None, B, 0, da)
            dqkv = ops.attn_bwd(qkv, a, da, lse, B, L, heads, causal, hd ** -0.5, hd, seq_off)
            dh1 = ops.gemm_nt(ops.EPI_BF16, dqkv, cache.get(wqkv, "t"), ops.empty((M, C), BF16, x))
            if need_w:
                ops.gemm_tn_accum(dqkv, h1, dwqkv, dbqkv, 1.0, det)
            dxp = None
        bf = x.dtype == BF16  # bf16 residual stream below this block: its input gradient leaves as bf16 (see _BlockFn.backward)
        keep32 = bf and cache.stream == "bf16-fp32grad"
        if dxp is not None:
            dx, dx16 = ops.layernorm_bwd(dh1, x, ln1w, mean1, rstd1, dln1w, dln1b, want_f32=(not bf or keep32), want_bf16=True, deterministic=cache.deterministic)
            ops.scatter_add_rows(dxp, rows, dx, B, 0, dx16)
        else:
            # the all-query form keeps the full block's arithmetic to the bit (the residual gradient enters INSIDE the LayerNorm backward, where the
            # compiler contracts it into an FMA): it is the form tests/test_model_gpu.py::test_pooled_last_block_equals_full_block proves exact
            dres = torch.zeros((M, C), dtype=F32, device=x.device)
            ops.scatter_rows(dxmid_p, rows, dres, B, 0, None)  # the residual path x -> xmid carries gradient on the pooled rows only
            dx, dx16 = ops.layernorm_bwd(dh1, x, ln1w, mean1, rstd1, dln1w, dln1b, dres=dres, want_f32=(not bf or keep32), want_bf16=True, deterministic=cache.deterministic)
        if bf:
            _publish_f32(dx16, dx)
            dx = dx16
        else:
            _publish_twin(dx, dx16)
        if not need_w:
            grads = [None] * 12
        return (dx, *grads, None, None, None, None, None, None, None, None, None)


# ------------------------------------------------------------------------------------------------------
# image embedding (transformer.py:793-808)
# ------------------------------------------------------------------------------------------------------
class _VisionEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, conv_w, cls, pos, lnw, lnb, cache, patch, norm=None):
        width = conv_w.shape[0]
        KP = 3 * patch * patch
        Kpad = _round_up(KP, 64)
        if image.dtype == torch.uint8:  # decoded pixels: ToTensor + Normalize happen inside the patch kernel (8f-4)
            mean, std, hwc = norm
            B = image.shape[0]
            H, W = (image.shape[1], image.shape[2]) if hwc else (image.shape[2], image.shape[3])
            patches = ops.patchify_u8(image.contiguous(), patch, Kpad, mean, std, hwc)
        else:
            B, _, H, W = image.shape
            patches = ops.patchify(image.contiguous(), patch, Kpad)
        G = (H // patch) * (W // patch)
        if Kpad == KP:
            w16 = cache.get(conv_w, "n")
        else:  # zero-padded K (patch 14: 588 -> 640); rare path, weight-sized
            w16 = torch.zeros(width, Kpad, dtype=BF16, device=image.device)
            w16[:, :KP].copy_(cache.get(conv_w, "n"))
        po = ops.gemm_nt(ops.EPI_F32, patches, w16, ops.empty((B * G, width), F32, patches))
        emb = ops.embed_assemble_fwd(po, cls, pos, B, G, width)
        bf = cache.stream != "fp32"  # bf16 residual stream: ln_pre hands its result on in bf16 (layers.py:23-26 under autocast); emb itself stays fp32 here
        x16, x0, mean, rstd = ops.layernorm_fwd(emb, lnw, lnb, want_bf16=bf, want_f32=not bf)
        x0 = x16 if bf else x0
        ctx.save_for_backward(patches, emb, mean, rstd, lnw, conv_w, cls, pos)
        ctx.meta = (B, G, width, KP, Kpad)
        ctx.cache = cache
        return x0

    @staticmethod
    def backward(ctx, dx0):
        patches, emb, mean, rstd, lnw, conv_w, cls, pos = ctx.saved_tensors
        B, G, width, KP, Kpad = ctx.meta
        dev = emb.device
        dlnw, dlnb = torch.zeros_like(lnw), torch.zeros_like(lnw)
        dy0 = (_take_f32(dx0) if dx0.dtype == BF16 else None)  # bf16 stream: the first block's fp32 companion when there is one
        dy0 = dx0.contiguous() if dy0 is None else dy0
        demb, _ = ops.layernorm_bwd(dy0, emb, lnw, mean, rstd, dlnw, dlnb, want_f32=True, deterministic=ctx.cache.deterministic)
        dpos, dcls = torch.zeros_like(pos), torch.zeros_like(cls)
        dpatch = ops.embed_assemble_bwd(demb, dpos, dcls, B, G, width, ctx.cache.deterministic)
        dw = torch.zeros(width, Kpad, dtype=F32, device=dev)
        ops.gemm_tn_accum(dpatch, patches, dw, None, 1.0, ctx.cache.deterministic)
        dconv = (dw if Kpad == KP else dw[:, :KP].contiguous()).view(conv_w.shape)
        return None, dconv, dcls, dpos, dlnw, dlnb, None, None, None


# ------------------------------------------------------------------------------------------------------
# text embedding (model.py:399-401)
# ------------------------------------------------------------------------------------------------------
class _TextEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, text, table, pos, pack=None, deterministic=False):
        if pack is None:
            x = ops.token_embed_fwd(text.contiguous(), table, pos)
        else:
            x = ops.token_embed_fwd_rows(pack.tokens, pack.posidx, table, pos)
        ctx.pack = pack
        ctx.det = bool(deterministic)
        ctx.save_for_backward(text, table, pos)
        return x

    @staticmethod
    def backward(ctx, dx):
        text, table, pos = ctx.saved_tensors
        dtable, dpos = torch.zeros_like(table), torch.zeros_like(pos)
        dxv = dx.contiguous()
        if ctx.pack is None:
            ops.token_embed_bwd_sorted(text.contiguous(), dxv, dtable, dpos, ctx.det)
        else:
            B, L = text.shape
            ops.token_embed_bwd_sorted_varlen(ctx.pack.tokens, ctx.pack.seq_off, B, L, dxv, dtable, dpos, ctx.det)
        return None, dtable, dpos, None, None


class _TextPack:
    """Packed layout of one text batch (``ocn_seq_pack_plan``): only the first ``argmax(text[b]) + 1`` tokens of each sequence exist
    in the text tower.  The reference pools ``x[b, text[b].argmax()]`` (transformer.py:941-944) behind a causal mask (:1716-1722),
    so the positions after the pooled one influence neither the feature nor any gradient -- dropping them changes no result and
    removes their share of every GEMM / LayerNorm / attention launch of the tower (tokenizer output is ~60 % padding on LAION-like
    captions).  The constructor enqueues the planning kernels and the few-byte device -> host copy of the packed row count; ``finish()``
    waits for that copy (the only host synchronisation of the step; it is issued BEFORE the image tower so that it never drains
    the queue) and builds the packed token / position lists.  A batch that arrives through ``DeviceBatchPipeline(plan_text_vocab=...)``
    carries the whole layout, computed on the host next to the tokenizer: then nothing is planned or read back here (``host_planned``)
    and the training step runs without a single host synchronisation."""

    def __init__(self, text, vocab_size=None, buckets=True):
        host = getattr(text, "_ocn_host_plan", None)
        self.text = text = text.contiguous()
        self.B, self.L = text.shape
        self.vocab_size = vocab_size
        self.host_planned = host is not None
        if host is not None:
            # the layout came with the batch (input_pipeline.HostTextPlan: computed on the host, where the tokens were produced): no kernel,
            # no read-back, no host synchronisation anywhere in the step
            if vocab_size is not None and host["n_bad"]:
                raise IndexError(f"index out of range in self: {host['n_bad']} token id(s) outside [0, {vocab_size}) (token_embedding has {vocab_size} rows)")
            self.eot, self.seq_off, self.last_row, self.order = host["eot"], host["seq_off"], host["last_row"], host["order"]
            self.M, self.tokens, self.posidx = host["M"], host["tokens"], host["posidx"]
            self.layout = ops.SeqLayout(self.seq_off, self.order, host["counts"]) if (buckets and self.order is not None) else self.seq_off
            return
        # with the vocabulary size the plan also counts ids outside [0, vocab): the embedding kernels clamp them (they can never read
        # outside the table), nn.Embedding raises (model.py:399) -- so does finish(), from the same 8-byte read-back
        self.eot, plan, self.last_row, self.order = ops.seq_pack_plan(text, vocab_size, buckets=buckets)
        self.seq_off = self.layout = plan[:self.B + 1]
        self._m_host = torch.empty(plan.numel() - self.B, dtype=torch.int32).pin_memory()  # [M, ids out of range, bucket counts ...]
        self._m_host.copy_(plan[self.B:], non_blocking=True)
        self._ready = torch.cuda.Event()
        self._ready.record()
        self.M = None

    def finish(self):
        if self.M is None:
            self._ready.synchronize()
            if self.vocab_size is not None and int(self._m_host[1]) != 0:
                raise IndexError(f"index out of range in self: {int(self._m_host[1])} token id(s) outside [0, {self.vocab_size}) "
                                 f"(token_embedding has {self.vocab_size} rows)")
            self.M = int(self._m_host[0])
            if self.order is not None:  # attention launches in buckets of equal block count (ops.SeqLayout)
                self.layout = ops.SeqLayout(self.seq_off, self.order, self._m_host[2:].tolist())
            self.tokens, self.posidx = ops.seq_pack_rows(self.text, self.seq_off, self.M)
        return self


# ------------------------------------------------------------------------------------------------------
# pooled head: LN on the pooled token only (== LN on all tokens then pool), projection, F.normalize
#   vision: transformer.py:829-831 + :923 ; text: model.py:403-411 + transformer.py:941-944
# ------------------------------------------------------------------------------------------------------
class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lnw, lnb, proj, idx, cache, B, L, normalize):
        pooled = ops.gather_rows(x, idx, B, L)
        p16, _, mean, rstd = ops.layernorm_fwd(pooled, lnw, lnb)
        E = proj.shape[1]
        feat = ops.gemm_nt(ops.EPI_F32, p16, cache.get(proj, "t"), ops.empty((B, E), F32, x))
        if normalize:
            y, _, inv = ops.l2norm_fwd(feat)
        else:
            y, inv = feat, None
        ctx.save_for_backward(pooled, p16, mean, rstd, lnw, proj, idx, y, inv)
        ctx.meta = (cache, B, L, normalize, x.shape, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        pooled, p16, mean, rstd, lnw, proj, idx, y, inv = ctx.saved_tensors
        cache, B, L, normalize, xshape, xdtype = ctx.meta
        dy = dy.contiguous().float()
        dfeat = ops.l2norm_bwd(dy, y, inv) if normalize else dy
        dfeat16 = ops.cast_bf16(dfeat)
        C, E = proj.shape
        # fp32 into the LayerNorm backward (B rows: free): ln_post / ln_final's bias gradient is a column sum over the B pooled rows
        dp32 = ops.gemm_nt(ops.EPI_F32, dfeat16, cache.get(proj, "n"), ops.empty((B, C), F32, dy))
        dproj = torch.zeros_like(proj)
        ops.gemm_tn_accum(p16, dfeat16, dproj, None, 1.0, cache.deterministic)
        dlnw, dlnb = torch.zeros_like(lnw), torch.zeros_like(lnw)
        dcol = torch.zeros_like(lnw)  # column sums of dpooled = of dx (zero elsewhere): the last block's c_proj bias gradient, in fp32
        dpooled, _ = ops.layernorm_bwd(dp32, pooled, lnw, mean, rstd, dlnw, dlnb, want_f32=True, dcol=dcol, deterministic=cache.deterministic)
        dx16 = torch.zeros(xshape, dtype=BF16, device=dy.device)  # bf16 twin for the last block's dgrad / wgrad GEMMs
        if xdtype == BF16:  # all rows of a bf16 residual stream (full last block): the gradient itself is the bf16 tensor
            ops.scatter_rows(dpooled, idx, None, B, L, dx16)
            _publish_f32(dx16, None, dcol)
            return dx16, dlnw, dlnb, dproj, None, None, None, None, None
        dx = torch.zeros(xshape, dtype=F32, device=dy.device)
        ops.scatter_rows(dpooled, idx, dx, B, L, dx16)
        _publish_twin(dx, dx16, dcol)
        return dx, dlnw, dlnb, dproj, None, None, None, None, None


# ------------------------------------------------------------------------------------------------------
# logits of CLIP.get_logits (model.py:413-420) as an autograd node over the library's GEMMs
# ------------------------------------------------------------------------------------------------------
class _LogitsFn(torch.autograd.Function):
    """li = exp(logit_scale) * I @ T^T (+ logit_bias), fp32 [B, N].  Backward with G = dL/dli: dI = s * G @ T, dT = G^T @ (s I), d logit_scale = sum(G * (li - bias))
    (the parameter is the LOG scale: d li / d logit_scale = li - bias), d logit_bias = sum(G) -- the two products as bf16-operand GEMMs with fp32
    accumulation, like the loss modules' (open_clip_amd/loss.py); the two scalars are fp32 reductions."""

    @staticmethod
    def forward(ctx, i, t, logit_scale, logit_bias):
        i32, t32 = i.detach().float().contiguous(), t.detach().float().contiguous()
        E, N = i32.shape[1], t32.shape[0]
        Ep = _round_up(E, 64)
        s = logit_scale.detach().exp().reshape(1).float()
        if Ep == E:
            i16, t16 = ops.cast_bf16_scaled(i32, s), ops.cast_bf16(t32)
        else:  # embed dims that are not a multiple of the GEMM's K step: zero-padded operands (no registered config needs it)
            i16, t16 = torch.zeros(i32.shape[0], Ep, dtype=BF16, device=i32.device), torch.zeros(N, Ep, dtype=BF16, device=i32.device)
            i16[:, :E].copy_(i32 * s)
            t16[:, :E].copy_(t32)
        bias = None if logit_bias is None else logit_bias.detach().reshape(1).float().expand(N).contiguous()
        li = torch.empty(i32.shape[0], _round_up(N, 4), dtype=F32, device=i32.device)[:, :N]
        ops.gemm_nt(ops.EPI_F32, i16, t16, li, bias=bias)
        ctx.save_for_backward(i16, t16, li, s, logit_bias)
        ctx.meta = (E, i.dtype, t.dtype)
        return li

    @staticmethod
    def backward(ctx, g):
        i16, t16, li, s, logit_bias = ctx.saved_tensors
        E, idt, tdt = ctx.meta
        B, N = li.shape
        g = g.float()
        Np = _round_up(N, 64)
        g16 = torch.zeros(B, Np, dtype=BF16, device=g.device)  # K of the dI product = N, padded to the GEMM's K step
        g16[:, :N].copy_(g)
        tT = torch.zeros(t16.shape[1], Np, dtype=BF16, device=g.device)
        tT[:, :N].copy_(t16.t())
        di = ops.gemm_nt(ops.EPI_F32, g16, tT, torch.empty(B, t16.shape[1], dtype=F32, device=g.device))[:, :E] * s  # s * G @ T
        dt = torch.zeros(Np, i16.shape[1], dtype=F32, device=g.device)
        ops.gemm_tn_accum(g16, i16, dt)  # G^T @ (s I): i16 already carries the scale
        dbias = g.sum() if logit_bias is not None else None
        lin = li if logit_bias is None else li - logit_bias.detach().float()
        dscale = (g * lin).sum()
        return di.to(idt), dt[:N, :E].to(tdt), dscale.reshape(()).to(s.dtype), (None if dbias is None else dbias.reshape(logit_bias.shape).to(logit_bias.dtype))


# ------------------------------------------------------------------------------------------------------
# parameter containers with the reference's names
# ------------------------------------------------------------------------------------------------------
class _Params(nn.Module):
    """Holds parameters only; calling it is a bug (the arithmetic lives in the autograd Functions)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} is a parameter container of the native path; it has no forward")


class LayerNorm(_Params):  # layers.py:20-26
    def __init__(self, width):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(width))
        self.bias = nn.Parameter(torch.zeros(width))
        self.eps = 1e-5
        self.normalized_shape = (width,)


class Linear(_Params):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        bound = 1 / math.sqrt(in_features)  # nn.Linear's default (kaiming_uniform_(a=sqrt(5)) == U(+-1/sqrt(in)) for weight and bias)
        self.weight = nn.Parameter(torch.empty(out_features, in_features).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound))


class Attention(_Params):  # transformer.py:61-155 (fused in_proj, out_proj)
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads, self.head_dim, self.scale = num_heads, dim // num_heads, (dim // num_heads) ** -0.5
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        nn.init.xavier_uniform_(self.in_proj_weight)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = Linear(dim, dim)
        nn.init.zeros_(self.out_proj.bias)  # transformer.py:145-155 (_reset_parameters: "match nn.MultiheadAttention init")


class Mlp(_Params):  # transformer.py:295-299 (OrderedDict c_fc / gelu / c_proj)
    def __init__(self, width, mlp_width):
        super().__init__()
        self.c_fc = Linear(width, mlp_width)
        self.c_proj = Linear(mlp_width, width)


class ResidualAttentionBlock(nn.Module):  # transformer.py:274-330
    def __init__(self, d_model, n_head, mlp_ratio=4.0, quick_gelu=False):
        super().__init__()
        self.ln_1 = LayerNorm(d_model)
        self.attn = Attention(d_model, n_head)
        self.ln_2 = LayerNorm(d_model)
        self.mlp = Mlp(d_model, int(d_model * mlp_ratio))
        self.n_head = n_head
        # act_layer of the MLP (transformer.py:295-299): nn.GELU (erf), or QuickGELU for the `quick_gelu` configs (model.py:172, layers.py:29-32)
        self.act_epilogue = ops.EPI_BIAS_QUICKGELU if quick_gelu else ops.EPI_BIAS_GELU

    def params(self):
        return (self.ln_1.weight, self.ln_1.bias, self.attn.in_proj_weight, self.attn.in_proj_bias,
                self.attn.out_proj.weight, self.attn.out_proj.bias, self.ln_2.weight, self.ln_2.bias,
                self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight, self.mlp.c_proj.bias)

    def get_weight_dtype(self):
        return self.mlp.c_fc.weight.dtype

    def forward(self, x, cache, B, L, causal, recompute=False, seq_off=None, pooled_rows=None):
        """``pooled_rows`` (int32 [B], absolute rows): the caller reads only these rows of the output -- everything behind the attention
        then runs on them alone and the result is [B, C] (_PooledBlockFn).  Both forms go through ``Module.__call__``, so forward /
        forward-pre hooks on the block (FSDP2's unshard, feature extraction, profilers) fire either way."""
        if pooled_rows is not None:
            return _PooledBlockFn.apply(x, *self.params(), pooled_rows, cache, B, L, self.n_head, causal, recompute, seq_off, self.act_epilogue)
        return _BlockFn.apply(x, *self.params(), cache, B, L, self.n_head, causal, recompute, seq_off, self.act_epilogue)


class Transformer(nn.Module):  # transformer.py:476-585
    def __init__(self, width, layers, heads, mlp_ratio=4.0, quick_gelu=False):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads, mlp_ratio, quick_gelu) for _ in range(layers)])
        self.grad_checkpointing = False
        self.keep_last_blocks = 0

    def get_cast_dtype(self):  # transformer.py:537-538
        return self.resblocks[0].get_weight_dtype()

    def set_grad_checkpointing(self, enable=True, impl="inline", keep_last=0):
        """transformer.py:577-585 recomputes EVERY block in the backward.  ``keep_last`` = n (native extension) keeps the activations of
        the last n blocks -- the ones whose backward runs first, so their memory is free again when the recompute of the earlier
        blocks needs its transient -- and recomputes only the others: same results, and 288 GB of HBM hold far more than one block
        input per layer (``NativeCLIP.plan_grad_checkpointing`` sizes n for a batch)."""
        self.grad_checkpointing = enable
        self.keep_last_blocks = max(0, int(keep_last))

    def forward(self, x, cache, B, L, causal, seq_off=None, pooled_rows=None):
        """``pooled_rows`` (int32 [B], absolute rows): the caller only reads these rows of the output -- the last block then runs as
        _PooledBlockFn and the result is [B, C] (the pooled rows, in order) instead of [M, C]"""
        rc = self.grad_checkpointing and torch.is_grad_enabled()
        blocks = list(self.resblocks)
        first_kept = len(blocks) - (self.keep_last_blocks if rc else 0)  # blocks from here on keep their activations
        last = blocks.pop() if pooled_rows is not None else None
        for i, r in enumerate(blocks):
            x = r(x, cache, B, L, causal, rc and i < first_kept, seq_off)
        if last is not None:
            x = last(x, cache, B, L, causal, rc and len(blocks) < first_kept, seq_off, pooled_rows)
        return x


def _pooled_last_block_ok(module) -> bool:
    """the pooled form of the last block (see _PooledBlockFn) unless switched off on the module"""
    return bool(getattr(module, "pooled_last_block", True))


def _set_group_requires_grad(members, requires_grad: bool):  # transformer.py:2034-2041
    for m in members:
        if isinstance(m, nn.Parameter):
            m.requires_grad = requires_grad
        else:
            for p in m.parameters():
                p.requires_grad = requires_grad


def _lock_layer_groups(groups, unlocked: int = 0):
    """transformer.py:2044-2053: freeze bottom-up, the top ``unlocked`` groups (projection head first) stay trainable; every group is
    set explicitly so repeated calls with different counts are idempotent"""
    n_freeze = len(groups) if not unlocked else len(groups) - unlocked
    for i, (_, members) in enumerate(groups):
        _set_group_requires_grad(members, requires_grad=(i >= n_freeze))


class _Conv1(_Params):
    def __init__(self, width, patch):
        super().__init__()
        bound = 1 / math.sqrt(3 * patch * patch)
        self.weight = nn.Parameter(torch.empty(width, 3, patch, patch).uniform_(-bound, bound))


class _Embedding(_Params):
    def __init__(self, vocab, width):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(vocab, width).normal_(std=0.02))
        self.num_embeddings, self.embedding_dim = vocab, width


class VisionTransformer(nn.Module):  # transformer.py:592-928 (default path: learnable pos, 'tok' pool, no patch dropout)
    def __init__(self, image_size, patch_size, width, layers, heads, mlp_ratio, output_dim, quick_gelu=False):
        super().__init__()
        self.image_size = (image_size, image_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (image_size // patch_size, image_size // patch_size)
        self.pooled_last_block = True  # the last block only on the class-token rows behind the attention (see _PooledBlockFn)
        self.output_dim = output_dim
        self.width = width
        scale = width ** -0.5
        self.conv1 = _Conv1(width, patch_size)
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid_size[0] * self.grid_size[1] + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, mlp_ratio, quick_gelu)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self._cache = _WeightCache()
        self.image_mean = (0.48145466, 0.4578275, 0.40821073)  # OPENAI_DATASET_MEAN / _STD (constants.py:1-2); used by the
        self.image_std = (0.26862954, 0.26130258, 0.27577711)  # uint8 input path only
        # what set_model_preprocess_cfg (model.py:874-878) leaves on the tower: transform.py:18-25 PreprocessCfg as a dict
        self.preprocess_cfg = {"size": (image_size, image_size), "mode": "RGB", "mean": self.image_mean, "std": self.image_std,
                               "interpolation": "bicubic", "resize_mode": "shortest", "fill_color": 0}

    def set_grad_checkpointing(self, enable=True, impl="inline", keep_last=0):
        self.transformer.set_grad_checkpointing(enable, impl, keep_last)

    def no_weight_decay(self):  # transformer.py:745-751
        return {"positional_embedding", "class_embedding"}

    def layer_groups(self, pooler_in_head: bool = True):
        """transformer.py:718-743: ordered, complete partition input -> output shared by ``lock`` and layer-wise LR decay:
        ``embeddings`` (patch conv + class / positional embeddings + ln_pre), ``layer.{i}`` (the last block together with
        ln_post), ``proj``"""
        groups = [("embeddings", [self.conv1, self.class_embedding, self.positional_embedding, self.ln_pre])]
        n = len(self.transformer.resblocks)
        for i, block in enumerate(self.transformer.resblocks):
            groups.append((f"layer.{i}", [block] + ([self.ln_post] if i == n - 1 else [])))
        groups.append(("proj", [self.proj]))
        return groups

    def lock(self, unlocked_groups=0, freeze_bn_stats=False):  # transformer.py:745-753
        _lock_layer_groups(self.layer_groups(), unlocked_groups)

    def forward(self, image, normalize=False):
        """``image``: float [B,3,H,W] (already normalised, as the reference's transform produces) or uint8 pixels
        ([B,3,H,W], or [B,H,W,3] as decoders emit them) which are normalised with ``image_mean`` / ``image_std`` in the
        patch kernel."""
        B = image.shape[0]
        norm = None
        if image.dtype == torch.uint8:
            hwc = image.shape[-1] == 3 and image.shape[1] != 3
            hw = tuple(image.shape[1:3]) if hwc else tuple(image.shape[2:])
            norm = (self.image_mean, self.image_std, hwc)
        else:
            hw = tuple(image.shape[2:])
        if hw != tuple(self.image_size):
            raise RuntimeError(f"image size {hw} != {self.image_size}")
        T = self.grid_size[0] * self.grid_size[1] + 1
        x = _VisionEmbedFn.apply(image, self.conv1.weight, self.class_embedding, self.positional_embedding,
                                 self.ln_pre.weight, self.ln_pre.bias, self._cache, self.patch_size[0], norm)
        if _pooled_last_block_ok(self):
            rows = torch.arange(B, device=x.device, dtype=torch.int32)
            x = self.transformer(x, self._cache, B, T, False, None, rows * T)  # the class token's rows; the result is [B, C]
            return _HeadFn.apply(x, self.ln_post.weight, self.ln_post.bias, self.proj, rows, self._cache, B, 0, normalize)
        x = self.transformer(x, self._cache, B, T, False)
        return _HeadFn.apply(x, self.ln_post.weight, self.ln_post.bias, self.proj, None, self._cache, B, T, normalize)


ATTENTION_HEAD_DIMS = (64, 80, 88, 96, 104, 112, 128)  # instantiations of csrc/attention_generic.hip (64 also: csrc/attention.hip)


class NativeCLIP(nn.Module):
    """Drop-in for ``open_clip.model.CLIP`` (model.py:318-548) on the ViT + causal-text path."""

    # options of the reference's CLIPVisionCfg / CLIPTextCfg / CLIP.__init__ (model.py:27-131, :318-365) that the native path implements,
    # with the only value it implements for the others (the reference dataclass default): anything else must fail loudly instead of
    # training a silently different model (e.g. a *-quickgelu config or a SigLIP-style pooling registered through add_model_config)
    _VISION_KEYS = {"layers", "width", "head_width", "mlp_ratio", "patch_size", "image_size"}
    _TEXT_KEYS_OK = {"context_length", "vocab_size", "width", "heads", "layers", "mlp_ratio"}
    _VISION_DEFAULTS = {"ls_init_value": None, "patch_dropout": 0.0, "attentional_pool": False, "pos_embed_type": "learnable", "no_ln_pre": False,
                        "pool_type": "tok", "final_ln_after_pool": False, "output_tokens": False, "act_kwargs": None, "norm_kwargs": None,
                        "block_type": None, "qk_norm": False, "scaled_cosine_attn": False, "scale_heads": False, "scale_attn_inner": False,
                        "scale_attn": False, "scale_fc": False, "timm_model_name": None, "in_chans": 3}
    _TEXT_DEFAULTS = {"hf_model_name": None, "hf_tokenizer_name": None, "tokenizer_kwargs": None, "tokenizer_mode": None, "ls_init_value": None,
                      "embed_cls": False, "pad_id": 0, "eos_id": None, "no_causal_mask": False, "final_ln_after_pool": False, "pool_type": "argmax",
                      "proj_bias": False, "proj_type": "linear", "output_tokens": False, "act_kwargs": None, "norm_kwargs": None, "block_type": None,
                      "qk_norm": False, "scaled_cosine_attn": False, "scale_heads": False, "scale_attn_inner": False, "scale_attn": False,
                      "scale_fc": False, "mlp_type": "mlp", "hf_proj_type": None, "hf_pooler_type": None}
    _MODEL_DEFAULTS = {"cast_dtype": None, "nonscalar_logit_scale": False, "custom_text": False,
                       "multimodal_cfg": None}

    @classmethod
    def _check_cfg(cls, vision_cfg, text_cfg, model_kwargs):
        for name, cfg, ok, defaults in (("vision_cfg", vision_cfg, cls._VISION_KEYS, cls._VISION_DEFAULTS), ("text_cfg", text_cfg, cls._TEXT_KEYS_OK, cls._TEXT_DEFAULTS),
                                        ("model", model_kwargs, set(), cls._MODEL_DEFAULTS)):
            for k, val in cfg.items():
                if k in ok:
                    continue
                if k in defaults and (val == defaults[k] or (val is None and not defaults[k])):
                    continue
                raise NotImplementedError(f"NativeCLIP: {name}[{k!r}] = {val!r} is not implemented by the native path "
                                          f"(supported: {sorted(ok)}; everything else only at the reference default)")

    def __init__(self, embed_dim, vision_cfg, text_cfg, init_logit_scale=math.log(1 / 0.07), init_logit_bias=None,
                 output_dict=False, *, pack_text=True, tower_streams=True, pooled_last_block=True, attn_buckets=True, pair_wgrad=True,
                 deterministic=False, quick_gelu=False, image_stream="fp32", **model_kwargs):
        """Reference arguments first (model.py:318-365).  Keyword-only switches of the native execution (every one also a plain attribute
        that may be flipped later; none changes a result beyond fp32 summation order):
        ``pack_text`` -- the text tower holds only the tokens up to the pooled EOT (_TextPack); ``tower_streams`` -- image tower on a
        stream of its own next to the text tower; ``pooled_last_block`` -- the last block of each tower behind its attention on the
        pooled rows only (_PooledBlockFn; ``model.visual.pooled_last_block`` is the image tower's switch); ``attn_buckets`` -- packed
        attention launches grouped by 32-row block count; ``pair_wgrad`` -- in one-stream mode, the blocks' wgrad GEMMs on a side
        stream under the HBM-bound kernels; ``deterministic`` -- every weight / bias gradient GEMM in its reproducible form (per-split
        slabs summed in a fixed order instead of fp32 atomics: 98 % of the gradient elements; the LayerNorm-affine, embedding and
        scalar-loss reductions still accumulate with fp32 atomics).
        ``image_stream`` DOES change results, inside the stated tolerances: the dtype of the IMAGE tower's residual stream -- "fp32" (default; stricter than
        the reference), "bf16" (stream and its gradient in bf16 = what the reference's autocast produces there, transformer.py:794 + layers.py:23-26;
        half the residual-stream bytes of 24 residual epilogues and 49 LayerNorm passes) or "bf16-fp32grad" (bf16 stream, fp32 residual-gradient path)."""
        super().__init__()
        if image_stream not in ("fp32", "bf16", "bf16-fp32grad"):
            raise ValueError(f"image_stream must be 'fp32', 'bf16' or 'bf16-fp32grad' (got {image_stream!r})")
        v, t = dict(vision_cfg), dict(text_cfg)
        self._check_cfg(v, t, model_kwargs)
        self.output_dict = output_dict
        self.embed_dim = embed_dim
        head_width = v.get("head_width", 64)
        for hd in (head_width, t["width"] // t["heads"]):
            if hd not in ATTENTION_HEAD_DIMS:
                raise NotImplementedError(f"the native attention kernels support head_dim {' / '.join(map(str, ATTENTION_HEAD_DIMS))} (got {hd})")
        # `quick_gelu` (model.py:172-176, :262): QuickGELU instead of nn.GELU in the MLPs of BOTH towers (ViT-B-32-quickgelu.json: the
        # OpenAI / LAION-400M checkpoints)
        self.quick_gelu = bool(quick_gelu)
        self.visual = VisionTransformer(v["image_size"], v["patch_size"], v["width"], v["layers"], v["width"] // head_width,
                                        v.get("mlp_ratio", 4.0), embed_dim, self.quick_gelu)
        tw = t["width"]
        self.transformer = Transformer(tw, t["layers"], t["heads"], t.get("mlp_ratio", 4.0), self.quick_gelu)
        self.context_length = t["context_length"]
        self.vocab_size = t["vocab_size"]
        self.token_embedding = _Embedding(self.vocab_size, tw)
        self.positional_embedding = nn.Parameter(torch.empty(self.context_length, tw).normal_(std=0.01))
        self.ln_final = LayerNorm(tw)
        self.text_projection = nn.Parameter(torch.empty(tw, embed_dim).normal_(std=tw ** -0.5))
        self.text_pool_type = "argmax"
        self.text_eos_id = None
        mask = torch.full((self.context_length, self.context_length), float("-inf")).triu_(1)
        self.register_buffer("attn_mask", mask, persistent=False)  # model.py:360 (kept for API parity; kernels use a predicate)
        self.logit_scale = nn.Parameter(torch.ones([]) * init_logit_scale)
        self.logit_bias = nn.Parameter(torch.ones([]) * init_logit_bias) if init_logit_bias is not None else None
        self._cache = _WeightCache()
        # packed text tower (see _TextPack): on by default where the varlen attention kernels apply (head_dim 64, L <= 320);
        # ``pack_text = False`` runs every one of the context_length positions like the reference does
        self.pack_text = bool(pack_text) and t["width"] // t["heads"] == 64 and self.context_length <= 320
        self.attn_buckets = bool(attn_buckets)
        self._tower_side = _StreamMap()  # device -> the image tower's stream (created on first use)
        # True: image tower on a stream of its own next to the text tower (see _AfterStream / forward); False: one stream; "serial": the same two
        # streams, one tower at a time (bench.py's event-timed steps)
        self.tower_streams = tower_streams
        # the last block of each tower only where its output is read (see _PooledBlockFn); the image tower has its own switch
        # (``model.visual.pooled_last_block``)
        self.pooled_last_block = self.visual.pooled_last_block = bool(pooled_last_block)
        # the pooled last block's attention as ONE query per sequence (K, V projection only; head_dim 64): see _pooled_block_forward
        self.pooled_single_query = True
        self.deterministic = bool(deterministic)
        self.pair_wgrad = bool(pair_wgrad)  # one-stream mode: the blocks' wgrad GEMMs on a side stream under the HBM-bound kernels (_Paired)
        self.image_stream = image_stream  # residual-stream dtype of the image tower (see the docstring); a plain attribute like the other switches
        self.init_parameters()
        # the bf16 operand copies are keyed by (address, version counter); writes through ``.data`` (checkpoint loading, EMA swaps,
        # manual re-initialisation) do not move the counter, so every load_state_dict drops them
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_weight_caches())

    def invalidate_weight_caches(self):
        """drop the cached bf16 operand copies of the GEMM weights: call after writing to parameters through ``.data`` (anything that
        does not bump ``Tensor._version``); load_state_dict does it by itself"""
        self._cache.clear()
        self.visual._cache.clear()

    def init_parameters(self):
        """the reference's from-scratch initialisation, drawn from the global RNG (``torch.manual_seed``): text tower
        transformer.py:1664-1685 (normal inits scaled by width / depth), image tower :641-645, :714 + module defaults
        (xavier_uniform in_proj, zero attention biases :145-155, nn.Linear defaults for the MLP), logit_scale model.py:326"""
        tw, tl = self.transformer.width, self.transformer.layers
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        proj_std, attn_std, fc_std = (tw ** -0.5) * ((2 * tl) ** -0.5), tw ** -0.5, (2 * tw) ** -0.5
        for block in self.transformer.resblocks:
            nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=tw ** -0.5)

    def text_layer_groups(self, pooler_in_head: bool = True):
        """transformer.py:1999-2031 (``_text_layer_groups`` on a CLIP that unpacks the text tower onto itself): ``embeddings``,
        ``layer.{i}`` (the last block with ln_final), ``proj``"""
        groups = [("embeddings", [self.token_embedding, self.positional_embedding])]
        n = len(self.transformer.resblocks)
        for i, block in enumerate(self.transformer.resblocks):
            groups.append((f"layer.{i}", [block] + ([self.ln_final] if i == n - 1 else [])))
        groups.append(("proj", [self.text_projection]))
        return groups

    def fsdp_shard_modules(self):
        """base_task.py:234-250: the units FSDP2 shards -- every residual block of both towers (the task's default discovery looks
        for the reference's block classes, which the native containers are not)"""
        return [(n, m) for n, m in self.named_modules() if isinstance(m, ResidualAttentionBlock)]

    # ---- reference API surface (model.py:369-411) ----
    def lock_image_tower(self, unlocked_groups=0, freeze_bn_stats=False):
        self.visual.lock(unlocked_groups=unlocked_groups, freeze_bn_stats=freeze_bn_stats)

    def lock_text_tower(self, unlocked_layers: int = 0, freeze_layer_norm: bool = True, pooler_in_head: bool = True):
        assert freeze_layer_norm, "Unfreezing LayerNorm is not supported. LayerNorm treated like other weights."  # model.py:373-375
        _lock_layer_groups(self.text_layer_groups(pooler_in_head), unlocked_layers)

    def set_grad_checkpointing(self, enable=True, impl="inline", keep_last=0):
        """model.py:377-379.  ``keep_last`` (native extension; an int for both towers or (image, text)): the last n blocks of a tower keep
        their activations instead of being recomputed (Transformer.set_grad_checkpointing)"""
        kv, kt = (keep_last, keep_last) if isinstance(keep_last, int) else keep_last
        self.visual.set_grad_checkpointing(enable, impl, kv)
        self.transformer.set_grad_checkpointing(enable, impl, kt)

    def activation_bytes_per_block(self, batch_size: int, text_rows=None):
        """(image, text) bytes one residual block saves for the backward when it is NOT recomputed: per row of width C the fp32 block
        output (4C), ln_1 / ln_2 outputs (2C + 2C), qkv (6C), the attention output (2C), the fp32 middle of the residual stream (4C), the
        8-bit gelu' and the MLP activation (1 + 2 bytes per element of the block's REAL MLP width: 12 C at mlp_ratio 4, 26 C for ViT-e-14) = 20 C +
        3 * mlp_width, + the attention row statistics.  A recomputed block keeps its 4C input only.
        ``text_rows``: rows of the packed text batch (default: every one of ``context_length`` positions)."""
        v, t = self.visual, self.transformer
        rows_v = batch_size * (v.grid_size[0] * v.grid_size[1] + 1)
        rows_t = batch_size * self.context_length if text_rows is None else int(text_rows)
        per = lambda rows, tr, sb: rows * ((12 + 2 * sb) * tr.width + 3 * tr.resblocks[0].mlp.c_fc.out_features + 4 * tr.resblocks[0].n_head + 8)
        # sb = bytes per element of the tower's residual stream: the block output and the middle of the stream are saved in that dtype
        return per(rows_v, v.transformer, 2 if self.image_stream != "fp32" else 4), per(rows_t, t, 4)

    def plan_grad_checkpointing(self, batch_size: int, budget_bytes: int, text_rows=None):
        """switch block recompute on and keep the activations of as many blocks as ``budget_bytes`` hold: the text tower's first (few
        rows once packed), then the image tower's.  Counted against the budget:
        32 C bytes per row for a kept block, 4 C for a recomputed one, and two blocks' worth of the larger tower for the transients of
        the backward (one block being recomputed + the gradients in flight).  Returns (image blocks kept, text blocks kept)."""
        bv, bt = self.activation_bytes_per_block(batch_size, text_rows)
        nv, nt = len(self.visual.transformer.resblocks), len(self.transformer.resblocks)
        left = int(budget_bytes) - 2 * max(bv, bt) - (nv * bv + nt * bt) // 8  # every block input is held either way
        kt = max(0, min(nt, left // max(1, bt - bt // 8)))
        left -= kt * (bt - bt // 8)
        kv = max(0, min(nv, left // max(1, bv - bv // 8)))
        self.set_grad_checkpointing(True, keep_last=(kv, kt))
        return kv, kt

    def no_weight_decay(self):
        return {"positional_embedding"} | {"visual." + n for n in self.visual.no_weight_decay()}

    def _sync_options(self, overlap=False):
        """execution switches of the model onto the two towers' operand caches (which the autograd Functions carry)"""
        self._cache.pair_wgrad = self.visual._cache.pair_wgrad = self.pair_wgrad and not overlap
        self._cache.deterministic = self.visual._cache.deterministic = self.deterministic
        self._cache.single_query = self.visual._cache.single_query = bool(self.pooled_single_query)
        self.visual._cache.stream = self.image_stream

    def encode_image(self, image, normalize: bool = False):
        self._cache.deterministic = self.visual._cache.deterministic = self.deterministic
        self.visual._cache.single_query = bool(self.pooled_single_query)
        self.visual._cache.stream = self.image_stream
        return self.visual(image, normalize)

    def encode_text(self, text, normalize: bool = False, _pack=None):
        B, L = text.shape
        if L != self.context_length:
            raise RuntimeError(f"text length {L} != context_length {self.context_length}")
        self._cache.deterministic = self.visual._cache.deterministic = self.deterministic
        self._cache.single_query = bool(self.pooled_single_query)
        if self.pack_text:
            pack = (_pack if _pack is not None else _TextPack(text, self.vocab_size, self.attn_buckets)).finish()
            x = _TextEmbedFn.apply(text, self.token_embedding.weight, self.positional_embedding, pack, self.deterministic)
            if _pooled_last_block_ok(self):
                x = self.transformer(x, self._cache, B, L, True, pack.layout, pack.last_row)  # [B, C]: the EOT rows
                rows = torch.arange(B, device=x.device, dtype=torch.int32)
                return _HeadFn.apply(x, self.ln_final.weight, self.ln_final.bias, self.text_projection, rows, self._cache, B, 0, normalize)
            x = self.transformer(x, self._cache, B, L, True, pack.layout)
            # L = 0: last_row holds absolute rows of the packed matrix
            return _HeadFn.apply(x, self.ln_final.weight, self.ln_final.bias, self.text_projection, pack.last_row, self._cache, B, 0, normalize)
        # ids outside the vocabulary raise like nn.Embedding (model.py:399); the packed path gets the count with its plan's read-back
        if int(ops.token_range_check(text.contiguous(), self.vocab_size)) != 0:
            raise IndexError(f"index out of range in self: token id(s) outside [0, {self.vocab_size}) (token_embedding has {self.vocab_size} rows)")
        x = _TextEmbedFn.apply(text, self.token_embedding.weight, self.positional_embedding, None, self.deterministic)
        idx = ops.argmax_rows(text.contiguous())
        if _pooled_last_block_ok(self):
            rows = torch.arange(B, device=x.device, dtype=torch.int32)
            x = self.transformer(x, self._cache, B, L, True, None, rows * L + idx)
            return _HeadFn.apply(x, self.ln_final.weight, self.ln_final.bias, self.text_projection, rows, self._cache, B, 0, normalize)
        x = self.transformer(x, self._cache, B, L, True)
        return _HeadFn.apply(x, self.ln_final.weight, self.ln_final.bias, self.text_projection, idx, self._cache, B, L, normalize)

    def get_logits(self, image, text):
        """model.py:413-420: `logit_scale.exp() * image_features @ text_features.T` (+ logit_bias), and its transpose, through the library's own GEMM like
        every other product of the path: (s I) rounded to bf16 -- the reference evaluates (s I) first, too, and the loss multiplies exactly these
        operands -- times T^T on MFMA with fp32 accumulation and output, the bias in the epilogue.  Differentiable like the reference's (``_LogitsFn``:
        a caller may put its own loss on these logits); under no_grad / inference_mode (the reference's callers: train.py:536-713) nothing is saved."""
        i, t = self.encode_image(image, True), self.encode_text(text, True)
        li = _LogitsFn.apply(i, t, self.logit_scale, self.logit_bias)
        return li, li.T

    def forward(self, image: Optional[torch.Tensor] = None, text: Optional[torch.Tensor] = None):
        # the packed text layout is planned first: its 4-byte read-back then completes while the image tower is being enqueued
        pack = _TextPack(text, self.vocab_size, self.attn_buckets) if (text is not None and self.pack_text) else None
        overlap = self.tower_streams and image is not None and text is not None
        self._sync_options(overlap)
        if overlap:
            dev = text.device
            cur = torch.cuda.current_stream(dev)
            side = self._tower_side.get(dev)
            if side is None:
                side = self._tower_side[dev] = torch.cuda.Stream(device=dev)
            serial = self.tower_streams == "serial"  # one tower at a time, on the same two streams (see _AfterStream)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                image_features = self.encode_image(image, normalize=True)
                if serial and image_features.requires_grad:
                    image_features = _AfterStream.apply(image_features, cur)
            if serial:
                cur.wait_stream(side)
            text_features = self.encode_text(text, normalize=True, _pack=pack)
            image.record_stream(side)
            cur.wait_stream(side)
            image_features.record_stream(cur)
        else:
            image_features = self.encode_image(image, normalize=True) if image is not None else None
            text_features = self.encode_text(text, normalize=True, _pack=pack) if text is not None else None
        if self.output_dict:
            out = {"image_features": image_features, "text_features": text_features, "logit_scale": self.logit_scale.exp()}
            if self.logit_bias is not None:
                out["logit_bias"] = self.logit_bias.clone()
            return out
        if self.logit_bias is not None:
            return image_features, text_features, self.logit_scale.exp(), self.logit_bias.clone()
        return image_features, text_features, self.logit_scale.exp()


_TEXT_KEYS = ("text_projection", "positional_embedding", "token_embedding", "transformer", "ln_final")


def convert_from_custom_text_state_dict(state_dict: dict) -> dict:
    """Inverse of the reference's ``convert_to_custom_text_state_dict`` (model.py:772-787): checkpoints written by a
    ``CustomTextCLIP`` keep the text tower under ``text.*``; NativeCLIP uses the ``CLIP`` layout (text tower unpacked onto the
    model, model.py:351-360).  ``text.attn_mask`` (a persistent buffer there) is dropped."""
    if not any(k.startswith("text.") for k in state_dict):
        return state_dict
    out = {}
    for k, v in state_dict.items():
        if k.startswith("text."):
            k2 = k[len("text."):]
            if k2 == "attn_mask":
                continue
            if not any(k2.startswith(p) for p in _TEXT_KEYS):
                raise KeyError(f"unexpected text-tower key {k!r} (only the built-in text transformer is supported)")
            k = k2
        out[k] = v
    return out


def convert_to_custom_text_state_dict(state_dict: dict) -> dict:
    """the reference's direction (model.py:772-787), for writing checkpoints a ``CustomTextCLIP`` can load"""
    if "text_projection" not in state_dict:
        return state_dict
    return {("text." + k if any(k.startswith(p) for p in _TEXT_KEYS) else k): v for k, v in state_dict.items()}


def create_model(model_name: str, pretrained: Optional[str] = None, precision: str = "amp_bf16", device="cuda",
                 output_dict: Optional[bool] = None, init_logit_scale=None, init_logit_bias=None, force_quick_gelu: bool = False, **model_kwargs):
    """Counterpart of ``open_clip.factory.create_model`` (factory.py:264-287) for the native path.
    ``pretrained`` may be a local ``.pt`` state-dict path (no hub access); ``precision`` must be an
    amp_bf16-equivalent mode (the kernels implement exactly that policy)."""
    if precision not in ("amp_bf16", "amp_bfloat16"):
        # the kernels implement exactly one policy: fp32 master weights, bf16 GEMM / attention operands, fp32 accumulation and
        # statistics (= --precision amp_bf16, precision.py:6-17).  fp32 / pure-bf16 / fp16 callers would silently get other numerics.
        raise ValueError(f"precision {precision!r} is not supported by the native path: it implements amp_bf16 semantics only")
    cfg = get_model_config(model_name)
    extra_model_kwargs = {}
    for k, val in model_kwargs.items():
        if k in ("embed_dim", "vision_cfg", "text_cfg"):
            cfg[k] = val
        else:
            extra_model_kwargs[k] = val
    for k in cfg:
        if k not in ("embed_dim", "vision_cfg", "text_cfg"):
            extra_model_kwargs.setdefault(k, cfg[k])
    # a registered config may carry init_logit_scale / init_logit_bias / output_dict itself (the reference's SigLIP JSONs do); the explicit
    # argument wins over the config's value, as in the reference (factory.py:547-556)
    kw = {k: extra_model_kwargs.pop(k) for k in ("init_logit_scale", "init_logit_bias") if k in extra_model_kwargs}
    cfg_output_dict = extra_model_kwargs.pop("output_dict", None)
    if init_logit_scale is not None:
        kw["init_logit_scale"] = init_logit_scale
    if init_logit_bias is not None:
        kw["init_logit_bias"] = init_logit_bias
    kw = {k: v for k, v in kw.items() if v is not None}
    if force_quick_gelu:  # factory.py:521-523: override for checkpoints trained with QuickGELU
        extra_model_kwargs["quick_gelu"] = True
    out_dict = output_dict if output_dict is not None else cfg_output_dict
    model = NativeCLIP(cfg["embed_dim"], cfg["vision_cfg"], cfg["text_cfg"], output_dict=bool(out_dict), **kw, **extra_model_kwargs)
    if pretrained:
        sd = torch.load(pretrained, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
        sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
        model.load_state_dict(convert_from_custom_text_state_dict(sd), strict=True)
    return model.to(device)
