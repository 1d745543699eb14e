"""Contrastive losses of the hot path over the HIP kernels: drop-ins for ``open_clip.loss.ClipLoss`` and
``SigLipLoss`` (reference ``src/open_clip/loss.py:57-141``, ``:314-489``) with the same constructor arguments,
``forward`` signature and return convention, so ``CLIPTask(model, loss=NativeClipLoss(...))`` works
(clip_task.py:26,38-39).

Each loss is one autograd Function that computes the loss AND its feature / logit_scale gradients in the
forward (the logit gradient G is a by-product of the same pass over the logits), so the backward is a scale by
the incoming scalar.  Logits are bf16 x bf16 -> fp32 on MFMA (``ocn_gemm_nt``), statistics fp32.

Distributed semantics reproduce ``gather_features`` (loss.py:29-54) exactly -- including which operands carry
gradient in each (local_loss, gather_with_grad) mode -- with ONE packed all-gather of
[B, 2E] per step instead of two, and a reduce-scatter in the backward when ``gather_with_grad``.
"""
import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops

F32, BF16 = torch.float32, torch.bfloat16
USE_FUSED_CE = True  # tests flip this to compare the fused cross-entropy with the materialised-logits path


def _round_up(a, b):
    return (a + b - 1) // b * b


def _bf16_rows(x, scale=None):
    """fp32 [R,E] -> bf16 [R,E] (times the 1-element device tensor ``scale`` when given); E % 64 == 0 for the hot
    configs, zero-padded K otherwise"""
    R, E = x.shape
    Ep = _round_up(E, 64)
    if Ep == E:
        x = x.contiguous()
        return ops.cast_bf16(x) if scale is None else ops.cast_bf16_scaled(x, scale)
    out = torch.zeros(R, Ep, dtype=BF16, device=x.device)
    out[:, :E].copy_(x if scale is None else x * scale)
    return out


def _bf16_transposed(x, npad):
    """fp32 [N,E] -> bf16 [E, npad] with zero columns beyond N (K-operand of the G @ Y product)"""
    N, E = x.shape
    out = torch.zeros(E, npad, dtype=BF16, device=x.device)
    if npad == N:
        ops.cast_transpose_bf16(x.contiguous(), out)
    else:
        out[:, :N].copy_(ops.cast_transpose_bf16(x.contiguous()))
    return out


def _reduce_scatter_sum(out, inp, comm=None):
    """reduce-scatter(sum) of inp [W*B, X] into out [B, X]; RCCL does it natively (through ``comm`` = a NativeComm, or the process
    group), gloo (CPU tests) lacks the collective, so fall back to all-reduce + slice there (same result)."""
    if comm is not None:
        comm.reduce_scatter_sum(out, inp)
    elif dist.get_backend() == "gloo":
        tmp = inp.clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
        r, b = dist.get_rank(), out.shape[0]
        out.copy_(tmp[r * b:(r + 1) * b])
    else:
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM)


class _PairTerm:
    """One logits matrix  L = (s X) @ Y^T (+ bias)  [R, N]  and what hangs off it.

    ``scale`` is a 1-element fp32 DEVICE tensor: the host never reads logit_scale (no synchronisation in the middle of the
    step).  It is folded into the bf16 cast of X -- `logit_scale * image_features @ text_features.T` (loss.py:103-110)
    evaluates (s X) first, too -- so the logits GEMM, the G^T (s X) product and the kernels take no scalar from the host;
    the two places where s multiplies / divides a small result are 1-element device ops."""

    def __init__(self, X, Y, scale):
        self.X, self.Y, self.s = X, Y, scale
        self.R, self.N, self.E = X.shape[0], Y.shape[0], X.shape[1]
        self.xs16, self.y16 = _bf16_rows(X, scale), _bf16_rows(Y)
        self.ldg = _round_up(self.N, 64)
        self.logits = None
        self._bias = None
        self._onehot = None  # (label_offset, grad_scale) once softmax_ce has filled G
        self.rowscale = None  # fused one-pass cross-entropy: G holds exp(logit - shift_r); the softmax part of the gradient is G * rowscale[:, None]
        self.deterministic = False  # reproducible sums (set by the loss Function): the rows' loss / d-scale contributions are added in a fixed order
        # the logit gradient's operand matrix: every column below N is written by the loss kernel; only the padding (K of the G @ Y product in steps of 64) must be zero
        self.G = torch.empty(self.R, self.ldg, dtype=BF16, device=X.device)
        if self.ldg > self.N:
            self.G[:, self.N:].zero_()

    def compute_logits(self, bias=None):
        """``bias``: None or a 1-element device tensor (SigLIP's logit_bias), broadcast to the epilogue's bias vector.  The fp32
        logits are only materialised when a consumer needs them (``siglip``; shapes the fused CE does not take)."""
        self._bias = bias
        return self

    def _materialise(self):
        if self.logits is None:
            self.logits = torch.empty(self.R, _round_up(self.N, 4), dtype=F32, device=self.X.device)[:, :self.N]
            bvec = None if self._bias is None else self._bias.detach().reshape(1).to(F32).expand(self.N).contiguous()
            ops.gemm_nt(ops.EPI_F32, self.xs16, self.y16, self.logits, bias=bvec)  # bias fused in the epilogue
        return self.logits

    def softmax_ce(self, label_offset, loss_scale, grad_scale, acc):
        """rows' CE against arange+offset -> acc[0] += loss, acc[1] += sum(G * logits) (= s * d/dscale); fills G.  Without a bias and
        on GEMM-sized shapes the logits never exist in memory: ONE pass of the MFMA GEMM with a cross-entropy epilogue (``ocn_fused_logits_ce``,
        round 6: G = exp(logit - shift_r) and ``rowscale`` = grad_scale / row sum; two passes until round 5) -- at the row-sharded global loss of
        8 GPUs ([4096, 32768] per matrix) that is 256 MiB of G written instead of 512 MiB of fp32 logits written and read three times besides."""
        # both kernels leave G = softmax * grad_scale; the -onehot * grad_scale part of the logit gradient is applied in dX / dY below, exactly
        self._onehot = (int(label_offset), float(grad_scale))
        if self.deterministic:
            # reproducible form: materialised logits, every row's contributions written to their own slot and added up in a fixed order (the fused
            # form's epilogues add per-wave partial sums atomically)
            rows = torch.empty(self.R, 3, dtype=F32, device=self.X.device)
            ops.softmax_ce_rows(self._materialise(), self.G, self.N, label_offset, loss_scale, grad_scale, 1.0, acc[0:1], acc[1:2], det_rows=rows)
            tot = torch.zeros(3, dtype=F32, device=self.X.device)
            ops.colsum_f32(rows, tot, deterministic=True)
            acc[0:2].add_(tot[0:2])
            return
        if self._bias is None and USE_FUSED_CE and ops.fused_logits_ce_supported(self.R, self.N, self.xs16.shape[1]):
            self.rowscale = ops.fused_logits_ce(self.xs16, self.y16, self.G, self.N, label_offset, loss_scale, grad_scale, acc[0:1], acc[1:2])
            return
        ops.softmax_ce_rows(self._materialise(), self.G, self.N, label_offset, loss_scale, grad_scale, 1.0, acc[0:1], acc[1:2])

    def siglip(self, label_offset, negative_only, loss_scale, grad_scale, acc):
        """acc[0] += loss, acc[1] += sum(g * (logits - bias)) (= s * d/dscale; the bias is read on the device and subtracted per element),
        acc[2] += sum(g) (= d/dbias)"""
        self._onehot = None if negative_only else (int(label_offset), float(grad_scale))  # G = sigmoid * grad_scale; the positives' -1 in dX / dY
        if self.deterministic:
            rows = torch.empty(self.R, 3, dtype=F32, device=self.X.device)
            ops.siglip_rows(self._materialise(), self.G, self.N, label_offset, negative_only, self._bias_dev(), loss_scale, grad_scale, 1.0,
                            acc[0:1], acc[1:2], acc[2:3], det_rows=rows)
            tot = torch.zeros(3, dtype=F32, device=self.X.device)
            ops.colsum_f32(rows, tot, deterministic=True)
            acc[0:3].add_(tot)
            return
        ops.siglip_rows(self._materialise(), self.G, self.N, label_offset, negative_only, self._bias_dev(), loss_scale, grad_scale, 1.0,
                        acc[0:1], acc[1:2], acc[2:3])

    def _bias_dev(self):
        return 0.0 if self._bias is None else self._bias.detach().reshape(1).to(F32).contiguous()

    def dX(self):
        """s * G @ Y  -> [R, E] fp32"""
        yt = _bf16_transposed(self.Y, self.ldg)
        out = torch.empty(self.R, self.E, dtype=F32, device=self.X.device)
        ks = ops.gemm_nt_splitk_plan(self.R, self.E, self.ldg) if self.E % 8 == 0 else 1
        if ks > 1:
            # few output tiles, long K (one rank's rows of the 8-GPU loss: [4096, 512] from K = 32768 is 32 tiles): K in slices that fill the chip; the
            # row scale of the one-pass cross-entropy, the exact -onehot part (see below) and logit_scale ride in the slices' reduction
            off, gs = self._onehot if self._onehot is not None else (0, 0.0)
            sub = self.y16[off:off + self.R, :self.E] if self._onehot is not None else None
            return ops.gemm_nt_splitk(self.G, yt, out, ks, rowscale=self.rowscale, sub_rows=sub, sub_alpha=gs, scale=self.s)
        ops.gemm_nt(ops.EPI_F32, self.G, yt, out)
        if self.rowscale is not None:  # one-pass cross-entropy: softmax * grad_scale = G * rowscale[:, None]: the row scale on the [R, E] result
            out.mul_(self.rowscale[:, None])
        if self._onehot is not None:
            # The label column of the logit gradient, (p - 1) * gs: its "p" is in G, its "-1" is applied here in fp32.  In bf16 (p - 1) * gs rounds
            # to -gs: the lost p_label (~ 1 / N) is tiny per row but has the same sign in EVERY row, and the parameter gradients are sums over the
            # batch in which everything else nearly cancels (batch 4096 against the fp32 reference: |sum_b error_b| / |sum_b gradient_b| of the
            # text-feature gradient 0.235 -> 0.004, the error of every parameter gradient more than halved; profiles/r04_parity_report.txt).
            # It is taken from the SAME bf16-rounded operand rows the GEMM multiplied: sum_i G_ij is ~ 0 for every column j as well, so the
            # rounding errors of the operand rows cancel in a sum over the batch only if both parts of the gradient see the same rounded rows
            # (with the exact rows the label part stops cancelling them: visible at batch 8, where nothing averages them out).
            off, gs = self._onehot
            out.sub_(self.y16[off:off + self.R, :self.E].float(), alpha=gs)
        return out.mul_(self.s)

    def dY(self, into=None):
        """G^T @ (s X) -> [N, E] fp32.  ``into``: a zeroed fp32 [N, E] view (row stride free) to accumulate into instead of a fresh buffer -- the two
        directions' results side by side in the [N, 2E] payload of the reduce-scatter, without a torch.cat of 2 x 64 MiB at N = 32768"""
        Np = _round_up(self.N, 8)
        Ep = self.xs16.shape[1]
        if into is not None and Np == self.N and Ep == self.E:
            out = into
        else:
            out = torch.zeros(Np, Ep, dtype=F32, device=self.X.device)
        if self.rowscale is not None:
            # one-pass cross-entropy: G^T @ (s X) with G = G' * rowscale[:, None]: the row scale goes onto the [R, E] operand, xw = bf16(rowscale_r * (s X)_r)
            xw16 = ops.scale_rows_bf16(self.xs16, self.rowscale)
            ops.gemm_tn_accum(self.G[:, :Np], xw16, out, None, 1.0, self.deterministic)
            if self._onehot is not None:
                # the label part from the SAME rounded rows the product multiplied (see dX): sum_j G'_rj xw_r = (grad_scale / rowscale_r) * xw_r
                off, gs = self._onehot
                ops.sub_scaled_rows(out[off:off + self.R], xw16, self.rowscale, gs)
            return self._into(out, into)
        ops.gemm_tn_accum(self.G[:, :Np], self.xs16, out, None, 1.0, self.deterministic)
        if self._onehot is not None:
            off, gs = self._onehot
            out[off:off + self.R].sub_(self.xs16.float(), alpha=gs)  # xs16 = bf16(s X): the rows the G^T (s X) product multiplied
        return self._into(out, into)

    def _into(self, out, into):
        if into is None or out is into:
            return out[:self.N, :self.E]
        into.copy_(out[:self.N, :self.E])
        return into


def _all_gather(out, inp, comm=None):
    if comm is not None:
        comm.all_gather_into_tensor(out, inp)
    else:
        dist.all_gather_into_tensor(out, inp)


def _all_reduce_sum(t, comm=None):
    if comm is not None:
        comm.all_reduce_sum(t)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def _device_scalar(t, dev):
    """1-element fp32 device tensor with the value of ``t`` (a 0-d parameter / tensor); no host read"""
    return t.detach().to(device=dev, dtype=F32).reshape(1)


def _term(X, Y, s, deterministic):
    t = PairTerm(X, Y, s)
    t.deterministic = bool(deterministic)
    return t


class _ClipLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image_features, text_features, logit_scale, local_loss, gather_with_grad, rank, world_size, row_sharded=False, comm=None,
                logit_bias=None, deterministic=False):
        I, T = image_features.detach().float().contiguous(), text_features.detach().float().contiguous()
        dev = I.device
        s = _device_scalar(logit_scale, dev)  # stays on the device: no host synchronisation inside the step
        B, E = I.shape
        acc = torch.zeros(2, dtype=F32, device=dev)  # [loss_sum, sum(G * logits) = s * dscale]
        # a communicator handed in with world_size 1 (bench.py --native-comm on one GPU, tests) still runs the distributed form: every collective is
        # an identity then, but counts, dtypes and pointers travel the path a node's ranks use
        use_dist = world_size > 1 or comm is not None
        if use_dist:
            packed = torch.cat([I, T], dim=1)
            allp = torch.empty(world_size * B, 2 * E, dtype=F32, device=dev)
            _all_gather(allp, packed, comm)
            I_all, T_all = allp[:, :E].contiguous(), allp[:, E:].contiguous()
        if not use_dist:
            # loss.py:109-110: li = s I T^T, lt = s T I^T ; labels arange(B)
            ti = _term(I, T, s, deterministic).compute_logits()
            tt = _term(T, I, s, deterministic).compute_logits()
            for term in (ti, tt):
                term.softmax_ce(0, 0.5 / B, 0.5 / B, acc)
            dI = ti.dX() + tt.dY()
            dT = tt.dX() + ti.dY()
            d_all = None
        elif local_loss:
            # loss.py:103-104 + :82-83: li = s I_loc T_all^T, lt = s T_loc I_all^T, labels arange(B) + B*rank
            ti = _term(I, T_all, s, deterministic).compute_logits()
            tt = _term(T, I_all, s, deterministic).compute_logits()
            for term in (ti, tt):
                term.softmax_ce(B * rank, 0.5 / B, 0.5 / B, acc)
            dI, dT = ti.dX(), tt.dX()            # through the local operands
            d_all = None
            if gather_with_grad:                  # ... and through the gathered ones (summed over ranks in backward)
                d_all = torch.zeros(world_size * B, 2 * E, dtype=F32, device=dev)  # [N, 2E]: d I_all | d T_all, written in place
                tt.dY(into=d_all[:, :E])
                ti.dY(into=d_all[:, E:])
        elif row_sharded and not gather_with_grad:
            # Same loss and the same local gradients as the branch below (loss.py:106-107, :47-50), without its W-fold
            # redundancy: rank r evaluates only ITS rows of logits_per_image and logits_per_text
            # ([B, N] each, labels arange(B) + B*rank) with the global 1/(2N) weight, the row partial sums of the loss and of
            # d/d logit_scale are all-reduced, and the gradient that reaches this rank's features through the COLUMNS of the
            # other ranks' rows arrives by one reduce-scatter of [N, 2E] -- done here, since the loss computes its gradients
            # in the forward.  6 GEMMs of 2*B*N*E flops instead of 6 of 2*N*N*E.
            N = world_size * B
            ti = _term(I, T_all, s, deterministic).compute_logits()
            tt = _term(T, I_all, s, deterministic).compute_logits()
            for term in (ti, tt):
                term.softmax_ce(B * rank, 0.5 / N, 0.5 / N, acc)
            dI, dT = ti.dX(), tt.dX()
            through_cols = torch.zeros(N, 2 * E, dtype=F32, device=dev)  # [N, 2E]: d I_all | d T_all from my rows, written in place
            tt.dY(into=through_cols[:, :E])
            ti.dY(into=through_cols[:, E:])
            mine = torch.empty(B, 2 * E, dtype=F32, device=dev)
            _reduce_scatter_sum(mine, through_cols, comm)
            dI = dI + mine[:, :E]
            dT = dT + mine[:, E:]
            _all_reduce_sum(acc, comm)
            d_all = None
        else:
            # loss.py:106-107: li = s I_all T_all^T, lt = li^T ; labels arange(N)
            N = world_size * B
            ti = _term(I_all, T_all, s, deterministic).compute_logits()
            tt = _term(T_all, I_all, s, deterministic).compute_logits()
            for term in (ti, tt):
                term.softmax_ce(0, 0.5 / N, 0.5 / N, acc)
            dI_all = ti.dX() + tt.dY()
            dT_all = tt.dX() + ti.dY()
            lo, hi = rank * B, (rank + 1) * B
            if gather_with_grad:
                dI = torch.zeros(B, E, dtype=F32, device=dev)
                dT = torch.zeros(B, E, dtype=F32, device=dev)
                d_all = torch.cat([dI_all, dT_all], dim=1).contiguous()
            else:                                   # loss.py:47-50: only the local slot carries gradient
                dI, dT = dI_all[lo:hi].contiguous(), dT_all[lo:hi].contiguous()
                d_all = None
        acc[1:2].div_(s)  # d loss / d logit_scale = sum(G * logits) / s
        ctx.save_for_backward(dI, dT, acc, d_all)
        ctx.meta = (world_size, B, E, image_features.dtype, text_features.dtype, comm, logit_bias is not None)
        return acc[0].clone()

    @staticmethod
    def backward(ctx, gout):
        dI, dT, acc, d_all = ctx.saved_tensors
        world_size, B, E, idt, tdt, comm, has_bias = ctx.meta
        if d_all is not None:  # backward of the differentiable all-gather = reduce-scatter(sum) (loss.py:23-26)
            mine = torch.empty(B, 2 * E, dtype=F32, device=dI.device)
            _reduce_scatter_sum(mine, d_all, comm)
            dI = dI + mine[:, :E]
            dT = dT + mine[:, E:]
        # logit_bias shifts every logit of a row alike: its gradient is exactly zero (the softmax gradient of a row sums to 0), but it is
        # a real gradient, as in the reference (loss.py:111-113) -- DDP then sees the parameter reduced like every other one
        dbias = (gout * 0.0).reshape(()) if has_bias else None
        return (dI * gout).to(idt), (dT * gout).to(tdt), acc[1] * gout, None, None, None, None, None, None, dbias, None


PairTerm = _PairTerm  # the one seam tests replace to exercise the collective plumbing on CPU/gloo


class NativeClipLoss(nn.Module):
    """``open_clip.loss.ClipLoss`` (loss.py:57-141) on the HIP path.  ``cache_labels`` is accepted for
    signature parity; labels are an arange predicate inside the kernel (nothing to cache)."""

    def __init__(self, local_loss=False, gather_with_grad=False, cache_labels=False, rank=0, world_size=1, row_sharded=False, comm=None,
                 deterministic=False):
        super().__init__()
        # native extension: loss and d/d logit_scale summed in a fixed order (materialised logits, one slot per row) instead of fp32 atomics --
        # the loss's share of torch.use_deterministic_algorithms; NativeCLIP(deterministic=True) is the towers'
        self.deterministic = bool(deterministic)
        self.comm = comm  # optional open_clip_amd.comm.NativeComm: the collectives through the C ABI instead of torch.distributed
        self.local_loss, self.gather_with_grad, self.cache_labels = local_loss, gather_with_grad, cache_labels
        self.rank, self.world_size = rank, world_size
        ops.multi_gpu_defaults(world_size)
        # native extension (not a reference argument): evaluate the global loss (local_loss=False, gather_with_grad=False)
        # by rows per rank -- identical value and gradients, 1/W of the logits work, one extra reduce-scatter
        self.row_sharded = row_sharded

    def forward(self, image_features, text_features, logit_scale, logit_bias=None, output_dict=False):
        # loss.py:111-113 adds logit_bias to both logit matrices; a constant added to every logit of a row changes neither the
        # softmax nor the cross-entropy: the loss ignores its value and hands back the exact (zero) gradient
        loss = _ClipLossFn.apply(image_features, text_features, logit_scale, self.local_loss, self.gather_with_grad,
                                 self.rank, self.world_size, self.row_sharded, self.comm, logit_bias, self.deterministic)
        return {"contrastive_loss": loss} if output_dict else loss


class _SigLipLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image_features, text_features, logit_scale, logit_bias, rank, world_size, comm=None, chunk_size=0, deterministic=False):
        I, T = image_features.detach().float().contiguous(), text_features.detach().float().contiguous()
        dev = I.device
        s, b = _device_scalar(logit_scale, dev), _device_scalar(logit_bias, dev)
        B, E = I.shape
        acc = torch.zeros(3, dtype=F32, device=dev)  # loss, dscale, dbias
        use_dist = world_size > 1 or comm is not None  # (a one-rank communicator still runs its collectives: see _ClipLossFn)
        if use_dist:
            T_all = torch.empty(world_size * B, E, dtype=F32, device=dev)
            _all_gather(T_all, T, comm)
        else:
            T_all = T
        # loss.py:406-489: local chunk with positives on the diagonal, every other rank's chunk negative-only.
        # One logits matrix [B, W*B]; the positive diagonal sits at column offset B*rank.
        if chunk_size and chunk_size < B:
            # loss.py:369-404 (_chunked_loss): the image rows in chunks of chunk_size against all texts -- peak memory O(chunk_size * N)
            # for logits and their gradient instead of O(B * N); the positive of chunk row k is column B*rank + i + k.  Same sums, same
            # gradients (a row's terms never leave its chunk).
            dI = torch.empty(B, E, dtype=F32, device=dev)
            dT_all = torch.zeros(T_all.shape[0], E, dtype=F32, device=dev)
            for i in range(0, B, chunk_size):
                term = _term(I[i:i + chunk_size], T_all, s, deterministic).compute_logits(bias=b)
                term.siglip(B * rank + i, 0, 1.0 / B, 1.0 / B, acc)
                dI[i:i + chunk_size] = term.dX()
                dT_all += term.dY()
                del term
        else:
            term = _term(I, T_all, s, deterministic).compute_logits(bias=b)
            term.siglip(B * rank, 0, 1.0 / B, 1.0 / B, acc)
            dI = term.dX()
            dT_all = term.dY().contiguous()
        acc[1:2].div_(s)  # d/dscale = sum(g * (logits - bias)) / s (the bias is subtracted per element inside the kernel)
        ctx.save_for_backward(dI, dT_all, acc)
        ctx.meta = (use_dist, B, E, image_features.dtype, text_features.dtype, comm)
        return acc[0].clone()

    @staticmethod
    def backward(ctx, gout):
        dI, dT_all, acc = ctx.saved_tensors
        use_dist, B, E, idt, tdt, comm = ctx.meta
        if use_dist:  # reverse of the neighbour exchange (loss.py:279-311): every chunk's grad returns to its owner
            dT = torch.empty(B, E, dtype=F32, device=dI.device)
            _reduce_scatter_sum(dT, dT_all, comm)
        else:
            dT = dT_all
        return (dI * gout).to(idt), (dT * gout).to(tdt), acc[1] * gout, acc[2] * gout, None, None, None, None, None


class NativeSigLipLoss(nn.Module):
    """``open_clip.loss.SigLipLoss`` (loss.py:314-489).  The W-1 neighbour exchanges of 'bidir'/'shift' are
    replaced by one all-gather of the text features (xGMI is fully connected; the payload is 2 MiB per rank)
    and one reduce-scatter in the backward -- the loss value and every gradient are identical.  ``chunk_size`` > 0 evaluates the image
    rows in chunks of that many (loss.py:369-404: peak memory O(chunk_size * N) for the logits and their gradient)."""

    def __init__(self, cache_labels=False, rank=0, world_size=1, dist_impl=None, chunk_size=0, comm=None, deterministic=False):
        super().__init__()
        self.deterministic = bool(deterministic)  # as NativeClipLoss: the three sums in a fixed order instead of fp32 atomics
        self.comm = comm
        self.cache_labels, self.rank, self.world_size = cache_labels, rank, world_size
        ops.multi_gpu_defaults(world_size)
        # loss.py:336-338: 'bidir' (default) / 'shift' / 'reduce' / 'gather' are four TRANSPORTS of the same sum -- every rank adds the negative-only loss
        # of its images against every other rank's text features.  The native form is the 'gather' one for all four (one all-gather forward, one
        # reduce-scatter backward: identical value and gradients, tests/test_dist_loss_gloo.py runs every name against the reference's vectors); the
        # name is kept for introspection, anything else is rejected as the reference's constructor rejects it.
        self.dist_impl = dist_impl or "bidir"
        if self.dist_impl not in ("bidir", "shift", "reduce", "gather"):
            raise ValueError(f"NativeSigLipLoss: dist_impl {dist_impl!r} is not one of 'bidir', 'shift', 'reduce', 'gather' (open_clip/loss.py:338)")
        self.chunk_size = chunk_size

    def forward(self, image_features, text_features, logit_scale, logit_bias, output_dict=False):
        loss = _SigLipLossFn.apply(image_features, text_features, logit_scale, logit_bias, self.rank, self.world_size, self.comm, self.chunk_size,
                                   self.deterministic)
        return {"contrastive_loss": loss} if output_dict else loss
