// Row-wise contrastive-loss kernels over materialised fp32 logits (HBM-bound; one workgroup per row).
// Each produces the loss contribution, the logit gradient G (bf16, the operand of the dI/dT GEMMs) and the
// logit_scale (/ logit_bias) gradient reductions in one read of the logits + one write of G.
#include "gemm_args.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

OCN_DEV float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}
OCN_DEV float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = -INFINITY;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s = fmaxf(s, red[i]);
    return s;
}

// F.cross_entropy(logits, arange(R)+label_offset) (loss.py:78-89, :136-139), mean folded into loss_scale
__global__ __launch_bounds__(256) void softmax_ce_rows_kernel(const float* __restrict__ logits, int ld, bf16* __restrict__ G,
                                                               int ldg, int R, int N, int label_offset, float loss_scale,
                                                               float grad_scale, float inv_logit_scale,
                                                               float* __restrict__ loss_sum, float* __restrict__ dscale_sum, float* __restrict__ det_rows) {
    __shared__ float red[8];
    const int r = blockIdx.x;
    const float* row = logits + (size_t)r * ld;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < N; c += blockDim.x) mx = fmaxf(mx, row[c]);
    mx = block_max(mx, red);
    float s = 0.f;
    for (int c = threadIdx.x; c < N; c += blockDim.x) s += __expf(row[c] - mx);
    s = block_sum(s, red);
    const float lse = mx + __logf(s);
    const int label = r + label_offset;
    const float inv = 1.0f / s;
    float ds = 0.f;
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        const float v = row[c];
        const float pr = __expf(v - mx) * inv;
        G[(size_t)r * ldg + c] = f2bf(pr * grad_scale);  // softmax * grad_scale only: the caller applies -onehot * grad_scale exactly (see ocn_fused_logits_ce)
        ds += (pr - (c == label ? 1.f : 0.f)) * grad_scale * v;
    }
    ds = block_sum(ds, red);
    if (threadIdx.x == 0) {
        if (det_rows) {  // reproducible form: the row's contributions go to det_rows[r][0..2]; the caller adds the rows up in a fixed order
            det_rows[(size_t)r * 3] = (lse - row[label]) * loss_scale;
            det_rows[(size_t)r * 3 + 1] = ds * inv_logit_scale;
            det_rows[(size_t)r * 3 + 2] = 0.f;
        } else {
            unsafeAtomicAdd(loss_sum, (lse - row[label]) * loss_scale);
            unsafeAtomicAdd(dscale_sum, ds * inv_logit_scale);
        }
    }
}

// SigLIP pairwise sigmoid loss (loss.py:344-367): labels +1 on (r, r+label_offset) unless negative_only, else -1
__global__ __launch_bounds__(256) void siglip_rows_kernel(const float* __restrict__ logits, int ld, bf16* __restrict__ G, int ldg,
                                                           int R, int N, int label_offset, int negative_only, float bias_host,
                                                           const float* __restrict__ bias_dev, float loss_scale, float grad_scale, float inv_logit_scale,
                                                           float* __restrict__ loss_sum, float* __restrict__ dscale_sum,
                                                           float* __restrict__ dbias_sum, float* __restrict__ det_rows) {
    __shared__ float red[8];
    const int r = blockIdx.x;
    const float* row = logits + (size_t)r * ld;
    const int pos = negative_only ? -1 : r + label_offset;
    // the bias the logits carry, read on the device when the caller has it there (logit_bias is a parameter: no host read inside the step).  It is
    // subtracted PER ELEMENT: sum(g * logits) - bias * sum(g) afterwards is the difference of two numbers of size |bias| * sum|g| whose result --
    // d loss / d logit_scale -- can be orders of magnitude smaller (ViT-H-14 at batch 512: relative error 0.9 that way, round 5)
    const float bias = bias_dev ? *bias_dev : bias_host;
    float ls = 0.f, ds = 0.f, dbs = 0.f;
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        const float v = row[c];
        const float lab = (c == pos) ? 1.f : -1.f;
        const float z = lab * v;
        const float e = __expf(-fabsf(z));
        ls += fmaxf(-z, 0.f) + log1pf(e);                             // -logsigmoid(z)
        // d/dlogit = -lab * sigmoid(-z) = sigmoid(v) - [c == pos]: G holds sigmoid(v) * grad_scale ONLY, the caller applies the -[c == pos] * grad_scale
        // part exactly in fp32 (as for the cross-entropy, ocn_fused_logits_ce: (sigmoid - 1) rounded to bf16 loses its sigmoid)
        const float sig = (v >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e);  // sigmoid(v); e = exp(-|v|) as |z| = |v|
        const float g = (sig - (c == pos ? 1.f : 0.f)) * grad_scale;
        G[(size_t)r * ldg + c] = f2bf(sig * grad_scale);
        ds += g * (v - bias);
        dbs += g;
    }
    ls = block_sum(ls, red);
    ds = block_sum(ds, red);
    dbs = block_sum(dbs, red);
    if (threadIdx.x == 0) {
        if (det_rows) {
            det_rows[(size_t)r * 3] = ls * loss_scale;
            det_rows[(size_t)r * 3 + 1] = ds * inv_logit_scale;
            det_rows[(size_t)r * 3 + 2] = dbs;
        } else {
            unsafeAtomicAdd(loss_sum, ls * loss_scale);
            unsafeAtomicAdd(dscale_sum, ds * inv_logit_scale);
            unsafeAtomicAdd(dbias_sum, dbs);
        }
    }
}

// row r: combine the per-strip (max, sum exp) partials of the fused logits pass into lse[r]; loss_sum += (lse - label logit) * loss_scale
__global__ __launch_bounds__(256) void ce_lse_reduce_kernel(const float* __restrict__ stats, const float* __restrict__ label_logit, float* __restrict__ lse,
                                                             int R, int parts, float loss_scale, float* __restrict__ loss_sum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    float contrib = 0.f;
    if (row < R) {
        const float* st = stats + (size_t)row * parts * 2;
        float m = -INFINITY;
        for (int p = lane; p < parts; p += 64) m = fmaxf(m, st[2 * p]);
        m = wave_max(m);
        float l = 0.f;
        for (int p = lane; p < parts; p += 64) {
            const float mp = st[2 * p];
            if (mp != -INFINITY) l += st[2 * p + 1] * __expf(mp - m);
        }
        l = wave_sum(l);
        const float v = m + __logf(l);
        if (lane == 0) {
            lse[row] = v;
            contrib = (v - label_logit[row]) * loss_scale;
        }
    }
    __shared__ float red[4];
    if (lane == 0) red[wave] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
}

}  // namespace

extern "C" int64_t ocn_fused_logits_ce_workspace_floats(int R, int N) { return (int64_t)R * (ocn_cdiv(N, 256) * 4) * 2 + 2 * (int64_t)R; }

// Logits + cross-entropy without the logits: two passes of the persistent NT GEMM over X . Y^T whose epilogues consume the fp32
// tile in registers -- pass 1 leaves per-row (max, sum exp) partials (one per 64-column strip) and the label logit, a small kernel
// combines them into the row log-sum-exp and the loss, pass 2 recomputes the tile and writes G = softmax * grad_scale as bf16 plus
// sum((softmax - onehot) * grad_scale * logits).  The -onehot * grad_scale part of the logit gradient is NOT in G: the caller applies it
// exactly in fp32 (dX_r -= grad_scale * Y[label_r], dY[label_r] -= grad_scale * X_r) -- in bf16 the label entry (p - 1) * grad_scale
// rounds to -grad_scale, and the lost p is a common-mode bias that sums coherently over the batch.  4 * R * N * E flops instead of 2, and R * N * 2 bytes of HBM writes instead of R * N * (4 + 3 * 4 + 2).
extern "C" int ocn_fused_logits_ce(const void* X, int ldx, const void* Y, int ldy, int R, int N, int E, int label_offset, float loss_scale,
                                   float grad_scale, void* G, int ldg, float* workspace, float* loss_sum, float* dscale_sum, ocn_stream_t stream) {
    OCN_CHECK_ARG(X && Y && G && workspace && loss_sum && dscale_sum, "ocn_fused_logits_ce: null operand");
    OCN_CHECK_ARG(R > 0 && N > 0 && E > 0 && E % 128 == 0 && N % 8 == 0 && ldg % 8 == 0 && ldg >= N && ldx >= E && ldy >= E,
                  "ocn_fused_logits_ce: unsupported shape R=%d N=%d E=%d (E must be a multiple of 128, N and ldg of 8)", R, N, E);
    OCN_CHECK_ARG(label_offset >= 0 && label_offset + R <= N, "ocn_fused_logits_ce: labels [%d,%d) outside N=%d", label_offset, label_offset + R, N);
    OCN_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)Y & 15) == 0 && ((uintptr_t)G & 15) == 0, "ocn_fused_logits_ce: operands must be 16-byte aligned");
    GemmNtArgs a;
    a.A = (const bf16*)X; a.B = (const bf16*)Y; a.out = G; a.bias = nullptr; a.resid = nullptr; a.aux = nullptr;
    a.lda = ldx; a.ldb = ldy; a.ldc = ldg; a.M = R; a.N = N; a.K = E; a.alpha = 1.0f;
    a.tiles_n = 0; a.ntiles = 0; a.band = 0; a.stagger = 0; a.first_wave = 0; a.ablate = 0;
    const int parts = ocn_cdiv(N, 256) * 4;
    a.ce_parts = parts; a.ce_label_offset = label_offset; a.ce_grad_scale = grad_scale;
    a.ce_stats = workspace;
    a.ce_label_logit = workspace + (size_t)R * parts * 2;
    float* lse = a.ce_label_logit + R;
    a.ce_lse = lse; a.ce_dscale = dscale_sum;
    hipStream_t st = (hipStream_t)stream;
    int rc = ocn_launch_nt5(OCN_EPI_CE_STATS, a, st);
    if (rc != 0) { if (rc > 0) ocn_set_error("ocn_fused_logits_ce: shape not supported by the persistent GEMM"); return rc > 0 ? OCN_ERR_UNSUPPORTED : rc; }
    hipLaunchKernelGGL(ce_lse_reduce_kernel, dim3(ocn_cdiv(R, 4)), dim3(256), 0, st, a.ce_stats, a.ce_label_logit, lse, R, parts, loss_scale, loss_sum);
    OCN_CHECK_LAUNCH("ocn_fused_logits_ce");
    rc = ocn_launch_nt5(OCN_EPI_CE_GRAD, a, st);
    return rc > 0 ? OCN_ERR_UNSUPPORTED : rc;
}

extern "C" int ocn_softmax_ce_rows(const float* logits, int ld, void* G, int ldg, int R, int N, int label_offset,
                                   float loss_scale, float grad_scale, float inv_logit_scale, float* loss_sum,
                                   float* dscale_sum, float* det_rows, ocn_stream_t stream) {
    OCN_CHECK_ARG(logits && G && loss_sum && dscale_sum, "ocn_softmax_ce_rows: null operand");
    OCN_CHECK_ARG(R > 0 && N > 0 && ld >= N && ldg >= N, "ocn_softmax_ce_rows: bad shape R=%d N=%d", R, N);
    OCN_CHECK_ARG(label_offset >= 0 && label_offset + R <= N, "ocn_softmax_ce_rows: labels [%d,%d) outside N=%d", label_offset, label_offset + R, N);
    hipLaunchKernelGGL(softmax_ce_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, (bf16*)G, ldg, R, N,
                       label_offset, loss_scale, grad_scale, inv_logit_scale, loss_sum, dscale_sum, det_rows);
    OCN_CHECK_LAUNCH("ocn_softmax_ce_rows");
    return OCN_OK;
}

extern "C" int ocn_siglip_rows(const float* logits, int ld, void* G, int ldg, int R, int N, int label_offset, int negative_only,
                               float bias, const float* bias_dev, float loss_scale, float grad_scale, float inv_logit_scale, float* loss_sum,
                               float* dscale_sum, float* dbias_sum, float* det_rows, ocn_stream_t stream) {
    OCN_CHECK_ARG(logits && G && loss_sum && dscale_sum && dbias_sum, "ocn_siglip_rows: null operand");
    OCN_CHECK_ARG(R > 0 && N > 0 && ld >= N && ldg >= N, "ocn_siglip_rows: bad shape R=%d N=%d", R, N);
    hipLaunchKernelGGL(siglip_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, (bf16*)G, ldg, R, N,
                       label_offset, negative_only, bias, bias_dev, loss_scale, grad_scale, inv_logit_scale, loss_sum, dscale_sum, dbias_sum, det_rows);
    OCN_CHECK_LAUNCH("ocn_siglip_rows");
    return OCN_OK;
}
