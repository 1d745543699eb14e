// Row-wise contrastive-loss kernels over materialised fp32 logits (HBM-bound; one workgroup per row).
// Each produces the loss contribution, the logit gradient G (bf16, the operand of the dI/dT GEMMs) and the
// logit_scale (/ logit_bias) gradient reductions in one read of the logits + one write of G.
#include "ocn_common.h"

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
                                                               float* __restrict__ loss_sum, float* __restrict__ dscale_sum) {
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
        const float g = (__expf(v - mx) * inv - (c == label ? 1.f : 0.f)) * grad_scale;
        G[(size_t)r * ldg + c] = f2bf(g);
        ds += g * v;
    }
    ds = block_sum(ds, red);
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(loss_sum, (lse - row[label]) * loss_scale);
        unsafeAtomicAdd(dscale_sum, ds * inv_logit_scale);
    }
}

// SigLIP pairwise sigmoid loss (loss.py:344-367): labels +1 on (r, r+label_offset) unless negative_only, else -1
__global__ __launch_bounds__(256) void siglip_rows_kernel(const float* __restrict__ logits, int ld, bf16* __restrict__ G, int ldg,
                                                           int R, int N, int label_offset, int negative_only, float bias,
                                                           float loss_scale, float grad_scale, float inv_logit_scale,
                                                           float* __restrict__ loss_sum, float* __restrict__ dscale_sum,
                                                           float* __restrict__ dbias_sum) {
    __shared__ float red[8];
    const int r = blockIdx.x;
    const float* row = logits + (size_t)r * ld;
    const int pos = negative_only ? -1 : r + label_offset;
    float ls = 0.f, ds = 0.f, dbs = 0.f;
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        const float v = row[c];
        const float lab = (c == pos) ? 1.f : -1.f;
        const float z = lab * v;
        const float e = __expf(-fabsf(z));
        ls += fmaxf(-z, 0.f) + log1pf(e);                             // -logsigmoid(z)
        const float sig_neg = (z >= 0.f) ? e / (1.f + e) : 1.f / (1.f + e);  // sigmoid(-z)
        const float g = -lab * sig_neg * grad_scale;
        G[(size_t)r * ldg + c] = f2bf(g);
        ds += g * (v - bias);
        dbs += g;
    }
    ls = block_sum(ls, red);
    ds = block_sum(ds, red);
    dbs = block_sum(dbs, red);
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(loss_sum, ls * loss_scale);
        unsafeAtomicAdd(dscale_sum, ds * inv_logit_scale);
        unsafeAtomicAdd(dbias_sum, dbs);
    }
}

}  // namespace

extern "C" int ocn_softmax_ce_rows(const float* logits, int ld, void* G, int ldg, int R, int N, int label_offset,
                                   float loss_scale, float grad_scale, float inv_logit_scale, float* loss_sum,
                                   float* dscale_sum, ocn_stream_t stream) {
    OCN_CHECK_ARG(logits && G && loss_sum && dscale_sum, "ocn_softmax_ce_rows: null operand");
    OCN_CHECK_ARG(R > 0 && N > 0 && ld >= N && ldg >= N, "ocn_softmax_ce_rows: bad shape R=%d N=%d", R, N);
    OCN_CHECK_ARG(label_offset >= 0 && label_offset + R <= N, "ocn_softmax_ce_rows: labels [%d,%d) outside N=%d", label_offset, label_offset + R, N);
    hipLaunchKernelGGL(softmax_ce_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, (bf16*)G, ldg, R, N,
                       label_offset, loss_scale, grad_scale, inv_logit_scale, loss_sum, dscale_sum);
    OCN_CHECK_LAUNCH("ocn_softmax_ce_rows");
    return OCN_OK;
}

extern "C" int ocn_siglip_rows(const float* logits, int ld, void* G, int ldg, int R, int N, int label_offset, int negative_only,
                               float bias, float loss_scale, float grad_scale, float inv_logit_scale, float* loss_sum,
                               float* dscale_sum, float* dbias_sum, ocn_stream_t stream) {
    OCN_CHECK_ARG(logits && G && loss_sum && dscale_sum && dbias_sum, "ocn_siglip_rows: null operand");
    OCN_CHECK_ARG(R > 0 && N > 0 && ld >= N && ldg >= N, "ocn_siglip_rows: bad shape R=%d N=%d", R, N);
    hipLaunchKernelGGL(siglip_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, (bf16*)G, ldg, R, N,
                       label_offset, negative_only, bias, loss_scale, grad_scale, inv_logit_scale, loss_sum, dscale_sum, dbias_sum);
    OCN_CHECK_LAUNCH("ocn_siglip_rows");
    return OCN_OK;
}
