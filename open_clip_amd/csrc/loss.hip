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

// ---- one-pass fused logits + cross-entropy (round 6) --------------------------------------------------------------------------------------
// max_j |y_j|^2 over the rows of Y (bf16 [N, E]) -> *out (fp32, zeroed by the caller; non-negative floats order like their bit patterns): a wave per row,
// two rows in flight, the workgroup's maximum through LDS and ONE atomic per workgroup (one per wave from 8192 workgroups serialised: 375 us at N = 32768)
__global__ __launch_bounds__(256) void ce_ynorm_kernel(const bf16* __restrict__ Y, int ldy, int N, int E, unsigned* __restrict__ out) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float best = 0.f;
    const int stride = gridDim.x * 4;
    for (int row = blockIdx.x * 4 + wave; row < N; row += 2 * stride) {
        const int row2 = row + stride < N ? row + stride : row;
        float q = 0.f, q2 = 0.f;
        for (int c = lane * 8; c < E; c += 512) {
            const bf16x8 v = *(const bf16x8*)(Y + (size_t)row * ldy + c), w = *(const bf16x8*)(Y + (size_t)row2 * ldy + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                q += bf2f(v[e]) * bf2f(v[e]);
                q2 += bf2f(w[e]) * bf2f(w[e]);
            }
        }
        best = fmaxf(best, fmaxf(wave_sum(q), wave_sum(q2)));
    }
    if (lane == 0) red[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __builtin_bit_cast(unsigned, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

// row r: its label logit l_rr = <x_r, y_{label_offset + r}> (fp32 sum of the bf16 products the MFMA forms) and its shift
//   c_r = max(l_rr, |x_r| * max_j |y_j| - 70):  every entry of the row is <= |x_r| max|y| (Cauchy-Schwarz), so exp(l - c_r) <= e^70 and a row of 2^20 entries
//   sums below fp32's range; when the label logit is within 70 of that bound it IS the shift and the row sum is >= 1.
__global__ __launch_bounds__(256) void ce_prep_rows_kernel(const bf16* __restrict__ X, int ldx, const bf16* __restrict__ Y, int ldy, int R, int E,
                                                            int label_offset, const float* __restrict__ ymax2, float* __restrict__ label_logit,
                                                            float* __restrict__ shift2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= R) return;
    float d = 0.f, q = 0.f;
    for (int c = lane * 8; c < E; c += 512) {
        const bf16x8 x = *(const bf16x8*)(X + (size_t)row * ldx + c);
        const bf16x8 y = *(const bf16x8*)(Y + (size_t)(label_offset + row) * ldy + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            d += bf2f(x[e]) * bf2f(y[e]);
            q += bf2f(x[e]) * bf2f(x[e]);
        }
    }
    d = wave_sum(d);
    q = wave_sum(q);
    if (lane == 0) {
        const float bound = sqrtf(q) * sqrtf(*ymax2);
        label_logit[row] = d;
        shift2[row] = fmaxf(d, bound - 70.0f) * 1.4426950408889634f;
    }
}

// row r: S = sum of the strips' sum e, SL = sum e * logit  ->  lse = c + ln S; loss_sum += (lse - l_rr) * loss_scale;
// dscale_sum += grad_scale * (SL / S - l_rr) (= sum_j (softmax_rj - [j = label]) * grad_scale * l_rj); rowscale[r] = grad_scale / S.
// A row whose S is not a positive finite number (every term under- or overflowed: only possible when |x| max|y| > 78 AND the row has no
// entry within 157 of that bound) is left to ce_fixup_kernel: bad[r] = 1, nothing is added here.
__global__ __launch_bounds__(256) void ce_finish_kernel(const float* __restrict__ stats, const float* __restrict__ label_logit, const float* __restrict__ shift2,
                                                         float* __restrict__ rowscale, int* __restrict__ bad, int R, int parts, float loss_scale,
                                                         float grad_scale, float* __restrict__ loss_sum, float* __restrict__ dscale_sum) {
    // 16 lanes per row: 16 rows per workgroup in flight (a wave per row left this kernel latency-bound: 32 us for 4096 rows of 512 partials)
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int row = blockIdx.x * 16 + grp;
    float c_loss = 0.f, c_ds = 0.f;
    float S = 0.f, SL = 0.f;
    if (row < R) {
        const float* st = stats + (size_t)row * parts * 2;
        for (int p = sub; p < parts; p += 16) {
            const float2 v = *(const float2*)(st + 2 * p);
            S += v.x;
            SL += v.y;
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        S += __shfl_xor(S, o, 64);
        SL += __shfl_xor(SL, o, 64);
    }
    if (row < R && sub == 0) {
        const bool ok = S > 0.f && S < INFINITY && fabsf(SL) < INFINITY;
        bad[row] = ok ? 0 : 1;
        if (ok) {
            const float ll = label_logit[row];
            c_loss = (shift2[row] * 0.6931471805599453f + __logf(S) - ll) * loss_scale;
            c_ds = grad_scale * (SL / S - ll);
            rowscale[row] = grad_scale / S;
        }
    }
    __shared__ float red[2][16];
    if (sub == 0) {
        red[0][grp] = c_loss;
        red[1][grp] = c_ds;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a0 = 0.f, a1 = 0.f;
        for (int i = 0; i < 16; ++i) {
            a0 += red[0][i];
            a1 += red[1][i];
        }
        unsafeAtomicAdd(loss_sum, a0);
        unsafeAtomicAdd(dscale_sum, a1);
    }
}

// The exact form for the rows ce_finish_kernel flagged (never one in training: logit_scale is clamped to 100 and features are unit vectors): the row's
// logits from the operands, its true maximum as the shift, G' = exp(l - max) rewritten for the row.  One workgroup per row; unflagged rows return at once.
__global__ __launch_bounds__(256) void ce_fixup_kernel(const bf16* __restrict__ X, int ldx, const bf16* __restrict__ Y, int ldy, int R, int N, int E,
                                                        int label_offset, const int* __restrict__ bad, bf16* __restrict__ G, int ldg,
                                                        float* __restrict__ rowscale, float loss_scale, float grad_scale,
                                                        float* __restrict__ loss_sum, float* __restrict__ dscale_sum) {
    const int row = blockIdx.x;
    if (!bad[row]) return;
    __shared__ float red[8];
    const bf16* xr = X + (size_t)row * ldx;
    auto logit = [&](int j) {
        float d = 0.f;
        for (int c = 0; c < E; c += 8) {
            const bf16x8 x = *(const bf16x8*)(xr + c), y = *(const bf16x8*)(Y + (size_t)j * ldy + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) d += bf2f(x[e]) * bf2f(y[e]);
        }
        return d;
    };
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < N; j += blockDim.x) mx = fmaxf(mx, logit(j));
    mx = block_max(mx, red);
    float S = 0.f, SL = 0.f;
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        const float l = logit(j), e = __expf(l - mx);
        S += e;
        SL += e * l;
        G[(size_t)row * ldg + j] = f2bf(e);
    }
    S = block_sum(S, red);
    SL = block_sum(SL, red);
    if (threadIdx.x == 0) {
        const float ll = logit(label_offset + row);
        rowscale[row] = grad_scale / S;
        unsafeAtomicAdd(loss_sum, (mx + __logf(S) - ll) * loss_scale);
        unsafeAtomicAdd(dscale_sum, grad_scale * (SL / S - ll));
    }
}

}  // namespace

// workspace of ocn_fused_logits_ce, in floats: [R][parts][2] strip partials + label logit [R] + shift [R] + bad-row flags [R] + max |y|^2 [1 (+3 pad)]
extern "C" int64_t ocn_fused_logits_ce_workspace_floats(int R, int N) { return (int64_t)R * (ocn_cdiv(N, 256) * 4) * 2 + 3 * (int64_t)R + 4; }

// Logits + cross-entropy without the logits, in ONE pass of the persistent NT GEMM over X . Y^T (round 6; until round 5: a statistics pass and a
// gradient pass = the logits GEMM twice).  The epilogue consumes the fp32 tile in registers: e_rj = exp(l_rj - c_r) with the per-row shift c_r of
// ce_prep_rows_kernel goes out as the bf16 matrix G' [R, ldg], its per-strip row sums to the workspace; ce_finish_kernel turns them into the loss,
// d/d logit_scale and rowscale[r] = grad_scale / sum_j e_rj.  The logit gradient is  G_rj = G'_rj * rowscale[r] - [j = label_r] * grad_scale:  the caller
// applies the row scale where it is free -- on the [R, E] result of G' @ Y and on the [R, E] operand of G'^T @ X (open_clip_amd/loss.py::_PairTerm) -- and
// the -onehot part exactly in fp32 as before (in bf16 the label entry (p - 1) * grad_scale loses its p: a common-mode bias over the batch).
// 2 R N E flops and R N 2 bytes of HBM writes; rows the shifted sums cannot represent are redone exactly by ce_fixup_kernel (see there).
extern "C" int ocn_fused_logits_ce(const void* X, int ldx, const void* Y, int ldy, int R, int N, int E, int label_offset, float loss_scale,
                                   float grad_scale, void* G, int ldg, float* workspace, float* rowscale, float* loss_sum, float* dscale_sum,
                                   ocn_stream_t stream) {
    OCN_CHECK_ARG(X && Y && G && workspace && rowscale && loss_sum && dscale_sum, "ocn_fused_logits_ce: null operand");
    OCN_CHECK_ARG(R > 0 && N > 0 && E > 0 && E % 128 == 0 && N % 8 == 0 && ldg % 8 == 0 && ldg >= N && ldx >= E && ldy >= E && ldx % 8 == 0 && ldy % 8 == 0,
                  "ocn_fused_logits_ce: unsupported shape R=%d N=%d E=%d (E must be a multiple of 128, N, ldg, ldx and ldy of 8)", R, N, E);
    OCN_CHECK_ARG(label_offset >= 0 && label_offset + R <= N, "ocn_fused_logits_ce: labels [%d,%d) outside N=%d", label_offset, label_offset + R, N);
    OCN_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)Y & 15) == 0 && ((uintptr_t)G & 15) == 0, "ocn_fused_logits_ce: operands must be 16-byte aligned");
    GemmNtArgs a;
    a.A = (const bf16*)X; a.B = (const bf16*)Y; a.out = G; a.bias = nullptr; a.resid = nullptr; a.aux = nullptr;
    a.lda = ldx; a.ldb = ldy; a.ldc = ldg; a.M = R; a.N = N; a.K = E; a.alpha = 1.0f;
    a.tiles_n = 0; a.ntiles = 0; a.band = 0; a.stagger = 0; a.first_wave = 0; a.ablate = 0; a.ksplit = 1; a.ws_stride = 0; a.rescue = nullptr;
    const int parts = ocn_cdiv(N, 256) * 4;
    a.ce_parts = parts; a.ce_label_offset = label_offset; a.ce_grad_scale = grad_scale;
    a.ce_stats = workspace;
    a.ce_label_logit = workspace + (size_t)R * parts * 2;
    float* shift2 = a.ce_label_logit + R;
    int* bad = (int*)(shift2 + R);
    float* ymax2 = (float*)(bad + R);
    a.ce_shift2 = shift2; a.ce_lse = nullptr; a.ce_dscale = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ymax2, 0, sizeof(float), st) != hipSuccess) { ocn_set_error("ocn_fused_logits_ce: hipMemsetAsync failed"); return OCN_ERR_LAUNCH; }
    hipLaunchKernelGGL(ce_ynorm_kernel, dim3(ocn_cdiv(N, 8) < 512 ? ocn_cdiv(N, 8) : 512), dim3(256), 0, st, (const bf16*)Y, ldy, N, E, (unsigned*)ymax2);
    hipLaunchKernelGGL(ce_prep_rows_kernel, dim3(ocn_cdiv(R, 4)), dim3(256), 0, st, (const bf16*)X, ldx, (const bf16*)Y, ldy, R, E, label_offset, ymax2,
                       a.ce_label_logit, shift2);
    OCN_CHECK_LAUNCH("ocn_fused_logits_ce");
    const int rc = ocn_launch_nt5((R % 256 == 0 && N % 256 == 0) ? OCN_EPI_CE_ONEPASS_FULL : OCN_EPI_CE_ONEPASS, a, st);
    if (rc != 0) { if (rc > 0) ocn_set_error("ocn_fused_logits_ce: shape not supported by the persistent GEMM"); return rc > 0 ? OCN_ERR_UNSUPPORTED : rc; }
    hipLaunchKernelGGL(ce_finish_kernel, dim3(ocn_cdiv(R, 16)), dim3(256), 0, st, a.ce_stats, a.ce_label_logit, shift2, rowscale, bad, R, parts, loss_scale,
                       grad_scale, loss_sum, dscale_sum);
    hipLaunchKernelGGL(ce_fixup_kernel, dim3(R), dim3(256), 0, st, (const bf16*)X, ldx, (const bf16*)Y, ldy, R, N, E, label_offset, bad, (bf16*)G, ldg, rowscale,
                       loss_scale, grad_scale, loss_sum, dscale_sum);
    OCN_CHECK_LAUNCH("ocn_fused_logits_ce");
    return OCN_OK;
}

namespace {
// out16[r, :] = bf16(scale[r] * x16[r, :])
__global__ __launch_bounds__(256) void scale_rows_bf16_kernel(const bf16* __restrict__ x, int ldx, const float* __restrict__ scale, bf16* __restrict__ out, int ldo, int R, int E) {
    const int e8 = E / 8;
    const long total = (long)R * e8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / e8), c = (int)(i % e8) * 8;
        const bf16x8 v = *(const bf16x8*)(x + (size_t)r * ldx + c);
        const float sc = scale[r];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(v[e]) * sc);
        *(bf16x8*)(out + (size_t)r * ldo + c) = o;
    }
}
// out[r, :] -= (alpha / scale[r]) * x16[r, :]
__global__ __launch_bounds__(256) void sub_scaled_rows_kernel(float* __restrict__ out, int ldo, const bf16* __restrict__ x, int ldx, const float* __restrict__ scale, float alpha, int R, int E) {
    const int e4 = E / 4;
    const long total = (long)R * e4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / e4), c = (int)(i % e4) * 4;
        const bf16x4 v = *(const bf16x4*)(x + (size_t)r * ldx + c);
        const float k = alpha / scale[r];
        f32x4 o = *(f32x4*)(out + (size_t)r * ldo + c);
        o = o - (f32x4){bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])} * k;
        *(f32x4*)(out + (size_t)r * ldo + c) = o;
    }
}
}  // namespace

// The caller's side of the one-pass cross-entropy's row scale (see ocn_fused_logits_ce): out_bf16[r] = bf16(scale[r] * x_bf16[r]) -- the [R, E] operand of
// G'^T @ X -- and out_f32[r] -= (alpha / scale[r]) * x_bf16[r] -- the label rows' -onehot part of that product, from the same rounded rows.  E % 8 == 0.
extern "C" int ocn_scale_rows_bf16(const void* x, int ldx, const float* scale, void* out, int ldo, int R, int E, ocn_stream_t stream) {
    OCN_CHECK_ARG(x && scale && out && R > 0 && E > 0 && E % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, "ocn_scale_rows_bf16: bad arguments");
    const long total = (long)R * (E / 8);
    hipLaunchKernelGGL(scale_rows_bf16_kernel, dim3((int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, scale,
                       (bf16*)out, ldo, R, E);
    OCN_CHECK_LAUNCH("ocn_scale_rows_bf16");
    return OCN_OK;
}
extern "C" int ocn_sub_scaled_rows(float* out, int ldo, const void* x, int ldx, const float* scale, float alpha, int R, int E, ocn_stream_t stream) {
    OCN_CHECK_ARG(out && x && scale && R > 0 && E > 0 && E % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0, "ocn_sub_scaled_rows: bad arguments");
    const long total = (long)R * (E / 4);
    hipLaunchKernelGGL(sub_scaled_rows_kernel, dim3((int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream, out, ldo, (const bf16*)x, ldx,
                       scale, alpha, R, E);
    OCN_CHECK_LAUNCH("ocn_sub_scaled_rows");
    return OCN_OK;
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
