// Weight casts (fp32 master -> bf16 operand copies), grad-norm and AdamW (HBM-bound elementwise kernels).
#include "ocn_common.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

__global__ void cast_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long n) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(src + i * 4);
        bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *(bf16x4*)(dst + i * 4) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] = f2bf(src[n4 * 4 + threadIdx.x]);
}

// dst = bf16(src * *scale): the scale stays on the device (logit_scale.exp() of the loss, loss.py:103-110 `logit_scale * image_features`)
__global__ void cast_scaled_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long n, const float* __restrict__ scale) {
    const float s = *scale;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(src + i * 4);
        bf16x4 o = {f2bf(v[0] * s), f2bf(v[1] * s), f2bf(v[2] * s), f2bf(v[3] * s)};
        *(bf16x4*)(dst + i * 4) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] = f2bf(src[n4 * 4 + threadIdx.x] * s);
}

// dst[C,R] = src[R,C]^T, 64x64 tiles through LDS (both sides coalesced)
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int R, int C) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) dst[(size_t)c * R + r] = f2bf(tile[tx][i]);
    }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(x + i * 4);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = x[n4 * 4 + threadIdx.x];
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// torch.optim.AdamW (single-tensor form): p *= 1 - lr*wd; m,v EMA; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adamw_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             bf16* __restrict__ w16, long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                             float bc2_sqrt, const float* __restrict__ clip_coef) {
    const float gs = clip_coef ? *clip_coef : 1.0f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gs;
        float p = w[i] * (1.0f - lr * wd);
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);          // lerp form used by torch
        const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p -= (lr / bc1) * (mi / denom);
        w[i] = p;
        m[i] = mi;
        v[i] = vi;
        if (w16) w16[i] = f2bf(p);
    }
}


// ---- whole-model optimizer step in ONE launch --------------------------------------------------------
// The host describes every tensor by an AdamEntry and cuts the work into chunks {entry, index}: a chunk is 8192 consecutive
// elements (mode 0: 16-byte vector path, mode 2: scalar path for odd sizes / alignments) or one 64x64 tile of a 2-D GEMM
// weight (mode 1).  Besides torch.optim.AdamW's update the kernel refreshes the bf16 operand copies of the GEMM weights in
// the same pass -- the plain copy [rows, cols] and, in tile mode, the transposed copy [cols, rows] through LDS -- and applies
// the clip_grad_norm_ coefficient computed from the squared gradient norm that ocn_sumsq_multi left on the device.
struct AdamEntry {
    float* w;
    const float* g;
    float* m;
    float* v;
    bf16* w16n;
    bf16* w16t;
    long numel;
    int rows, cols, mode, pad;
    float lr, wd, bc1, bc2_sqrt;
};
static_assert(sizeof(AdamEntry) == 88, "AdamEntry layout is mirrored by open_clip_amd/optim.py");
constexpr int ADAM_CHUNK = 8192;

OCN_DEV float adam_one(float w, float gi, float& m, float& v, float lr, float wd, float b1, float b2, float eps, float bc1, float bc2_sqrt) {
    float p = w * (1.0f - lr * wd);
    m = m + (gi - m) * (1.0f - b1);
    v = v * b2 + gi * gi * (1.0f - b2);
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    return p - (lr / bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamEntry* __restrict__ entries, const int2* __restrict__ chunks,
                                                          float b1, float b2, float eps, const float* __restrict__ gnorm_sq, float max_norm) {
    __shared__ float tile[64][65];
    const int2 ch = chunks[blockIdx.x];
    const AdamEntry e = entries[ch.x];
    float gs = 1.0f;
    if (gnorm_sq) gs = fminf(1.0f, max_norm / (sqrtf(*gnorm_sq) + 1e-6f));  // torch.nn.utils.clip_grad_norm_
    if (e.mode == 1) {
        const int tiles_c = e.cols >> 6;
        const int r0 = (ch.y / tiles_c) * 64, c0 = (ch.y % tiles_c) * 64;
        const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
        const bool g_vec = ((uintptr_t)e.g & 15) == 0;  // gradients may be views into a DDP bucket at any 4-byte offset
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = ty + 16 * k;
            const size_t o = (size_t)(r0 + r) * e.cols + c0 + tx * 4;
            const f32x4 g4 = g_vec ? *(const f32x4*)(e.g + o) : (f32x4){e.g[o], e.g[o + 1], e.g[o + 2], e.g[o + 3]};
            const f32x4 w4 = *(const f32x4*)(e.w + o);
            f32x4 m4 = *(const f32x4*)(e.m + o), v4 = *(const f32x4*)(e.v + o), p4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float mj = m4[j], vj = v4[j];
                p4[j] = adam_one(w4[j], g4[j] * gs, mj, vj, e.lr, e.wd, b1, b2, eps, e.bc1, e.bc2_sqrt);
                m4[j] = mj;
                v4[j] = vj;
            }
            *(f32x4*)(e.w + o) = p4;
            *(f32x4*)(e.m + o) = m4;
            *(f32x4*)(e.v + o) = v4;
            if (e.w16n) *(bf16x4*)(e.w16n + o) = (bf16x4){f2bf(p4[0]), f2bf(p4[1]), f2bf(p4[2]), f2bf(p4[3])};
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[r][tx * 4 + j] = p4[j];
        }
        if (e.w16t) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = ty + 16 * k;  // row of the transposed copy
                const bf16x4 o4 = {f2bf(tile[tx * 4][c]), f2bf(tile[tx * 4 + 1][c]), f2bf(tile[tx * 4 + 2][c]), f2bf(tile[tx * 4 + 3][c])};
                *(bf16x4*)(e.w16t + (size_t)(c0 + c) * e.rows + r0 + tx * 4) = o4;
            }
        }
        return;
    }
    const long base = (long)ch.y * ADAM_CHUNK;
    const long end = base + ADAM_CHUNK < e.numel ? base + ADAM_CHUNK : e.numel;
    if (e.mode == 0) {
        for (long i = base + threadIdx.x * 4; i + 3 < end; i += 1024) {
            const f32x4 g4 = *(const f32x4*)(e.g + i), w4 = *(const f32x4*)(e.w + i);
            f32x4 m4 = *(const f32x4*)(e.m + i), v4 = *(const f32x4*)(e.v + i), p4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float mj = m4[j], vj = v4[j];
                p4[j] = adam_one(w4[j], g4[j] * gs, mj, vj, e.lr, e.wd, b1, b2, eps, e.bc1, e.bc2_sqrt);
                m4[j] = mj;
                v4[j] = vj;
            }
            *(f32x4*)(e.w + i) = p4;
            *(f32x4*)(e.m + i) = m4;
            *(f32x4*)(e.v + i) = v4;
            if (e.w16n) *(bf16x4*)(e.w16n + i) = (bf16x4){f2bf(p4[0]), f2bf(p4[1]), f2bf(p4[2]), f2bf(p4[3])};
        }
        const long tail = end - ((end - base) & 3);  // numel % 4 leftovers of the last chunk
        if (tail + threadIdx.x < end) {
            const long i = tail + threadIdx.x;
            float mi = e.m[i], vi = e.v[i];
            const float p = adam_one(e.w[i], e.g[i] * gs, mi, vi, e.lr, e.wd, b1, b2, eps, e.bc1, e.bc2_sqrt);
            e.w[i] = p; e.m[i] = mi; e.v[i] = vi;
            if (e.w16n) e.w16n[i] = f2bf(p);
        }
    } else {
        for (long i = base + threadIdx.x; i < end; i += 256) {
            float mi = e.m[i], vi = e.v[i];
            const float p = adam_one(e.w[i], e.g[i] * gs, mi, vi, e.lr, e.wd, b1, b2, eps, e.bc1, e.bc2_sqrt);
            e.w[i] = p; e.m[i] = mi; e.v[i] = vi;
            if (e.w16n) e.w16n[i] = f2bf(p);
        }
    }
}

// out[0] += sum over every tensor of g^2 (same entry / chunk tables as adamw_multi_kernel)
// chunk_ws != nullptr: the reproducible form -- the chunk's partial is stored to its own slot, sumsq_fold_kernel adds the slots in order
__global__ __launch_bounds__(256) void sumsq_multi_kernel(const AdamEntry* __restrict__ entries, const int2* __restrict__ chunks,
                                                          float* __restrict__ out, float* __restrict__ chunk_ws) {
    __shared__ float red[4];
    const int2 ch = chunks[blockIdx.x];
    const AdamEntry e = entries[ch.x];
    float s = 0.f;
    if (e.mode == 1) {
        const int tiles_c = e.cols >> 6;
        const int r0 = (ch.y / tiles_c) * 64, c0 = (ch.y % tiles_c) * 64;
        const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* gp = e.g + (size_t)(r0 + ty + 16 * k) * e.cols + c0 + tx * 4;
            s += gp[0] * gp[0] + gp[1] * gp[1] + gp[2] * gp[2] + gp[3] * gp[3];
        }
    } else {
        const long base = (long)ch.y * ADAM_CHUNK;
        const long end = base + ADAM_CHUNK < e.numel ? base + ADAM_CHUNK : e.numel;
        for (long i = base + threadIdx.x; i < end; i += 256) s += e.g[i] * e.g[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = red[0] + red[1] + red[2] + red[3];
        if (chunk_ws) chunk_ws[blockIdx.x] = t; else unsafeAtomicAdd(out, t);
    }
}

// out[0] += sum of ws[0 .. n) in ONE fixed order: thread t adds ws[t], ws[t + 256], ... ; the 256 partials fold through the wave butterflies and
// four LDS slots (the same additions in the same order in every run)
__global__ __launch_bounds__(256) void sumsq_fold_kernel(const float* __restrict__ ws, int n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += ws[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] += ((red[0] + red[1]) + red[2]) + red[3];
}

int grid_for(long items, int block) {
    long g = (items + block - 1) / block;
    return (int)(g < 4096 ? (g > 0 ? g : 1) : 4096);
}

}  // namespace

extern "C" int ocn_cast_f32_bf16(const float* src, void* dst, int64_t n, ocn_stream_t stream) {
    OCN_CHECK_ARG(src && dst && n > 0, "ocn_cast_f32_bf16: bad arguments");
    OCN_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, "ocn_cast_f32_bf16: misaligned");
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, (long)n);
    OCN_CHECK_LAUNCH("ocn_cast_f32_bf16");
    return OCN_OK;
}

extern "C" int ocn_cast_f32_bf16_scaled(const float* src, void* dst, int64_t n, const float* scale_dev, ocn_stream_t stream) {
    OCN_CHECK_ARG(src && dst && n > 0 && scale_dev, "ocn_cast_f32_bf16_scaled: bad arguments");
    OCN_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, "ocn_cast_f32_bf16_scaled: misaligned");
    hipLaunchKernelGGL(cast_scaled_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, (long)n, scale_dev);
    OCN_CHECK_LAUNCH("ocn_cast_f32_bf16_scaled");
    return OCN_OK;
}

extern "C" int ocn_cast_transpose_f32_bf16(const float* src, void* dst, int R, int C, ocn_stream_t stream) {
    OCN_CHECK_ARG(src && dst && R > 0 && C > 0, "ocn_cast_transpose_f32_bf16: bad arguments");
    hipLaunchKernelGGL(cast_transpose_kernel, dim3(ocn_cdiv(C, 64), ocn_cdiv(R, 64)), dim3(256), 0, (hipStream_t)stream, src, (bf16*)dst, R, C);
    OCN_CHECK_LAUNCH("ocn_cast_transpose_f32_bf16");
    return OCN_OK;
}

extern "C" int ocn_sumsq_accum(const float* x, int64_t n, float* out, ocn_stream_t stream) {
    OCN_CHECK_ARG(x && out && n > 0, "ocn_sumsq_accum: bad arguments");
    OCN_CHECK_ARG(((uintptr_t)x & 15) == 0, "ocn_sumsq_accum: misaligned");
    int g = grid_for((n + 3) / 4, 256);
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(sumsq_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, (long)n, out);
    OCN_CHECK_LAUNCH("ocn_sumsq_accum");
    return OCN_OK;
}

extern "C" int ocn_adamw_step(float* w, const float* g, float* m, float* v, void* w_bf16, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, const float* clip_coef,
                              ocn_stream_t stream) {
    OCN_CHECK_ARG(w && g && m && v && n > 0 && step >= 1, "ocn_adamw_step: bad arguments");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, (bf16*)w_bf16, (long)n,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, clip_coef);
    OCN_CHECK_LAUNCH("ocn_adamw_step");
    return OCN_OK;
}

extern "C" int ocn_adamw_multi(const void* entries, const void* chunks, int n_chunks, float beta1, float beta2, float eps,
                               const float* gnorm_sq, float max_norm, ocn_stream_t stream) {
    OCN_CHECK_ARG(entries && chunks && n_chunks > 0, "ocn_adamw_multi: bad arguments");
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const AdamEntry*)entries,
                       (const int2*)chunks, beta1, beta2, eps, gnorm_sq, max_norm);
    OCN_CHECK_LAUNCH("ocn_adamw_multi");
    return OCN_OK;
}

extern "C" int ocn_sumsq_multi(const void* entries, const void* chunks, int n_chunks, float* out, float* chunk_ws, ocn_stream_t stream) {
    OCN_CHECK_ARG(entries && chunks && out && n_chunks > 0, "ocn_sumsq_multi: bad arguments");
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const AdamEntry*)entries,
                       (const int2*)chunks, out, chunk_ws);
    OCN_CHECK_LAUNCH("ocn_sumsq_multi");
    if (chunk_ws) {
        hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, chunk_ws, n_chunks, out);
        OCN_CHECK_LAUNCH("ocn_sumsq_multi (fold)");
    }
    return OCN_OK;
}
