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
