// Developer probe: sustained v_mfma_f32_32x32x16_bf16 rate on the whole chip with zero vs random operands
// (DVFS: the clock the chip sustains depends on the data), 1 or 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_probe tools/probes/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(512, 2) void k(const bf16x8* in, float* out, int iters, long long* cyc) {
    const int lane = threadIdx.x;
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = in[(lane * 4 + i) % 4096];
    for (int i = 0; i < 2; ++i) b[i] = in[(lane * 2 + i + 1000) % 4096];
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2 & 1], acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    bf16x8* in; float* out; long long* cyc;
    hipMalloc(&in, 4096 * 16); hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 8);
    unsigned short* h = (unsigned short*)malloc(4096 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        for (int i = 0; i < 4096 * 8; ++i) {
            float f = mode ? ((rand() / (float)RAND_MAX) * 2.f - 1.f) : 0.f;
            unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16);
        }
        hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
        for (int threads : {256, 512}) {
            const int iters = 20000;
            hipLaunchKernelGGL(k<8>, dim3(256), dim3(threads), 0, 0, in, out, 100, cyc);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<8>, dim3(256), dim3(threads), 0, 0, in, out, iters, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double nm = (double)iters * 8 * (threads / 64) * 256;
            printf("%s data, %d waves/CU: %.2f ms, %.0f TFLOP/s, %.1f cycles(s_memtime)/MFMA/wave, counter rate %.0f MHz\n", mode ? "random" : "zero",
                   threads / 64, ms, nm * 2 * 32 * 32 * 16 / ms / 1e9, (double)c / (iters * 8.0), c / ms / 1e3);
        }
    }
    return 0;
}
