// Developer probe (not product code): LDS-DMA throughput of the NT-GEMM access pattern without any MFMA work.
//   pattern 0: k32 stages, one wave-instruction = 16 rows x 64 B   (ring kernel as of round 1a)
//   pattern 1: k64 stages, one wave-instruction = 8 rows x 128 B   (full cache lines)
//   pattern 2: k64 unit, lanes 0-31 -> k-half 0 (8 rows x 64 B), lanes 32-63 -> k-half 1
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/dma_probe tools/probes/dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16;
#define LDSP __attribute__((address_space(3)))
#define GLBP __attribute__((address_space(1)))
__device__ __forceinline__ void glds16(const void* g, LDSP void* l) { __builtin_amdgcn_global_load_lds((const GLBP void*)g, l, 16, 0, 0); }
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8; if (nwg < nx) return bid;
    const int xcd = bid % nx, idx = bid / nx, q = nwg / nx, r = nwg % nx;
    return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
// pattern 3: 16 KB half-tile units (128 rows x 128 B; order A0,B0,A1,B1 per k64 tile), `depth` units in flight,
// one barrier per unit; AUX = cache-policy bits of the DMA instruction (0 default, 2 = nt, 1 = sc0, 16 = sc1 ...)
template <int AUX>
__global__ __launch_bounds__(512, 2) void probe_units(const bf16* A, const bf16* B, int lda, int ldb, int M, int N, int K, int tiles_n, int ntiles, int depth, int order, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int u = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int tile = xcd_remap(t, ntiles);
        int tm, tn;
        if (order == 0) { tm = tile / tiles_n; tn = tile % tiles_n; }
        else { const int tiles_m = ntiles / tiles_n; tm = tile % tiles_m; tn = tile / tiles_m; }
        const int m0 = tm * 256, n0 = tn * 256;
        const bf16* p[4];  // A0, B0, A1, B1 sources for this wave: rows wave*16 + j*8 + lane>>3 of the 128-row half
        for (int h = 0; h < 2; ++h) {
            p[2 * h] = A + (size_t)min(m0 + h * 128 + wave * 16 + (lane >> 3), M - 1) * lda + (lane & 7) * 8;
            p[2 * h + 1] = B + (size_t)min(n0 + h * 128 + wave * 16 + (lane >> 3), N - 1) * ldb + (lane & 7) * 8;
        }
        const int nk = K / 64;
        for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                char* dst = smem + (u % 8) * 16384 + wave * 2048;
                const size_t ld8 = (size_t)((q & 1) ? ldb : lda) * 8;
                __builtin_amdgcn_global_load_lds((const GLBP void*)(p[q]), (LDSP void*)dst, 16, 0, AUX);
                __builtin_amdgcn_global_load_lds((const GLBP void*)(p[q] + ld8), (LDSP void*)(dst + 1024), 16, 0, AUX);
                p[q] += 64;
                ++u;
                switch (depth) {
                    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                    case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                }
                __builtin_amdgcn_s_barrier();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[0] = ((float*)smem)[lane];
}
template <int PAT>
__global__ __launch_bounds__(512, 2) void probe(const bf16* A, const bf16* B, int lda, int ldb, int M, int N, int K, int tiles_n, int ntiles, int depth, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tile = xcd_remap(t, ntiles);
        const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
        if (PAT == 0) {
            const bf16* pa[2]; const bf16* pb[2];
            for (int j = 0; j < 2; ++j) {
                const int r = (wave + j * 8) * 16 + (lane >> 2), c = lane & 3;
                pa[j] = A + (size_t)min(m0 + r, M - 1) * lda + c * 8;
                pb[j] = B + (size_t)min(n0 + r, N - 1) * ldb + c * 8;
            }
            const int nk = K / 32;
            for (int kt = 0; kt < nk; ++kt) {
                const int slot = kt & 3;
                for (int j = 0; j < 2; ++j) {
                    glds16(pa[j], (LDSP void*)(smem + slot * 32768 + (wave + j * 8) * 1024)); pa[j] += 32;
                    glds16(pb[j], (LDSP void*)(smem + slot * 32768 + 16384 + (wave + j * 8) * 1024)); pb[j] += 32;
                }
                if (depth == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (depth == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        } else {
            const bf16* pa[4]; const bf16* pb[4];
            for (int j = 0; j < 4; ++j) {
                int r, c;
                if (PAT == 1) { r = (wave + j * 8) * 8 + (lane >> 3); c = lane & 7; }
                else { r = (wave + j * 8) * 8 + ((lane & 31) >> 2); c = (lane & 3) + 4 * (lane >> 5); }
                pa[j] = A + (size_t)min(m0 + r, M - 1) * lda + c * 8;
                pb[j] = B + (size_t)min(n0 + r, N - 1) * ldb + c * 8;
            }
            const int nk = K / 64;
            for (int kt = 0; kt < nk; ++kt) {
                const int slot = kt & 1;
                for (int j = 0; j < 4; ++j) {
                    glds16(pa[j], (LDSP void*)(smem + slot * 65536 + (wave + j * 8) * 1024)); pa[j] += 64;
                    glds16(pb[j], (LDSP void*)(smem + slot * 65536 + 32768 + (wave + j * 8) * 1024)); pb[j] += 64;
                }
                if (depth >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[0] = ((float*)smem)[lane];
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 204800, N = argc > 2 ? atoi(argv[2]) : 768, K = argc > 3 ? atoi(argv[3]) : 3072;
    bf16 *A, *B; float* sink;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&sink, 256);
    hipMemset(A, 0x11, (size_t)M * K * 2); hipMemset(B, 0x22, (size_t)N * K * 2);
    const int tiles_n = (N + 255) / 256, ntiles = ((M + 255) / 256) * tiles_n;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int pat, int depth, int grid) {
        auto launch = [&]() {
            if (pat == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 131072, 0, A, B, K, K, M, N, K, tiles_n, ntiles, depth, sink);
            if (pat == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 131072, 0, A, B, K, K, M, N, K, tiles_n, ntiles, depth, sink);
            if (pat == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(512), 131072, 0, A, B, K, K, M, N, K, tiles_n, ntiles, depth, sink);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double bytes = (double)ntiles * 512.0 * K * 2;  // LDS-bound bytes moved
        printf("pat %d depth %d grid %5d: %.3f ms  %.1f GB/s per CU  (%.2f TB/s into LDS; GEMM at this feed rate = %.0f TF)\n", pat, depth, grid, ms,
               bytes / ms / 1e6 / 256, bytes / ms / 1e9, 2.0 * M * N * K / ms / 1e9);
    };
    hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    printf("M=%d N=%d K=%d ntiles=%d\n", M, N, K, ntiles);
    for (int pat = 0; pat < 3; ++pat)
        for (int depth = 1; depth <= 3; ++depth) {
            if (pat > 0 && depth == 3) continue;
            run(pat, depth, ntiles);
            run(pat, depth, 256);
        }
    hipFuncSetAttribute((const void*)probe_units<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)probe_units<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    auto runu = [&](int aux, int depth, int order) {
        auto launch = [&]() {
            if (aux == 0) hipLaunchKernelGGL(probe_units<0>, dim3(256), dim3(512), 131072, 0, A, B, K, K, M, N, K, tiles_n, ntiles, depth, order, sink);
            else hipLaunchKernelGGL(probe_units<2>, dim3(256), dim3(512), 131072, 0, A, B, K, K, M, N, K, tiles_n, ntiles, depth, order, sink);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double bytes = (double)ntiles * 512.0 * K * 2;
        printf("units aux %d depth %d order %d: %.3f ms  %.1f GB/s per CU  (GEMM at this feed rate = %.0f TF)\n", aux, depth, order, ms, bytes / ms / 1e6 / 256, 2.0 * M * N * K / ms / 1e9);
    };
    for (int aux = 0; aux <= 2; aux += 2)
        for (int order = 0; order < 2; ++order)
            for (int depth : {1, 2, 3, 4, 6, 8}) runu(aux, depth, order);
    return 0;
}
