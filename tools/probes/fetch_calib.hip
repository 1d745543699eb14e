// Developer probe: calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns the hot kernels use.
// Every kernel moves EXACTLY `bytes` (1 GiB, far beyond the 256 MiB Infinity Cache, every byte touched once) with one pattern:
//   read16   16 B per lane, a wave covers 1024 contiguous bytes            (GEMM operand DMA, LayerNorm rows)
//   read8r   8 B per lane in 64-byte row segments: 8 lanes x 8 B contiguous, rows 6144 B apart   (the bf16 gelu' loads of round 2)
//   read4r   4 B per lane in 32-byte row segments: 8 lanes x 4 B contiguous, rows 3072 B apart   (the 8-bit gelu' loads of round 3)
//   read16r  16 B per lane in 128-byte row segments, rows 3072 B apart     (the fp32 residual loads of the epilogue)
//   write16 / write8r / write16r   the same shapes as stores
// Run under  rocprofv3 --pmc FETCH_SIZE  and  --pmc WRITE_SIZE  (separate passes) and compare the counters with `bytes`:
// tools/pmc_calib.py prints counter / bytes per kernel = the factor to apply to that pattern.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/fetch_calib_bin tools/probes/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// contiguous: lane l of global wave w reads chunk (w * 64 + l)
template <typename T>
__global__ __launch_bounds__(256) void k_read_contig(const T* __restrict__ p, float* __restrict__ sink, size_t n) {
    T acc = {};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += __builtin_nontemporal_load(p + i);
    if (acc[0] == 12345.678f) sink[0] = acc[0];
}
// row segments: the matrix is [rows][row_bytes]; a wave instruction touches 8 rows x (8 lanes x sizeof(T)) bytes of one column block, the
// way a 32-column block of a GEMM epilogue does; consecutive instructions of a wave walk the column blocks of its 8 rows
template <typename T>
__global__ __launch_bounds__(256) void k_read_rows(const char* __restrict__ base, float* __restrict__ sink, int rows, int row_bytes) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const int seg = 8 * sizeof(T), nseg = row_bytes / seg;
    T acc = {};
    for (int r8 = wave; r8 < rows / 8; r8 += nwaves) {
        const char* rp = base + (size_t)(r8 * 8 + (lane >> 3)) * row_bytes + (lane & 7) * sizeof(T);
        for (int s = 0; s < nseg; ++s) acc += *(const T*)(rp + (size_t)s * seg);
    }
    if (acc[0] == 12345.678f) sink[0] = acc[0];
}
template <typename T>
__global__ __launch_bounds__(256) void k_write_contig(T* __restrict__ p, size_t n) {
    T v = {};
    v[0] = 1.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
template <typename T>
__global__ __launch_bounds__(256) void k_write_rows(char* __restrict__ base, int rows, int row_bytes) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const int seg = 8 * sizeof(T), nseg = row_bytes / seg;
    T v = {};
    v[0] = 1.0f;
    for (int r8 = wave; r8 < rows / 8; r8 += nwaves) {
        char* rp = base + (size_t)(r8 * 8 + (lane >> 3)) * row_bytes + (lane & 7) * sizeof(T);
        for (int s = 0; s < nseg; ++s) *(T*)(rp + (size_t)s * seg) = v;
    }
}
struct f32x1 {
    float x;
    __device__ float& operator[](int) { return x; }
    __device__ f32x1& operator+=(const f32x1& o) { x += o.x; return *this; }
};
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
int main() {
    const size_t bytes = 1ull << 30;
    char* buf; float* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, bytes));
    const int grid = 256 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char* name, auto launch) {
        launch();  // warm (and the launch the counters see first)
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-10s %zu bytes  %.3f ms  %.2f TB/s\n", name, bytes, ms, bytes / ms / 1e9);
    };
    timed("read16", [&] { k_read_contig<f32x4><<<grid, 256>>>((const f32x4*)buf, sink, bytes / 16); });
    timed("read8r", [&] { k_read_rows<f32x2><<<grid, 256>>>(buf, sink, (int)(bytes / 6144), 6144); });
    timed("read4r", [&] { k_read_rows<f32x1><<<grid, 256>>>(buf, sink, (int)(bytes / 3072), 3072); });
    timed("read16r", [&] { k_read_rows<f32x4><<<grid, 256>>>(buf, sink, (int)(bytes / 3072), 3072); });
    timed("write16", [&] { k_write_contig<f32x4><<<grid, 256>>>((f32x4*)buf, bytes / 16); });
    timed("write8r", [&] { k_write_rows<f32x2><<<grid, 256>>>(buf, (int)(bytes / 6144), 6144); });
    timed("write16r", [&] { k_write_rows<f32x4><<<grid, 256>>>(buf, (int)(bytes / 3072), 3072); });
    CK(hipDeviceSynchronize());
    return 0;
}
