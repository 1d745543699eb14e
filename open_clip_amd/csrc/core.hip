// Error plumbing, version, and the two hardware-layout probes the tests use.
#include <mutex>
#include <unordered_map>
#include "gemm_args.h"

#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void ocn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ocn_last_error(void) { return g_err; }
extern "C" int ocn_version(void) { return OCN_ABI_VERSION; }

// developer tuning / ablation knobs (process-global; timing experiments only)
int g_ocn_tuning[16] = {0};
extern "C" int ocn_set_tuning(int key, int value) {
    OCN_CHECK_ARG(key >= 0 && key < 16, "ocn_set_tuning: key %d out of range", key);
    g_ocn_tuning[key] = value;
    return OCN_OK;
}

// ---- tile rescue (multi-GPU form of the persistent GEMMs; gemm_nt5.hip / gemm_tn5.hip) ----------------------------------------------------------
// Off: every workgroup computes its static share and nothing else (the single-GPU kernels, untouched).  On: the launches carry a board of counters
// through which workgroups that finish hand out the shares of workgroups that have not started (their CU is held by a collective's kernel).
// Boards come from a per-stream ring (below).
int g_ocn_tile_rescue = 0;
extern "C" int ocn_set_tile_rescue(int on) {
    g_ocn_tile_rescue = on ? 1 : 0;
    return OCN_OK;
}
extern "C" int ocn_get_tile_rescue(void) { return g_ocn_tile_rescue; }

int* ocn_rescue_board(hipStream_t st, int workgroups) {
    // A ring of OCN_RESCUE_RING boards (2 MiB) per stream, one per launch, zeroed all at once (stream-ordered) every time the ring wraps: no launch has to clean
    // up behind itself (a "last one out" counter costs 256 same-address atomics at the moment every workgroup finishes).
    if (!g_ocn_tile_rescue || workgroups > OCN_RESCUE_SLOTS) return nullptr;
    // a launch recorded into a graph would meet ITS board again at every replay, with the owners' marks of the first run still on it (every share would
    // look taken): captured launches get the static form
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    struct Ring { int* base; unsigned next; };
    static std::mutex mu;
    static std::unordered_map<uint64_t, Ring> rings;  // per (device, stream): the null stream has the same handle on every device
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t key = (uint64_t)(uintptr_t)st ^ ((uint64_t)(dev + 1) << 56);
    std::lock_guard<std::mutex> lock(mu);
    auto it = rings.find(key);
    if (it == rings.end()) {
        int* b = nullptr;
        if (hipMalloc((void**)&b, (size_t)OCN_RESCUE_RING * OCN_RESCUE_SLOTS * OCN_RESCUE_STRIDE * sizeof(int)) != hipSuccess) return nullptr;
        it = rings.emplace(key, Ring{b, 0u}).first;
    }
    Ring& r = it->second;
    const unsigned slot = r.next++ % OCN_RESCUE_RING;
    if (slot == 0 && hipMemsetAsync(r.base, 0, (size_t)OCN_RESCUE_RING * OCN_RESCUE_SLOTS * OCN_RESCUE_STRIDE * sizeof(int), st) != hipSuccess) return nullptr;
    return r.base + (size_t)slot * OCN_RESCUE_SLOTS * OCN_RESCUE_STRIDE;
}

namespace {
// C[32,32] = A[32,16] . B[32,16]^T with ONE v_mfma_f32_32x32x16_bf16 -- pins operand/accumulator lane maps
__global__ void probe_mfma32_kernel(const bf16* a, const bf16* b, float* c) {
    const int lane = threadIdx.x;
    const bf16x8 af = *(const bf16x8*)(a + (lane & 31) * 16 + (lane >> 5) * 8);
    const bf16x8 bq = *(const bf16x8*)(b + (lane & 31) * 16 + (lane >> 5) * 8);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = mfma32(af, bq, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) c[mfma32_row(r, lane) * 32 + (lane & 31)] = acc[r];
}

// in: [16 rows][32 cols] bf16.  16-lane group g=(lane>>4)&1, half h=lane>>5: reads the 4x16 block at rows
// 4*h*2.. (rows h*8 + 0..3), cols g*16.. ; lane i gives the address of [i>>2][(i&3)*4] and must get column i.
__global__ void probe_tr16_kernel(const bf16* in, bf16* out) {
    __shared__ __attribute__((aligned(16))) bf16 s[16 * 32];
    const int lane = threadIdx.x;
    for (int k = lane; k < 16 * 32; k += 64) s[k] = in[k];
    __syncthreads();
    const int i = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
    const bf16* p = s + (h * 8 + (i >> 2)) * 32 + g * 16 + (i & 3) * 4;
    const s16x4 v = lds_read_tr16((const OCN_LDS void*)p);
    *(s16x4*)(out + lane * 4) = v;
}
}  // namespace

namespace {
// developer probe: `n` workgroups that each hold a CU's LDS for `ticks` (100 MHz) -- stands in for collective kernels that occupy
// CUs while a persistent GEMM is launched (tools/occupancy_hazard_probe.py)
__global__ void occupy_kernel(long long ticks, int* sink) {
    extern __shared__ char smem[];
    const long long until = wall_clock64() + ticks;
    int acc = 0;
    while (wall_clock64() < until) {
        __builtin_amdgcn_s_sleep(32);
        acc += smem[threadIdx.x];
    }
    if (acc == 0x7fffffff) *sink = acc;
}
}  // namespace

extern "C" int ocn_debug_occupy(int n_workgroups, int micros, int* sink, ocn_stream_t stream) {
    OCN_CHECK_ARG(n_workgroups > 0 && micros > 0 && sink, "ocn_debug_occupy: bad arguments");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(occupy_kernel, dim3(n_workgroups), dim3(64), 140 * 1024, (hipStream_t)stream, (long long)micros * 100, sink);
    OCN_CHECK_LAUNCH("ocn_debug_occupy");
    return OCN_OK;
}

// developer probe: a HIP stream whose kernels may only run on the CUs named in `mask` (bit i of word i/32 = CU i); tools/cu_mask_probe.py
extern "C" int ocn_debug_stream_with_cu_mask(const uint32_t* mask, int nwords, void** stream_out) {
    OCN_CHECK_ARG(mask && nwords > 0 && stream_out, "ocn_debug_stream_with_cu_mask: bad arguments");
    hipStream_t st = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask);
    if (e != hipSuccess) {
        ocn_set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
        return OCN_ERR_LAUNCH;
    }
    *stream_out = (void*)st;
    return OCN_OK;
}

extern "C" int ocn_probe_mfma32(const void* a, const void* b, float* c, ocn_stream_t stream) {
    hipLaunchKernelGGL(probe_mfma32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b, c);
    OCN_CHECK_LAUNCH("ocn_probe_mfma32");
    return OCN_OK;
}
extern "C" int ocn_probe_tr16(const void* in, void* out, ocn_stream_t stream) {
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16*)in, (bf16*)out);
    OCN_CHECK_LAUNCH("ocn_probe_tr16");
    return OCN_OK;
}
