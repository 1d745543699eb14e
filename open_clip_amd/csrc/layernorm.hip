// LayerNorm forward / backward over the last dimension (HBM-bound; fp32 statistics).
// One wave per row, 16-byte vector accesses, rows grid-strided; the backward keeps per-lane partial sums
// of dgamma/dbeta in registers across its rows, folds the 16 waves of a workgroup (one workgroup per CU) through LDS and issues
// one fp32 atomic per column per workgroup.
// Algorithmic bytes per row of C columns: fwd 4C (x fp32) + 2C (y bf16) [+4C if y fp32];
// bwd 2C|4C (dy) + 4C (x) [+4C dres] + 4C (dx fp32) + 2C (dx bf16).
// Round 6: x (and, in the backward, the residual gradient) may be bf16 -- the image tower's residual stream as the reference's autocast
// runs it (transformer.py:794 conv1 under autocast -> bf16, layers.py:23-26 casts LayerNorm's result back to the input dtype): fwd 2C + 2C,
// bwd 2C (dy) + 2C (x) + 2C (dres) + 2C (dx bf16) = 8C instead of 16C.  Statistics and arithmetic stay fp32.
#include "ocn_common.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

extern int g_ocn_tuning[16];
namespace {

// 4 consecutive elements of a row as fp32: from fp32 (16 bytes) or bf16 (8 bytes) memory; NT = non-temporal policy
template <bool B16, bool NT>
OCN_DEV f32x4 ld4(const void* base, size_t idx) {
    if constexpr (B16) {
        const bf16x4 v = NT ? __builtin_nontemporal_load((const bf16x4*)((const bf16*)base + idx)) : *(const bf16x4*)((const bf16*)base + idx);
        return (f32x4){bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
    } else {
        return NT ? __builtin_nontemporal_load((const f32x4*)((const float*)base + idx)) : *(const f32x4*)((const float*)base + idx);
    }
}

// NTX: x (read again only by the backward, a whole forward later) is loaded with the non-temporal policy: 100 -> 89 us on the packed
// text rows, nothing on the image rows (profiles/r03_layernorm_forward_variants.txt; fetching gamma / beta once per wave before the row
// loop instead of per row was 3-10 % slower there and is gone).
template <int NV, bool NTX, bool X16>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ b, bf16* __restrict__ y16,
                                                      float* __restrict__ y32, float* __restrict__ mean,
                                                      float* __restrict__ rstd, int M, int C, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float invC = 1.0f / (float)C;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        f32x4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < C) v[i] = ld4<X16, NTX>(x, (size_t)row * C + c);
            else v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
        const float mu = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < C) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[i][e] - mu;
                    q += d * d;
                }
            }
        }
        const float rs = rsqrtf(wave_sum(q) * invC + eps);
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < C) {
                const f32x4 wv = *(const f32x4*)(w + c), bv = *(const f32x4*)(b + c);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * wv[e] + bv[e];
                if (y32) *(f32x4*)(y32 + (size_t)row * C + c) = o;
                if (y16) {
                    bf16x4 o4 = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
                    *(bf16x4*)(y16 + (size_t)row * C + c) = o4;
                }
            }
        }
    }
}

// NT: x and the residual gradient (read once, long after they were written) are loaded and the fp32 dx (read again only by the
// next LayerNorm backward, two GEMMs later) is stored with the non-temporal policy; dy and the bf16 dx, which the neighbouring
// GEMMs produce / consume right away, keep the default one.
// Workgroups of BW = 16 waves (12 for C <= 1280, 8 beyond: registers), ONE per CU, rows grid-strided: the dgamma / dbeta partials of a workgroup's waves
// are folded through LDS (a tree, BW / 2 slabs) and leave as one fp32 atomic per column per WORKGROUP -- with 2048 four-wave workgroups
// the 2048-way contended atomics were 0.06-0.17 ms of a 0.37-0.52 ms launch (profiles/r02_layernorm_grid.txt).
template <int NV>
constexpr int ln_bwd_waves() { return NV <= 3 ? 16 : (NV <= 5 ? 12 : 8); }  // what the registers of a row's NV x 16 bytes per lane leave room for

// CS: additionally dcol[c] += sum over rows of dx[row, c] in fp32, BEFORE dx is rounded to its bf16 twin.  dx is the gradient of the output of the
// linear layer in front of this LayerNorm's input (out_proj resp. the previous block's c_proj), so this IS that layer's bias gradient
// (transformer.py:246, :299) -- summed from fp32 values instead of from the bf16 operand of the weight-gradient GEMM: bias gradients are column
// sums with heavy cancellation across a contrastive batch, and at batch 4096 the rounding of the summands alone put them at 1.0 of their
// parity bound (profiles/r04_parity_report.txt).
// X16 / DR16: x resp. the residual gradient are bf16 (the image tower's bf16 residual stream, see the header)
template <int NV, bool DY32, bool NT, bool CS, bool X16 = false, bool DR16 = false>
__global__ __launch_bounds__(ln_bwd_waves<NV>() * 64) void ln_bwd_kernel(const void* __restrict__ dyv, const void* __restrict__ x,
                                                      const float* __restrict__ w, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const void* __restrict__ dres,
                                                      float* __restrict__ dx32, bf16* __restrict__ dx16,
                                                      float* __restrict__ dw, float* __restrict__ db, float* __restrict__ dcol,
                                                      float* __restrict__ det_ws, int M, int C) {
    constexpr int BW = ln_bwd_waves<NV>();
    constexpr int NS = CS ? 3 : 2;  // sums per column: dgamma, dbeta (, dx)
    __shared__ float red[(BW + 1) / 2][NV * 256 * NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float invC = 1.0f / (float)C;
    f32x4 aw[NV], ab[NV], ac[CS ? NV : 1], wv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        aw[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        ab[i] = aw[i];
        if (CS) ac[i] = aw[i];
        wv[i] = (c < C) ? *(const f32x4*)(w + c) : aw[i];
    }
    for (int row = blockIdx.x * BW + wave; row < M; row += gridDim.x * BW) {
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[NV], g[NV], dr[NV];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            xh[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            g[i] = xh[i];
            dr[i] = xh[i];
            if (c < C) {
                // the residual gradient is fetched together with x and dy (not after the row reductions): one memory
                // round trip per row instead of two
                if (dres) dr[i] = ld4<DR16, NT>(dres, (size_t)row * C + c);
                const f32x4 xv = ld4<X16, NT>(x, (size_t)row * C + c);
                f32x4 dy;
                if (DY32) {
                    dy = *(const f32x4*)((const float*)dyv + (size_t)row * C + c);
                } else {
                    const bf16x4 d4 = *(const bf16x4*)((const bf16*)dyv + (size_t)row * C + c);
                    dy = (f32x4){bf2f(d4[0]), bf2f(d4[1]), bf2f(d4[2]), bf2f(d4[3])};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (xv[e] - mu) * rs;
                    g[i][e] = dy[e] * wv[i][e];
                    c1 += g[i][e];
                    c2 += g[i][e] * xh[i][e];
                    aw[i][e] += dy[e] * xh[i][e];
                    ab[i][e] += dy[e];
                }
            }
        }
        c1 = wave_sum(c1) * invC;
        c2 = wave_sum(c2) * invC;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < C) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[i][e] - c1 - xh[i][e] * c2);
                o = o + dr[i];
                if (CS) ac[i] = ac[i] + o;
                if (dx32) {
                    if (NT) __builtin_nontemporal_store(o, (f32x4*)(dx32 + (size_t)row * C + c));
                    else *(f32x4*)(dx32 + (size_t)row * C + c) = o;
                }
                if (dx16) {
                    bf16x4 o4 = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
                    *(bf16x4*)(dx16 + (size_t)row * C + c) = o4;
                }
            }
        }
    }
    // fold the BW waves pairwise through LDS (upper half writes, lower half adds), then one atomic per column per workgroup
#pragma unroll
    for (int n = BW; n > 1; n = (n + 1) / 2) {  // n live partials -> ceil(n / 2)
        const int half = (n + 1) / 2;
        if (wave >= half && wave < n) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    red[wave - half][((i * 4 + e) * 64 + lane) * NS] = aw[i][e];
                    red[wave - half][((i * 4 + e) * 64 + lane) * NS + 1] = ab[i][e];
                    if (CS) red[wave - half][((i * 4 + e) * 64 + lane) * NS + 2] = ac[i][e];
                }
        }
        __syncthreads();
        if (wave < n - half) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    aw[i][e] += red[wave][((i * 4 + e) * 64 + lane) * NS];
                    ab[i][e] += red[wave][((i * 4 + e) * 64 + lane) * NS + 1];
                    if (CS) ac[i][e] += red[wave][((i * 4 + e) * 64 + lane) * NS + 2];
                }
        }
        __syncthreads();
    }
    if (wave == 0) {
        // det_ws (NativeCLIP(deterministic=True)): this workgroup's partial rows go to its own slab [gridDim.x][3][C]; ln_bwd_finish_kernel adds the
        // slabs up in workgroup order -- no two runs can differ in the order of the fp32 additions
        float* slab = det_ws ? det_ws + (size_t)blockIdx.x * 3 * C : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (c < C) {
                    if (slab) {
                        slab[c + e] = aw[i][e];
                        slab[C + c + e] = ab[i][e];
                        slab[2 * C + c + e] = CS ? ac[CS ? i : 0][e] : 0.f;
                    } else {
                        unsafeAtomicAdd(dw + c + e, aw[i][e]);
                        unsafeAtomicAdd(db + c + e, ab[i][e]);
                        if (CS) unsafeAtomicAdd(dcol + c + e, ac[i][e]);
                    }
                }
            }
        }
    }
}

// ---- bf16 residual stream (round 6): software-pipelined forms ---------------------------------------------------------------------------
// With a bf16 x a row is 8 bytes per lane and access: the one-row-at-a-time kernels above keep half the bytes in flight that they do on an fp32
// stream and stop being bandwidth-bound (measured on the image tower's 204 800 x 768 rows: forward 147 us = the fp32 stream's time at 2/3 of the
// bytes; backward 374 us = 3.4 TB/s).  Here a wave requests row i + 1 (still PACKED: 2 dwords per tensor and 4 columns) before it touches row i,
// so two rows per wave are in flight at all times.  vmcnt retires in issue order: gamma / beta of the forward (L2 hits) are requested BEFORE the
// next row so that their wait does not drain it; mean / rstd of the backward come through the scalar cache (wave-uniform row).
template <int NV, bool Y16, bool Y32>
__global__ __launch_bounds__(256) void ln_fwd16_kernel(const bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                        bf16* __restrict__ y16, float* __restrict__ y32, float* __restrict__ mean,
                                                        float* __restrict__ rstd, int M, int C, float eps) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float invC = 1.0f / (float)C;
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    if (row >= M) return;
    f32x4 wv[NV], bv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        wv[i] = *(const f32x4*)(w + (i * 64 + lane) * 4);
        bv[i] = *(const f32x4*)(b + (i * 64 + lane) * 4);
    }
    bf16x4 nx[2][NV];  // two rows in flight (packed); a row index beyond M re-reads row M - 1 (never used): the loop body stays one basic block
    auto fetch = [&](int slot, int r) {
        r = r < M ? r : M - 1;
#pragma unroll
        for (int i = 0; i < NV; ++i) nx[slot][i] = *(const bf16x4*)(x + (size_t)r * C + (i * 64 + lane) * 4);
    };
    auto body = [&](int slot, int r) {
        f32x4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = (f32x4){bf2f(nx[slot][i][0]), bf2f(nx[slot][i][1]), bf2f(nx[slot][i][2]), bf2f(nx[slot][i][3])};
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
        fetch(slot, r + 2 * stride);
        const float mu = wave_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mu;
                q += d * d;
            }
        const float rs = rsqrtf(wave_sum(q) * invC + eps);
        if (lane == 0) {
            mean[r] = mu;
            rstd[r] = rs;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mu) * rs * wv[i][e] + bv[i][e];
            if (Y32) *(f32x4*)(y32 + (size_t)r * C + c) = o;
            if (Y16) {
                bf16x4 o4 = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
                *(bf16x4*)(y16 + (size_t)r * C + c) = o4;
            }
        }
    };
    fetch(0, row);
    fetch(1, row + stride);
    for (; row < M; row += 2 * stride) {
        body(0, row);
        if (row + stride < M) body(1, row + stride);
    }
}

// DR: the residual gradient -- 0 none, 1 bf16, 2 fp32 (NativeCLIP(image_stream="bf16-fp32grad")); dx32 optional (DX32), dx16 always
template <int NV>
constexpr int ln_bwd16_waves() { return NV <= 2 ? 16 : (NV == 3 ? 12 : 8); }  // (NV = 5: 172-234 registers at 8 waves, no spills)  // two packed rows per wave cost registers: fewer waves than ln_bwd_waves

template <int NV, int DR, bool DX32>
__global__ __launch_bounds__(ln_bwd16_waves<NV>() * 64) void ln_bwd16_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                      const float* __restrict__ w, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const void* __restrict__ dres,
                                                      float* __restrict__ dx32, bf16* __restrict__ dx16,
                                                      float* __restrict__ dw, float* __restrict__ db, float* __restrict__ det_ws, int M, int C) {
    constexpr int BW = ln_bwd16_waves<NV>();
    __shared__ float red[(BW + 1) / 2][NV * 256 * 2];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float invC = 1.0f / (float)C;
    f32x4 aw[NV], ab[NV], wv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        aw[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        ab[i] = aw[i];
        wv[i] = *(const f32x4*)(w + (i * 64 + lane) * 4);
    }
    const int stride = gridDim.x * BW;
    int row = blockIdx.x * BW + wave;
    bf16x4 nx[2][NV], ny[2][NV], nr16[2][DR == 1 ? NV : 1];
    f32x4 nr32[2][DR == 2 ? NV : 1];
    auto fetch = [&](int slot, int r) {
        r = r < M ? r : M - 1;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const size_t o = (size_t)r * C + (i * 64 + lane) * 4;
            if constexpr (DR == 1) nr16[slot][i] = __builtin_nontemporal_load((const bf16x4*)((const bf16*)dres + o));
            if constexpr (DR == 2) nr32[slot][i] = __builtin_nontemporal_load((const f32x4*)((const float*)dres + o));
            nx[slot][i] = __builtin_nontemporal_load((const bf16x4*)(x + o));
            ny[slot][i] = *(const bf16x4*)(dy + o);
        }
    };
    auto body = [&](int slot, int r) {
        const float mu = mean[r], rs = rstd[r];
        f32x4 xh[NV], g[NV], dr[NV];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)  // unpack the row that has arrived
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dyv = bf2f(ny[slot][i][e]);
                xh[i][e] = (bf2f(nx[slot][i][e]) - mu) * rs;
                g[i][e] = dyv * wv[i][e];
                aw[i][e] += dyv * xh[i][e];
                ab[i][e] += dyv;
                c1 += g[i][e];
                c2 += g[i][e] * xh[i][e];
                if constexpr (DR == 1) dr[i][e] = bf2f(nr16[slot][i][e]);
                else if constexpr (DR == 2) dr[i][e] = nr32[slot][i][e];
                else dr[i][e] = 0.f;
            }
        fetch(slot, r + 2 * stride);  // two rows ahead: the slot is free again
        c1 = wave_sum(c1) * invC;
        c2 = wave_sum(c2) * invC;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (g[i][e] - c1 - xh[i][e] * c2);
            o = o + dr[i];
            if (DX32) __builtin_nontemporal_store(o, (f32x4*)(dx32 + (size_t)r * C + c));
            bf16x4 o4 = {f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
            *(bf16x4*)(dx16 + (size_t)r * C + c) = o4;
        }
    };
    if (row < M) {
        fetch(0, row);
        fetch(1, row + stride);
    }
    for (; row < M; row += 2 * stride) {
        body(0, row);
        if (row + stride < M) body(1, row + stride);
    }
#pragma unroll
    for (int n = BW; n > 1; n = (n + 1) / 2) {  // fold the BW waves pairwise through LDS, as ln_bwd_kernel does
        const int half = (n + 1) / 2;
        if (wave >= half && wave < n) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    red[wave - half][((i * 4 + e) * 64 + lane) * 2] = aw[i][e];
                    red[wave - half][((i * 4 + e) * 64 + lane) * 2 + 1] = ab[i][e];
                }
        }
        __syncthreads();
        if (wave < n - half) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    aw[i][e] += red[wave][((i * 4 + e) * 64 + lane) * 2];
                    ab[i][e] += red[wave][((i * 4 + e) * 64 + lane) * 2 + 1];
                }
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* slab = det_ws ? det_ws + (size_t)blockIdx.x * 3 * C : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (slab) {
                    slab[c + e] = aw[i][e];
                    slab[C + c + e] = ab[i][e];
                    slab[2 * C + c + e] = 0.f;
                } else {
                    unsafeAtomicAdd(dw + c + e, aw[i][e]);
                    unsafeAtomicAdd(db + c + e, ab[i][e]);
                }
            }
        }
    }
}

// second stage of the reproducible form: column c of dw / db / dcol += the G workgroups' partials, added in workgroup order by ONE thread
__global__ __launch_bounds__(256) void ln_bwd_finish_kernel(const float* __restrict__ ws, int G, int C, float* __restrict__ dw, float* __restrict__ db,
                                                            float* __restrict__ dcol) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b = 0.f, d = 0.f;
    for (int g = 0; g < G; ++g) {
        const float* slab = ws + (size_t)g * 3 * C;
        a += slab[c];
        b += slab[C + c];
        d += slab[2 * C + c];
    }
    dw[c] += a;
    db[c] += b;
    if (dcol) dcol[c] += d;
}

int ln_grid(int M) {
    int g = ocn_cdiv(M, 4);
    return g < 2048 ? g : 2048;
}
int g_ln_num_cu = 0;
template <int NV>
int ln_bwd_grid(int M) {
    if (g_ln_num_cu == 0) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_ln_num_cu = n;
    }
    const int g = ocn_cdiv(M, ln_bwd_waves<NV>());
    // developer knob 14: workgroups of the backward's grid
    const int cap = g_ocn_tuning[14] > 0 ? g_ocn_tuning[14] : g_ln_num_cu;
    return g < cap ? g : cap;
}

template <int NV>
void launch_fwd(hipStream_t st, const void* x, int x_is_bf16, const float* w, const float* b, bf16* y16, float* y32, float* mean, float* rstd,
                int M, int C, float eps) {
    const dim3 g(ln_grid(M)), t(256);
    if (x_is_bf16 && C == NV * 256 && g_ocn_tuning[12] != 2) {  // the pipelined form (developer knob 12 = 2: the one-row-at-a-time kernel on the bf16 stream, A/B)
        if (y16 && y32) ln_fwd16_kernel<NV, true, true><<<g, t, 0, st>>>((const bf16*)x, w, b, y16, y32, mean, rstd, M, C, eps);
        else if (y16) ln_fwd16_kernel<NV, true, false><<<g, t, 0, st>>>((const bf16*)x, w, b, y16, y32, mean, rstd, M, C, eps);
        else ln_fwd16_kernel<NV, false, true><<<g, t, 0, st>>>((const bf16*)x, w, b, y16, y32, mean, rstd, M, C, eps);
    } else if (x_is_bf16)
        ln_fwd_kernel<NV, false, true><<<g, t, 0, st>>>(x, w, b, y16, y32, mean, rstd, M, C, eps);
    else if (g_ocn_tuning[12] == 1)  // developer knob 12 = 1: default cache policy for x
        ln_fwd_kernel<NV, false, false><<<g, t, 0, st>>>(x, w, b, y16, y32, mean, rstd, M, C, eps);
    else
        ln_fwd_kernel<NV, true, false><<<g, t, 0, st>>>(x, w, b, y16, y32, mean, rstd, M, C, eps);
}
template <int NV, bool DY32, bool NT, bool CS, bool X16 = false, bool DR16 = false>
void launch_bwd4(hipStream_t st, const void* dy, const void* x, const float* w, const float* mean, const float* rstd, const void* dres, float* dx32,
                 bf16* dx16, float* dw, float* db, float* dcol, float* det_ws, int M, int C) {
    const int G = ln_bwd_grid<NV>(M);
    ln_bwd_kernel<NV, DY32, NT, CS, X16, DR16><<<dim3(G), dim3(ln_bwd_waves<NV>() * 64), 0, st>>>(dy, x, w, mean, rstd, dres, dx32, dx16, dw, db, dcol, det_ws, M, C);
    if (det_ws) ln_bwd_finish_kernel<<<dim3(ocn_cdiv(C, 256)), dim3(256), 0, st>>>(det_ws, G, C, dw, db, dcol);
}
// returns false for a dtype combination that has no instantiation (the caller reports it)
template <int NV>
bool launch_bwd(hipStream_t st, const void* dy, int dy_is_f32, const void* x, int x_is_bf16, const float* w, const float* mean,
                const float* rstd, const void* dres, int dres_is_bf16, float* dx32, bf16* dx16, float* dw, float* db, float* dcol, float* det_ws, int M, int C) {
    if (x_is_bf16) {
        // the bf16 residual stream (image tower): dy is always the bf16 output of a dgrad GEMM there; the residual gradient is bf16 (the
        // reference's autograd: the gradient of a bf16 tensor is bf16) or the fp32 companion (NativeCLIP(image_stream="bf16-fp32grad"))
        if (dy_is_f32 || dcol) return false;
        // the pipelined form up to C = 1280 (ViT-H-14; 8 waves of up to 256 registers there; beyond, the second row's registers spill; developer knob
        // 8 = 2: the one-row-at-a-time kernel, A/B)
        if constexpr (NV <= 5) if (dx16 && C == NV * 256 && g_ocn_tuning[8] != 2) {
            const int G = ln_bwd_grid<NV>(M);  // (the reproducible form's workspace is sized for this grid; rows are grid-strided)
            const dim3 g(G), t(ln_bwd16_waves<NV>() * 64);
#define OCN_LN_BWD16(DR)                                                                                                                                   \
    {                                                                                                                                                     \
        if (dx32) ln_bwd16_kernel<NV, DR, true><<<g, t, 0, st>>>((const bf16*)dy, (const bf16*)x, w, mean, rstd, dres, dx32, dx16, dw, db, det_ws, M, C);  \
        else ln_bwd16_kernel<NV, DR, false><<<g, t, 0, st>>>((const bf16*)dy, (const bf16*)x, w, mean, rstd, dres, dx32, dx16, dw, db, det_ws, M, C);      \
    }
            if (!dres) OCN_LN_BWD16(0) else if (dres_is_bf16) OCN_LN_BWD16(1) else OCN_LN_BWD16(2)
#undef OCN_LN_BWD16
            if (det_ws) ln_bwd_finish_kernel<<<dim3(ocn_cdiv(C, 256)), dim3(256), 0, st>>>(det_ws, G, C, dw, db, dcol);
            return true;
        }
        if (dres && dres_is_bf16) launch_bwd4<NV, false, true, false, true, true>(st, dy, x, w, mean, rstd, dres, dx32, dx16, dw, db, dcol, det_ws, M, C);
        else launch_bwd4<NV, false, true, false, true, false>(st, dy, x, w, mean, rstd, dres, dx32, dx16, dw, db, dcol, det_ws, M, C);
        return true;
    }
    if (dres && dres_is_bf16) return false;
    const bool nt = g_ocn_tuning[8] != 1;  // developer knob 8 = 1: default cache policy everywhere
#define OCN_LN_BWD(DY32, NT)                                                                                                  \
    {                                                                                                                         \
        if (dcol) launch_bwd4<NV, DY32, NT, true>(st, dy, x, w, mean, rstd, dres, dx32, dx16, dw, db, dcol, det_ws, M, C);     \
        else launch_bwd4<NV, DY32, NT, false>(st, dy, x, w, mean, rstd, dres, dx32, dx16, dw, db, dcol, det_ws, M, C);         \
    }
    if (dy_is_f32) {
        if (nt) OCN_LN_BWD(true, true) else OCN_LN_BWD(true, false)
    } else {
        if (nt) OCN_LN_BWD(false, true) else OCN_LN_BWD(false, false)
    }
#undef OCN_LN_BWD
    return true;
}

// out[c] += sum over rows of x[r, c] (fp32): the bias gradients of the B pooled rows' linears (model.py::_PooledBlockFn), summed from fp32 values
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int C) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float s = 0.f;
    if (col < C)
        for (int r = blockIdx.y * 4 + grp; r < R; r += gridDim.y * 4) s += x[(size_t)r * C + col];
    red[grp][threadIdx.x & 63] = s;
    __syncthreads();
    if (grp == 0 && col < C) unsafeAtomicAdd(out + col, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

}  // namespace

extern "C" int ocn_layernorm_fwd(const void* x, int x_is_bf16, const float* w, const float* b, void* y_bf16, float* y_f32, float* mean,
                                 float* rstd, int M, int C, float eps, ocn_stream_t stream) {
    OCN_CHECK_ARG(x && w && b && mean && rstd && (y_bf16 || y_f32), "ocn_layernorm_fwd: null operand");
    OCN_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0 && C <= 2048, "ocn_layernorm_fwd: bad shape M=%d C=%d", M, C);
    hipStream_t st = (hipStream_t)stream;
    bf16* y16 = (bf16*)y_bf16;
    switch (ocn_cdiv(C, 256)) {
        case 1: launch_fwd<1>(st, x, x_is_bf16, w, b, y16, y_f32, mean, rstd, M, C, eps); break;
        case 2: launch_fwd<2>(st, x, x_is_bf16, w, b, y16, y_f32, mean, rstd, M, C, eps); break;
        case 3: launch_fwd<3>(st, x, x_is_bf16, w, b, y16, y_f32, mean, rstd, M, C, eps); break;
        case 4: launch_fwd<4>(st, x, x_is_bf16, w, b, y16, y_f32, mean, rstd, M, C, eps); break;
        case 5: launch_fwd<5>(st, x, x_is_bf16, w, b, y16, y_f32, mean, rstd, M, C, eps); break;
        default: launch_fwd<8>(st, x, x_is_bf16, w, b, y16, y_f32, mean, rstd, M, C, eps); break;
    }
    OCN_CHECK_LAUNCH("ocn_layernorm_fwd");
    return OCN_OK;
}

// floats of workspace the reproducible form of ocn_layernorm_bwd needs: one [3][C] slab per workgroup of its grid
extern "C" int64_t ocn_layernorm_bwd_det_workspace_floats(int M, int C) {
    int g = 0;
    switch (ocn_cdiv(C, 256)) {
        case 1: g = ln_bwd_grid<1>(M); break;
        case 2: g = ln_bwd_grid<2>(M); break;
        case 3: g = ln_bwd_grid<3>(M); break;
        case 4: g = ln_bwd_grid<4>(M); break;
        case 5: g = ln_bwd_grid<5>(M); break;
        default: g = ln_bwd_grid<8>(M); break;
    }
    return (int64_t)g * 3 * C;
}

extern "C" int ocn_colsum_f32(const float* x, float* out, int R, int C, int deterministic, ocn_stream_t stream) {
    OCN_CHECK_ARG(x && out && R > 0 && C > 0, "ocn_colsum_f32: bad operand (R=%d C=%d)", R, C);
    // deterministic: ONE row slice -- a column is summed by four threads over fixed row subsets and folded in a fixed order: no atomic ever meets another
    const int slices = deterministic ? 1 : (R >= 1024 ? 32 : (R >= 64 ? 8 : 1));
    colsum_f32_kernel<<<dim3(ocn_cdiv(C, 64), slices), dim3(256), 0, (hipStream_t)stream>>>(x, out, R, C);
    OCN_CHECK_LAUNCH("ocn_colsum_f32");
    return OCN_OK;
}

extern "C" int ocn_layernorm_bwd(const void* dy, int dy_is_f32, const void* x, int x_is_bf16, const float* w, const float* mean,
                                 const float* rstd, const void* dres, int dres_is_bf16, float* dx_f32, void* dx_bf16, float* dw, float* db, float* dcol,
                                 float* det_workspace, int M, int C, ocn_stream_t stream) {
    OCN_CHECK_ARG(dy && x && w && mean && rstd && dw && db && (dx_f32 || dx_bf16), "ocn_layernorm_bwd: null operand");
    OCN_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0 && C <= 2048, "ocn_layernorm_bwd: bad shape M=%d C=%d", M, C);
    hipStream_t st = (hipStream_t)stream;
    bf16* dx16 = (bf16*)dx_bf16;
    bool ok = true;
    switch (ocn_cdiv(C, 256)) {
        case 1: ok = launch_bwd<1>(st, dy, dy_is_f32, x, x_is_bf16, w, mean, rstd, dres, dres_is_bf16, dx_f32, dx16, dw, db, dcol, det_workspace, M, C); break;
        case 2: ok = launch_bwd<2>(st, dy, dy_is_f32, x, x_is_bf16, w, mean, rstd, dres, dres_is_bf16, dx_f32, dx16, dw, db, dcol, det_workspace, M, C); break;
        case 3: ok = launch_bwd<3>(st, dy, dy_is_f32, x, x_is_bf16, w, mean, rstd, dres, dres_is_bf16, dx_f32, dx16, dw, db, dcol, det_workspace, M, C); break;
        case 4: ok = launch_bwd<4>(st, dy, dy_is_f32, x, x_is_bf16, w, mean, rstd, dres, dres_is_bf16, dx_f32, dx16, dw, db, dcol, det_workspace, M, C); break;
        case 5: ok = launch_bwd<5>(st, dy, dy_is_f32, x, x_is_bf16, w, mean, rstd, dres, dres_is_bf16, dx_f32, dx16, dw, db, dcol, det_workspace, M, C); break;
        default: ok = launch_bwd<8>(st, dy, dy_is_f32, x, x_is_bf16, w, mean, rstd, dres, dres_is_bf16, dx_f32, dx16, dw, db, dcol, det_workspace, M, C); break;
    }
    OCN_CHECK_ARG(ok, "ocn_layernorm_bwd: unsupported dtype combination (bf16 x needs bf16 dy and no dcol; a bf16 residual gradient needs bf16 x)");
    OCN_CHECK_LAUNCH("ocn_layernorm_bwd");
    return OCN_OK;
}
