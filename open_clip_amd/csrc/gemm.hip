// MFMA GEMMs for the CLIP towers on gfx950.
//
//   ocn_gemm_nt        C[M,N]  = A[M,K] . B[N,K]^T  (+ fused epilogue)      forward linears, dgrads, logits
//   ocn_gemm_tn_accum  dW[N,K] += A[M,N]^T . B[M,K] (+ dbias = colsum(A))   wgrads, loss G^T products
//
// The tower-sized launches go to the hand-scheduled persistent kernels (gemm_nt5.hip, gemm_tn5.hip).  This file holds the C entry
// points, the dispatch and ONE general kernel per operation for everything those do not take (small or ragged shapes: pooled
// heads at tiny batch, K % 128 != 0, N % 8 != 0, test configurations):
//   NT  : 256x256 tile, 8 waves, K in steps of 32 through a 4-stage LDS ring (LDS-DMA, counted vmcnt), scalar epilogue for N % 4
//   TN  : 128x128 tile of dW, 4 waves, rows of 128 bf16 (256 B), ds_read_b64_tr_b16 transposing reads, fp32 atomics
// LDS images are XOR-swizzled on the *source* address (the DMA destination is lane-linear).  Tiles are walked in an XCD-aware
// order (ocn_common.h xcd_remap).
#include "gemm_args.h"

extern int g_ocn_tuning[16];

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 16384;            // one operand tile (128x64 or 64x128 bf16)
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // A + B
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;  // double buffered: 64 KiB -> 2 workgroups / CU

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4];  // zero source for out-of-range DMA lanes

template <int EPI>
OCN_DEV void epilogue_store1(const GemmNtArgs& a, int gm, int gn, float v) {
    const size_t o = (size_t)gm * a.ldc + gn;
    const float b = a.bias ? a.bias[gn] : 0.f;
    if (EPI == OCN_EPI_BF16) {
        ((bf16*)a.out)[o] = f2bf(v * a.alpha + b);
    } else if (EPI == OCN_EPI_BIAS_GELU || EPI == OCN_EPI_BIAS_QUICKGELU) {
        float g1, d1;
        act_both<EPI == OCN_EPI_BIAS_QUICKGELU>(v + b, g1, d1);
        a.aux[o] = (unsigned char)(dgelu_q_bits(d1) & 255u);
        ((bf16*)a.out)[o] = f2bf(g1);
    } else if (EPI == OCN_EPI_BIAS_RESID_F32) {
        ((float*)a.out)[o] = v + b + ((const float*)a.resid)[o];
    } else if (EPI == OCN_EPI_BIAS_RESID_BF16) {  // the reference's autocast arithmetic: F.linear's result rounded to bf16, then the bf16 add
        ((bf16*)a.out)[o] = f2bf(bf2f(f2bf(v + b)) + bf2f(((const bf16*)a.resid)[o]));
    } else if (EPI == OCN_EPI_DGELU) {
        ((bf16*)a.out)[o] = f2bf((v + b) * dgelu_unq(a.aux[o]));
    } else {
        ((float*)a.out)[o] = v * a.alpha + b;
    }
}

// Geometry: workgroup tile BM x BN, WR x WC waves, each wave (BM/WR) x (BN/WC) = TI x TJ blocks of 32x32.
//   <128,128,2,2>: 4 waves, 64 KiB ring, 2 workgroups / CU (small problems, ragged edges)
//   <256,256,2,4>: 8 waves, 128 KiB ring, 1 workgroup / CU, wave tile 128x64 (less LDS traffic per MFMA)
// ------------------------------------------------------------------------------------------------
// NT "ring" kernel: 256x256 tile, 8 waves (2x4, wave tile 128x64), BK = 32, 4-stage LDS ring (4 x 32 KiB),
// LDS-DMA prefetch three K-steps ahead with COUNTED vmcnt (never drained in the main loop), one raw
// s_barrier per K-step.  Epilogue operands (residual / saved pre-activation) are prefetched into registers
// one 32-row slab ahead so the epilogue is bandwidth- not latency-bound.
// LDS rows are 64 B (32 bf16): 4 chunks of 16 B, chunk ^= (row>>2)&3 (conflict-free ds_read_b128 fragments).
// ------------------------------------------------------------------------------------------------
OCN_DEV int swz64(int r) { return (r >> 2) & 3; }

template <int ROWS>
OCN_DEV void stage_ring(const bf16* __restrict__ G, int ld, int row0, int nrows, int k0, char* sT, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < ROWS / 16 / 8; ++j) {
        const int seg = wave + j * 8;  // 16 rows (x 64 B) per wave-instruction
        const int r = seg * 16 + (lane >> 2);
        const int c = (lane & 3) ^ swz64(r);
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1;
        glds16(G + (size_t)gr * ld + k0 + c * 8, (OCN_LDS void*)(sT + seg * 1024));
    }
}

template <int EPI>
struct EpiBuf {
    f32x4 r[8];  // residual (EPI 2) or pre-activation as floats (EPI 3) for the 8 row-iterations of a slab
};

// unconditional (address-clamped) loads: straight-line code lets the compiler keep counted vmcnt waits
template <int EPI>
OCN_DEV void epi_prefetch(const GemmNtArgs& a, int gm_base, int gn, int lane, EpiBuf<EPI>& b) {
    if (EPI != OCN_EPI_BIAS_RESID_F32 && EPI != OCN_EPI_BIAS_RESID_BF16 && EPI != OCN_EPI_DGELU) return;
    const int gnc = gn < a.N ? gn : a.N - 4;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        int gm = gm_base + it * 4 + (lane >> 4);
        gm = gm < a.M ? gm : a.M - 1;
        const size_t o = (size_t)gm * a.ldc + gnc;
        if (EPI == OCN_EPI_BIAS_RESID_F32) {
            b.r[it] = *(const f32x4*)((const float*)a.resid + o);
        } else if (EPI == OCN_EPI_BIAS_RESID_BF16) {
            const bf16x4 h = *(const bf16x4*)((const bf16*)a.resid + o);
            b.r[it] = (f32x4){bf2f(h[0]), bf2f(h[1]), bf2f(h[2]), bf2f(h[3])};
        } else {
            b.r[it] = dgelu_unpack4(*(const unsigned*)(a.aux + o));
        }
    }
}

template <int EPI>
OCN_DEV void epi_apply_store(const GemmNtArgs& a, int gm, int gn, f32x4 v, f32x4 bias, f32x4 extra) {
    const size_t o = (size_t)gm * a.ldc + gn;
    if (EPI == OCN_EPI_BF16 || EPI == OCN_EPI_F32) v = v * a.alpha + bias; else v = v + bias;
    if (EPI == OCN_EPI_BF16) {
        bf16x4 o4 = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *(bf16x4*)((bf16*)a.out + o) = o4;
    } else if (EPI == OCN_EPI_BIAS_GELU || EPI == OCN_EPI_BIAS_QUICKGELU) {
        f32x4 gv, dv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float g1, d1;
            act_both<EPI == OCN_EPI_BIAS_QUICKGELU>(v[e], g1, d1);
            gv[e] = g1;
            dv[e] = d1;
        }
        *(unsigned*)(a.aux + o) = dgelu_pack4(dv[0], dv[1], dv[2], dv[3]);  // aux = gelu'(pre-activation) in 8 bits
        bf16x4 o4 = {f2bf(gv[0]), f2bf(gv[1]), f2bf(gv[2]), f2bf(gv[3])};
        *(bf16x4*)((bf16*)a.out + o) = o4;
    } else if (EPI == OCN_EPI_BIAS_RESID_F32) {
        *(f32x4*)((float*)a.out + o) = v + extra;
    } else if (EPI == OCN_EPI_BIAS_RESID_BF16) {
        bf16x4 o4 = {f2bf(bf2f(f2bf(v[0])) + extra[0]), f2bf(bf2f(f2bf(v[1])) + extra[1]), f2bf(bf2f(f2bf(v[2])) + extra[2]), f2bf(bf2f(f2bf(v[3])) + extra[3])};
        *(bf16x4*)((bf16*)a.out + o) = o4;
    } else if (EPI == OCN_EPI_DGELU) {
        bf16x4 o4 = {f2bf(v[0] * extra[0]), f2bf(v[1] * extra[1]), f2bf(v[2] * extra[2]), f2bf(v[3] * extra[3])};
        *(bf16x4*)((bf16*)a.out + o) = o4;
    } else {
        *(f32x4*)((float*)a.out + o) = v;
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_ring_kernel(GemmNtArgs a) {
    constexpr int RBM = 256, RBN = 256, RBK = 32, NSTAGE = 4;
    constexpr int A_BYTES = RBM * 64, STAGE = (RBM + RBN) * 64;  // 32 KiB per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];  // NSTAGE * STAGE = 128 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = xcd_remap(blockIdx.x, a.ntiles);
    const int m0 = (tile / a.tiles_n) * RBM, n0 = (tile % a.tiles_n) * RBN;
    const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves, wave tile 128 x 64
    const int lr = lane & 31, lh = lane >> 5;
    const int nk = a.K / RBK;

    // per-lane DMA source pointers (advance by one K-step = 64 bytes); wave w moves rows [16w,16w+16) and
    // [128+16w, ...) of the A tile and the same of the B tile: 4 LDS-DMA instructions per stage per wave
    const bf16* pa[2];
    const bf16* pb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave + j * 8) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ swz64(r);
        int ga = m0 + r, gb = n0 + r;
        ga = ga < a.M ? ga : a.M - 1;
        gb = gb < a.N ? gb : a.N - 1;
        pa[j] = a.A + (size_t)ga * a.lda + c * 8;
        pb[j] = a.B + (size_t)gb * a.ldb + c * 8;
    }
#define OCN_DMA_A(J, SLOT) glds16(pa[J], (OCN_LDS void*)(smem + (SLOT) * STAGE + (wave + (J) * 8) * 1024)); pa[J] += RBK;
#define OCN_DMA_B(J, SLOT) glds16(pb[J], (OCN_LDS void*)(smem + (SLOT) * STAGE + A_BYTES + (wave + (J) * 8) * 1024)); pb[J] += RBK;
    // issue the first three stages, then the first epilogue slab's operands (older loads retire first)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s < nk) {
            OCN_DMA_A(0, s) OCN_DMA_A(1, s) OCN_DMA_B(0, s) OCN_DMA_B(1, s)
        }
    }
    const bool vec_ok = ((a.N & 3) == 0) && ((a.ldc & 3) == 0);
    const int col = (lane & 15) * 4;
    const int gn = n0 + wn * 64 + col;
    const int gm_wave = m0 + wm * 128;
    EpiBuf<EPI> ebuf;
    if (vec_ok) epi_prefetch<EPI>(a, gm_wave, gn, lane, ebuf);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_row = (wm * 128 + lr) * 64, b_row = (wn * 64 + lr) * 64;
    const unsigned lds_base = (unsigned)(size_t)(OCN_LDS char*)smem;
    const int sw = swz64(lr);
    const int cpos0 = ((0 + lh) ^ sw) << 4, cpos1 = ((2 + lh) ^ sw) << 4;  // k-substeps 0 / 1 of a stage
    bf16x8 a0[4], b0[2], a1[4], b1[2];
    // Fragment reads are inline asm so that hipcc does not count them: its own bookkeeping waits lgkmcnt(0) for
    // loop-carried reads, which exposes the LDS latency of the reads issued just before.  Waits are placed by hand
    // (LDS returns in order): every read group has 8 MFMAs of cover before the counted wait that retires it, and the
    // loop-carried fragments are complete at the back-edge (so compiler copies of them are safe).
#define OCN_LOAD_FRAGS(AF, BF, STG, CPOS)                                                        \
    {                                                                                            \
        const unsigned sA_ = lds_base + (unsigned)(((STG)&3) * STAGE + a_row + (CPOS));           \
        const unsigned sB_ = lds_base + (unsigned)(((STG)&3) * STAGE + A_BYTES + b_row + (CPOS)); \
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\t"              \
                     "ds_read_b128 %2, %4 offset:4096\n\tds_read_b128 %3, %4 offset:6144"       \
                     : "=&v"(AF[0]), "=&v"(AF[1]), "=&v"(AF[2]), "=&v"(AF[3]) : "v"(sA_));            \
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:2048"                   \
                     : "=&v"(BF[0]), "=&v"(BF[1]) : "v"(sB_));                                      \
    }
#define OCN_MFMA_BLOCK(AF, BF)                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(AF[i], BF[j], acc[i][j]);

    // stage 0 must be visible before the first fragment read
    if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    OCN_LOAD_FRAGS(a0, b0, 0, cpos0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // One K-step.  Fragments of (kt, k-substep 0) are already in registers.  Make stage kt+1 visible (its first
    // fragments are fetched at the end of the step); stage kt+2 may stay in flight (4 DMA ops per stage per wave).
    // The 4 DMA issues of stage kt+3 are spread between the first MFMAs so that they sit in the shadow of queued
    // matrix work instead of in front of it.  DMA / WAIT4 are compile-time so the body is branch-free.
#define OCN_RING_STEP(DMA, WAIT4)                                                                  \
    {                                                                                              \
        if (WAIT4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
        __builtin_amdgcn_s_barrier();                                                              \
        const int slot = (kt + 3) & 3;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        OCN_LOAD_FRAGS(a1, b1, kt, cpos1);                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        acc[0][0] = mfma32(a0[0], b0[0], acc[0][0]);                                               \
        acc[0][1] = mfma32(a0[0], b0[1], acc[0][1]);                                               \
        if (DMA) { OCN_DMA_A(0, slot) }                                                            \
        acc[1][0] = mfma32(a0[1], b0[0], acc[1][0]);                                               \
        acc[1][1] = mfma32(a0[1], b0[1], acc[1][1]);                                               \
        if (DMA) { OCN_DMA_A(1, slot) }                                                            \
        acc[2][0] = mfma32(a0[2], b0[0], acc[2][0]);                                               \
        acc[2][1] = mfma32(a0[2], b0[1], acc[2][1]);                                               \
        if (DMA) { OCN_DMA_B(0, slot) }                                                            \
        acc[3][0] = mfma32(a0[3], b0[0], acc[3][0]);                                               \
        acc[3][1] = mfma32(a0[3], b0[1], acc[3][1]);                                               \
        if (DMA) { OCN_DMA_B(1, slot) }                                                            \
        const int nstg = kt + 1 < nk ? kt + 1 : kt;                                                \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        OCN_LOAD_FRAGS(a0, b0, nstg, cpos0);                                                       \
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");  /* a1, b1 landed */                    \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        OCN_MFMA_BLOCK(a1, b1);                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  /* a0, b0 of the next step landed */   \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
    int kt = 0;
    for (; kt + 3 < nk; ++kt) OCN_RING_STEP(true, true)
    for (; kt + 2 < nk; ++kt) OCN_RING_STEP(false, true)
    for (; kt < nk; ++kt) OCN_RING_STEP(false, false)
#undef OCN_RING_STEP
#undef OCN_LOAD_FRAGS
#undef OCN_MFMA_BLOCK
#undef OCN_DMA_A
#undef OCN_DMA_B
    __builtin_amdgcn_s_barrier();  // all waves finished reading the ring; reuse it as per-wave C staging

    float* sC = (float*)(smem + wave * 8192);  // 32 rows x 64 cols fp32 per wave
    if (vec_ok) {
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (a.bias && gn < a.N) bias4 = *(const f32x4*)(a.bias + gn);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sC[mfma32_row(r, lane) * 64 + j * 32 + lr] = acc[i][j][r];
            EpiBuf<EPI> cur = ebuf;
            if (i + 1 < 4) epi_prefetch<EPI>(a, gm_wave + (i + 1) * 32, gn, lane, ebuf);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + (lane >> 4);
                const int gm = gm_wave + i * 32 + row;
                const f32x4 v = *(const f32x4*)(sC + row * 64 + col);
                if (gm < a.M && gn < a.N) epi_apply_store<EPI>(a, gm, gn, v, bias4, cur.r[it]);
            }
        }
    } else {  // ragged N / unaligned ldc: scalar stores (tests and tiny heads only)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sC[mfma32_row(r, lane) * 64 + j * 32 + lr] = acc[i][j][r];
#pragma unroll 1
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + (lane >> 4);
                const int gm = gm_wave + i * 32 + row;
                const f32x4 v = *(const f32x4*)(sC + row * 64 + col);
                if (gm < a.M) {
#pragma unroll 1
                    for (int e = 0; e < 4; ++e)
                        if (gn + e < a.N) epilogue_store1<EPI>(a, gm, gn + e, v[e]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// TN: dW[n,k] += alpha * sum_m A[m,n] * B[m,k]
// ------------------------------------------------------------------------------------------------

// rows [m_base, m_base+64) (zero beyond m_end) x cols [col0, col0+128) -> LDS tile of 64 rows x 256 B
OCN_DEV void stage_tn(const bf16* __restrict__ G, int ld, int m_base, int m_end, int col0, int ncols, char* sT, int wave,
                      int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int seg = wave * 4 + j;  // 4 rows per wave-instruction
        const int r = seg * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((r & 3) << 2);
        const int gm = m_base + r, col = col0 + c * 8;
        const void* src = (gm < m_end && col < ncols) ? (const void*)(G + (size_t)gm * ld + col) : (const void*)g_zero16;
        glds16(src, (OCN_LDS void*)(sT + seg * 1024));
    }
}

// MFMA operand (8 k-slots of this lane) for the 32-wide block starting at column cb of a TN tile, k-step s
OCN_DEV bf16x8 frag_tn(const char* sT, int cb, int s, int lane) {
    const int i = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
    const int chunk = ((cb + g * 16) >> 3) + ((i & 3) >> 1);
    const int p = chunk ^ ((i >> 2) << 2);
    const int row = s * 16 + h * 8 + (i >> 2);
    const char* base = sT + row * 256 + p * 16 + (i & 1) * 8;
    const s16x4 lo = lds_read_tr16((const OCN_LDS void*)base);
    const s16x4 hi = lds_read_tr16((const OCN_LDS void*)(base + 4 * 256));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTnArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wid = xcd_remap(blockIdx.x, a.nwg);
    const int ntile = a.tiles_n * a.tiles_k;
    const int split = wid / ntile, tile = wid % ntile;
    const int n0 = (tile / a.tiles_k) * 128, k0 = (tile % a.tiles_k) * 128;
    const int m_begin = split * a.chunk;
    const int m_end = min(a.M, m_begin + a.chunk);
    const int wn = wave >> 1, wk = wave & 1;
    const bool do_bias = (a.dbias != nullptr) && (k0 == 0) && (wk == 0);

    f32x16 acc[2][2], accb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    const int nsteps = (m_end - m_begin + 63) / 64;
    if (nsteps > 0) {
        stage_tn(a.A, a.lda, m_begin, m_end, n0, a.N, smem, wave, lane);
        stage_tn(a.B, a.ldb, m_begin, m_end, k0, a.K, smem + TILE_BYTES, wave, lane);
    }
    for (int st = 0; st < nsteps; ++st) {
        char* cur = smem + (st & 1) * STAGE_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 1 < nsteps) {
            char* nxt = smem + ((st + 1) & 1) * STAGE_BYTES;
            stage_tn(a.A, a.lda, m_begin + (st + 1) * 64, m_end, n0, a.N, nxt, wave, lane);
            stage_tn(a.B, a.ldb, m_begin + (st + 1) * 64, m_end, k0, a.K, nxt + TILE_BYTES, wave, lane);
        }
        const char* sA = cur;
        const char* sB = cur + TILE_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 af[2], bq[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = frag_tn(sA, wn * 64 + i * 32, s, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) bq[j] = frag_tn(sB, wk * 64 + j * 32, s, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(af[i], bq[j], acc[i][j]);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 2; ++i) accb[i] = mfma32(af[i], ones, accb[i]);
            }
        }
    }
    const int lr = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gk = k0 + wk * 64 + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gn = n0 + wn * 64 + i * 32 + mfma32_row(r, lane);
                if (gn < a.N && gk < a.K) unsafeAtomicAdd(a.dW + (size_t)gn * a.ldw + gk, a.alpha * acc[i][j][r]);
            }
        }
    if (do_bias && lr == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gn = n0 + wn * 64 + i * 32 + mfma32_row(r, lane);
                if (gn < a.N) unsafeAtomicAdd(a.dbias + gn, a.alpha * accb[i][r]);
            }
    }
}

int g_tn_variant = 0;  // 0 = auto, 1 = general 128x128 kernel, 3 = 256x256 hand-scheduled (gemm_tn5.hip)
int g_nt_ablate = 0;
int g_nt_variant = 0;  // 0 = auto, 4 = general 256x256 ring kernel (any M, N; K % 32), 5 = persistent 256x256 (gemm_nt5.hip; K % 128)

template <int EPI>
int launch_nt(const GemmNtArgs& a, hipStream_t st) {
    int v = g_nt_variant;
    if (v == 0) v = (a.M >= 1024 && a.N >= 192) ? 5 : 4;
    if (v == 5) {  // persistent 256x256 kernel (gemm_nt5.hip); falls through to the general kernel when the shape does not fit it
        GemmNtArgs a5 = a;
        a5.ablate = g_nt_ablate;
        const int rc = ocn_launch_nt5(EPI, a5, st);
        if (rc <= 0) return rc;
    }
    GemmNtArgs b = a;  // the one general kernel: ragged M / N (scalar epilogue when N % 4), K in steps of 32
    b.ablate = g_nt_ablate;
    b.tiles_n = ocn_cdiv(b.N, 256);
    b.ntiles = ocn_cdiv(b.M, 256) * b.tiles_n;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_nt_ring_kernel<EPI>, dim3(b.ntiles), dim3(512), 131072, st, b);
    OCN_CHECK_LAUNCH("ocn_gemm_nt");
    return OCN_OK;
}

}  // namespace

extern "C" int ocn_gemm_nt(int epilogue, const void* A, int lda, const void* B, int ldb, void* out, int ldc, int M, int N,
                           int K, const float* bias, const void* resid, void* aux, float alpha, ocn_stream_t stream) {
    OCN_CHECK_ARG(A && B && out, "ocn_gemm_nt: null operand");
    OCN_CHECK_ARG(M > 0 && N > 0 && K > 0, "ocn_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
    OCN_CHECK_ARG(K % 32 == 0, "ocn_gemm_nt: K=%d must be a multiple of 32", K);
    OCN_CHECK_ARG(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0, "ocn_gemm_nt: bad leading dims");
    OCN_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)out & 15) == 0,
                  "ocn_gemm_nt: operands must be 16-byte aligned");
    OCN_CHECK_ARG((epilogue != OCN_EPI_BIAS_RESID_F32 && epilogue != OCN_EPI_BIAS_RESID_BF16) || resid, "ocn_gemm_nt: residual epilogue needs resid");
    OCN_CHECK_ARG((epilogue != OCN_EPI_BIAS_GELU && epilogue != OCN_EPI_BIAS_QUICKGELU && epilogue != OCN_EPI_DGELU) || aux, "ocn_gemm_nt: gelu epilogues need aux");
    GemmNtArgs a;
    a.A = (const bf16*)A; a.B = (const bf16*)B; a.out = out; a.bias = bias; a.resid = resid; a.aux = (unsigned char*)aux;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.alpha = alpha;
    a.tiles_n = 0; a.ntiles = 0; a.band = 0; a.stagger = 0; a.first_wave = 0; a.ablate = 0; a.ksplit = 1; a.ws_stride = 0; a.rescue = nullptr;
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case OCN_EPI_BF16: return launch_nt<OCN_EPI_BF16>(a, st);
        case OCN_EPI_BIAS_GELU: return launch_nt<OCN_EPI_BIAS_GELU>(a, st);
        case OCN_EPI_BIAS_QUICKGELU: return launch_nt<OCN_EPI_BIAS_QUICKGELU>(a, st);
        case OCN_EPI_BIAS_RESID_F32: return launch_nt<OCN_EPI_BIAS_RESID_F32>(a, st);
        case OCN_EPI_BIAS_RESID_BF16: return launch_nt<OCN_EPI_BIAS_RESID_BF16>(a, st);
        case OCN_EPI_DGELU: return launch_nt<OCN_EPI_DGELU>(a, st);
        case OCN_EPI_F32: return launch_nt<OCN_EPI_F32>(a, st);
    }
    ocn_set_error("ocn_gemm_nt: unknown epilogue %d", epilogue);
    return OCN_ERR_INVALID;
}

// ---- split-K NT GEMM: few output tiles, long K (the loss's G @ Y: [4096 x 512] from K = 32768 has 32 tiles for 256 CUs) -------------------------
namespace {
// out[m, n] = scale * (rowscale[m] * sum_s ws[s][m][n] - sub_alpha * sub[m][n]);  rowscale / sub / scale optional
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, long slab, int ksplit, float* __restrict__ out, int ldc, int M, int N,
                                                             const float* __restrict__ rowscale, const bf16* __restrict__ sub, int ld_sub, float sub_alpha,
                                                             const float* __restrict__ scale_dev) {
    const int n4 = N / 4;
    const long total = (long)M * n4;
    const float sc = scale_dev ? *scale_dev : 1.0f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n = (int)(i % n4) * 4;
        const size_t o = (size_t)m * ldc + n;
        f32x4 v = *(const f32x4*)(ws + o);
        for (int s = 1; s < ksplit; ++s) v = v + *(const f32x4*)(ws + (size_t)s * slab + o);
        if (rowscale) v = v * rowscale[m];
        if (sub) {
            const bf16x4 h = *(const bf16x4*)(sub + (size_t)m * ld_sub + n);
            v = v - (f32x4){bf2f(h[0]), bf2f(h[1]), bf2f(h[2]), bf2f(h[3])} * sub_alpha;
        }
        *(f32x4*)(out + o) = v * sc;
    }
}
int g_splitk_cus = 0;
}  // namespace

// K-slices a [M, N] = A[M, K] . B[N, K]^T product should be cut into so that its 256 x 256 tiles fill the chip: 1 (= use ocn_gemm_nt) unless the
// tiles cover at most half of the CUs and every slice keeps K >= 1024 (a multiple of 128)
extern "C" int ocn_gemm_nt_splitk_plan(int M, int N, int K) {
    if (g_splitk_cus == 0) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_splitk_cus = n;
    }
    const int tiles = ocn_cdiv(M, 256) * ocn_cdiv(N, 256);
    int ks = 1;
    while (tiles * ks * 2 <= g_splitk_cus && K % (128 * ks * 2) == 0 && K / (ks * 2) >= 1024) ks *= 2;
    return (M >= 1024 && N >= 192 && N % 8 == 0) ? ks : 1;
}

extern "C" int ocn_gemm_nt_splitk(const void* A, int lda, const void* B, int ldb, float* out, int ldc, int M, int N, int K, int ksplit, float* workspace,
                                  const float* rowscale, const void* sub_rows_bf16, int ld_sub, float sub_alpha, const float* scale_dev, ocn_stream_t stream) {
    OCN_CHECK_ARG(A && B && out && workspace, "ocn_gemm_nt_splitk: null operand");
    OCN_CHECK_ARG(M > 0 && N > 0 && K > 0 && ksplit >= 2 && K % (128 * ksplit) == 0 && N % 8 == 0 && ldc % 8 == 0 && ldc >= N && lda >= K && ldb >= K && lda % 8 == 0 && ldb % 8 == 0,
                  "ocn_gemm_nt_splitk: unsupported shape M=%d N=%d K=%d ksplit=%d (K must be a multiple of 128 * ksplit; N, ldc, lda, ldb of 8)", M, N, K, ksplit);
    OCN_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)workspace & 15) == 0, "ocn_gemm_nt_splitk: operands must be 16-byte aligned");
    OCN_CHECK_ARG(!sub_rows_bf16 || (ld_sub >= N && ld_sub % 4 == 0), "ocn_gemm_nt_splitk: bad ld_sub");
    GemmNtArgs a;
    a.A = (const bf16*)A; a.B = (const bf16*)B; a.out = workspace; a.bias = nullptr; a.resid = nullptr; a.aux = nullptr;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.alpha = 1.0f;
    a.tiles_n = 0; a.ntiles = 0; a.band = 0; a.stagger = 0; a.first_wave = 0; a.ablate = 0;
    a.ksplit = ksplit; a.ws_stride = (long)M * ldc;
    hipStream_t st = (hipStream_t)stream;
    const int rc = ocn_launch_nt5(OCN_EPI_F32, a, st);
    if (rc != 0) { if (rc > 0) ocn_set_error("ocn_gemm_nt_splitk: shape not supported by the persistent GEMM"); return rc > 0 ? OCN_ERR_UNSUPPORTED : rc; }
    const long total = (long)M * (N / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, workspace, (long)M * ldc, ksplit, out, ldc, M, N, rowscale, (const bf16*)sub_rows_bf16, ld_sub,
                       sub_alpha, scale_dev);
    OCN_CHECK_LAUNCH("ocn_gemm_nt_splitk");
    return OCN_OK;
}

extern "C" int ocn_set_gemm_variant(int nt_variant) {
    g_nt_ablate = nt_variant >> 8;   // developer ablation mask in the high bits
    g_tn_variant = (nt_variant >> 4) & 15;  // bits 4..7: TN kernel choice
    nt_variant &= 15;
    g_nt_variant = nt_variant;
    return OCN_OK;
}

namespace {
// shared body of ocn_gemm_tn_accum / ocn_gemm_tn_accum_det.  `det`: no two workgroups ever add into the same address -- the hand-scheduled
// kernel stores per-split slabs into `workspace` and sums them in split order, the general kernel runs with ONE M-split.
int tn_accum(const void* A, int lda, const void* B, int ldb, float* dW, int ldw, int M, int N, int K, float* dbias, float alpha, bool det,
             void* workspace, int64_t workspace_bytes, ocn_stream_t stream) {
    OCN_CHECK_ARG(A && B && dW, "ocn_gemm_tn_accum: null operand");
    OCN_CHECK_ARG(M > 0 && N > 0 && K > 0, "ocn_gemm_tn_accum: bad shape M=%d N=%d K=%d", M, N, K);
    OCN_CHECK_ARG(N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "ocn_gemm_tn_accum: N, K, lda, ldb must be multiples of 8");
    OCN_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "ocn_gemm_tn_accum: operands must be 16-byte aligned");
    GemmTnArgs a;
    a.A = (const bf16*)A; a.B = (const bf16*)B; a.dW = dW; a.dbias = dbias;
    a.lda = lda; a.ldb = ldb; a.ldw = ldw; a.M = M; a.N = N; a.K = K; a.alpha = alpha; a.ablate = 0; a.nsplit = 0; a.ws = nullptr;
    const long need = det ? ocn_tn5_workspace_bytes(M, N, K) : 0;
    if (det && need > 0) {
        OCN_CHECK_ARG(workspace && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0 && ((uintptr_t)dW & 15) == 0 && ldw % 4 == 0,
                      "ocn_gemm_tn_accum_det: needs %ld bytes of 16-byte aligned workspace (ocn_gemm_tn_det_workspace_bytes) and an aligned dW", need);
        a.ws = (float*)workspace;
    }
    const bool big = (long)M * N * K >= (1L << 31) && N >= 256 && K >= 256;
    if ((!det && (g_tn_variant == 3 || (g_tn_variant == 0 && big))) || (det && need > 0)) {  // hand-scheduled 256x256 kernel (gemm_tn5.hip); falls through if the shape does not fit it
        const int rc = ocn_launch_tn5(a, (hipStream_t)stream);
        if (rc < 0) ocn_set_error("ocn_gemm_tn_accum: launch failed");
        if (rc <= 0) return rc;
        a.ws = nullptr;
    }
    a.A2 = a.B2 = nullptr; a.dW2 = a.dbias2 = nullptr; a.lda2 = a.ldb2 = a.ldw2 = a.N2 = a.ntile1 = a.ntile_all = 0;
    const int T = 128, RS = 64;  // the general kernel: 128x128 tile of dW, 64 reduction rows per step, M split to ~6 workgroups per CU
    a.tiles_n = ocn_cdiv(N, T);
    a.tiles_k = ocn_cdiv(K, T);
    const int ntile = a.tiles_n * a.tiles_k;
    const int msteps = ocn_cdiv(M, RS);
    int splits = det ? 1 : ocn_cdiv(1536, ntile);
    if (splits > msteps) splits = msteps;
    if (splits < 1) splits = 1;
    a.chunk = ocn_cdiv(msteps, splits) * RS;
    splits = ocn_cdiv(M, a.chunk);
    a.nwg = splits * ntile;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(a.nwg), dim3(256), 0, (hipStream_t)stream, a);
    OCN_CHECK_LAUNCH("ocn_gemm_tn_accum");
    return OCN_OK;
}
}  // namespace

extern "C" int ocn_gemm_tn_accum(const void* A, int lda, const void* B, int ldb, float* dW, int ldw, int M, int N, int K,
                                 float* dbias, float alpha, ocn_stream_t stream) {
    return tn_accum(A, lda, B, ldb, dW, ldw, M, N, K, dbias, alpha, false, nullptr, 0, stream);
}

extern "C" int64_t ocn_gemm_tn_det_workspace_bytes(int M, int N, int K) { return ocn_tn5_workspace_bytes(M, N, K); }

extern "C" int ocn_gemm_tn_accum_det(const void* A, int lda, const void* B, int ldb, float* dW, int ldw, int M, int N, int K, float* dbias,
                                     float alpha, void* workspace, int64_t workspace_bytes, ocn_stream_t stream) {
    return tn_accum(A, lda, B, ldb, dW, ldw, M, N, K, dbias, alpha, true, workspace, workspace_bytes, stream);
}

// Two weight gradients over the SAME rows and the same K in one launch:  dW1[N1,K] += alpha * A1[M,N1]^T . B1[M,K]  and
// dW2[N2,K] += alpha * A2[M,N2]^T . B2[M,K]  (+ their bias gradients, both or neither).  A block's out-proj and QKV wgrads are such a
// pair (model.py::_BlockFn.backward): alone the 768 x 768 one needs 28 M-splits to fill the chip and spends a quarter of its time
// in contended atomics.  Shapes the paired kernel does not take run as two ocn_gemm_tn_accum calls.
extern "C" int ocn_gemm_tn_accum2(const void* A1, int lda1, const void* B1, int ldb1, float* dW1, int ldw1, float* dbias1, int N1,
                                  const void* A2, int lda2, const void* B2, int ldb2, float* dW2, int ldw2, float* dbias2, int N2,
                                  int M, int K, float alpha, ocn_stream_t stream) {
    OCN_CHECK_ARG(A1 && B1 && dW1 && A2 && B2 && dW2, "ocn_gemm_tn_accum2: null operand");
    OCN_CHECK_ARG(M > 0 && N1 > 0 && N2 > 0 && K > 0, "ocn_gemm_tn_accum2: bad shape M=%d N1=%d N2=%d K=%d", M, N1, N2, K);
    const bool big = (long)M * (N1 + N2) * K >= (1L << 31) && N1 >= 256 && N2 >= 256 && K >= 256;
    const bool aligned = (((uintptr_t)A1 | (uintptr_t)B1 | (uintptr_t)A2 | (uintptr_t)B2) & 15) == 0;
    if (big && aligned && (g_tn_variant == 0 || g_tn_variant == 3) && g_ocn_tuning[13] != 1) {  // developer knob 13 = 1: never pair
        GemmTnArgs a;
        a.A = (const bf16*)A1; a.B = (const bf16*)B1; a.dW = dW1; a.dbias = dbias1; a.lda = lda1; a.ldb = ldb1; a.ldw = ldw1; a.N = N1;
        a.A2 = (const bf16*)A2; a.B2 = (const bf16*)B2; a.dW2 = dW2; a.dbias2 = dbias2; a.lda2 = lda2; a.ldb2 = ldb2; a.ldw2 = ldw2; a.N2 = N2;
        a.M = M; a.K = K; a.alpha = alpha; a.ablate = 0; a.nsplit = 0; a.ws = nullptr;
        const int rc = ocn_launch_tn5_pair(a, (hipStream_t)stream);
        if (rc < 0) ocn_set_error("ocn_gemm_tn_accum2: launch failed");
        if (rc <= 0) return rc;
    }
    if (int e = ocn_gemm_tn_accum(A1, lda1, B1, ldb1, dW1, ldw1, M, N1, K, dbias1, alpha, stream)) return e;
    return ocn_gemm_tn_accum(A2, lda2, B2, ldb2, dW2, ldw2, M, N2, K, dbias2, alpha, stream);
}
