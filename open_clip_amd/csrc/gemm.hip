// MFMA GEMMs for the CLIP towers on gfx950.
//
//   ocn_gemm_nt        C[M,N]  = A[M,K] . B[N,K]^T  (+ fused epilogue)      forward linears, dgrads, logits
//   ocn_gemm_tn_accum  dW[N,K] += A[M,N]^T . B[M,K] (+ dbias = colsum(A))   wgrads, loss G^T products
//
// Both: 128x128 workgroup tile, 4 waves (2x2), each wave a 64x64 sub-tile as 2x2 v_mfma_f32_32x32x16_bf16
// accumulators; operands reach LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) into a
// double-buffered 2 x 32 KiB ring, one barrier per K-step; two workgroups per CU so one workgroup's
// epilogue overlaps the other's main loop.  LDS images are XOR-swizzled on the *source* address (the DMA
// destination is lane-linear) so that fragment reads are bank-conflict free:
//   NT  : rows of 64 bf16 (128 B), ds_read_b128 fragments, chunk ^= f(row)          (f: see swz_nt)
//   TN  : rows of 128 bf16 (256 B), ds_read_b64_tr_b16 transposing reads, chunk ^= (row&3)<<2
// Tiles are walked in an XCD-aware order (ocn_common.h xcd_remap).
#include "ocn_common.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 16384;            // one operand tile (128x64 or 64x128 bf16)
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // A + B
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;  // double buffered: 64 KiB -> 2 workgroups / CU

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4];  // zero source for out-of-range DMA lanes

// chunk swizzle for 128-byte rows: bijection on 3 bits built from row bits 1..3, chosen so that
// (a) the four 16-lane groups of a ds_read_b128 fragment read hit 16 distinct 16-byte slots and
// (b) 4 consecutive rows land in different 64-byte quarters (needed by tr16 reads of the same image).
OCN_DEV int swz_nt(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

struct GemmNtArgs {
    const bf16* A;
    const bf16* B;
    void* out;
    const float* bias;
    const float* resid;
    bf16* aux;
    int lda, ldb, ldc, M, N, K;
    float alpha;
    int tiles_n, ntiles;
};

// rows [row0, row0+ROWS) x k [k0, k0+64) of a row-major bf16 matrix -> LDS tile (rows clamped), NW waves
template <int ROWS, int NW>
OCN_DEV void stage_nt(const bf16* __restrict__ G, int ld, int row0, int nrows, int k0, char* sT, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < ROWS / 8 / NW; ++j) {
        const int seg = wave + j * NW;  // 8 rows per wave-instruction
        const int r = seg * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz_nt(r);
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1;
        glds16(G + (size_t)gr * ld + k0 + c * 8, (OCN_LDS void*)(sT + seg * 1024));
    }
}

template <int EPI>
OCN_DEV void epilogue_store4(const GemmNtArgs& a, int gm, int gn, f32x4 v) {
    const size_t o = (size_t)gm * a.ldc + gn;
    if (a.bias) {
        const f32x4 b = *(const f32x4*)(a.bias + gn);
        if (EPI == OCN_EPI_BF16 || EPI == OCN_EPI_F32) v = v * a.alpha + b; else v = v + b;
    } else if (EPI == OCN_EPI_BF16 || EPI == OCN_EPI_F32) {
        v = v * a.alpha;
    }
    if (EPI == OCN_EPI_BF16) {
        bf16x4 o4 = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *(bf16x4*)((bf16*)a.out + o) = o4;
    } else if (EPI == OCN_EPI_BIAS_GELU) {
        bf16x4 p4 = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *(bf16x4*)(a.aux + o) = p4;
        bf16x4 o4 = {f2bf(gelu_f(v[0])), f2bf(gelu_f(v[1])), f2bf(gelu_f(v[2])), f2bf(gelu_f(v[3]))};
        *(bf16x4*)((bf16*)a.out + o) = o4;
    } else if (EPI == OCN_EPI_BIAS_RESID_F32) {
        const f32x4 r = *(const f32x4*)(a.resid + o);
        *(f32x4*)((float*)a.out + o) = v + r;
    } else if (EPI == OCN_EPI_DGELU) {
        const bf16x4 p4 = *(const bf16x4*)(a.aux + o);
        bf16x4 o4 = {f2bf(v[0] * dgelu_f(bf2f(p4[0]))), f2bf(v[1] * dgelu_f(bf2f(p4[1]))),
                     f2bf(v[2] * dgelu_f(bf2f(p4[2]))), f2bf(v[3] * dgelu_f(bf2f(p4[3])))};
        *(bf16x4*)((bf16*)a.out + o) = o4;
    } else {  // OCN_EPI_F32
        *(f32x4*)((float*)a.out + o) = v;
    }
}

template <int EPI>
OCN_DEV void epilogue_store1(const GemmNtArgs& a, int gm, int gn, float v) {
    const size_t o = (size_t)gm * a.ldc + gn;
    const float b = a.bias ? a.bias[gn] : 0.f;
    if (EPI == OCN_EPI_BF16) {
        ((bf16*)a.out)[o] = f2bf(v * a.alpha + b);
    } else if (EPI == OCN_EPI_BIAS_GELU) {
        a.aux[o] = f2bf(v + b);
        ((bf16*)a.out)[o] = f2bf(gelu_f(v + b));
    } else if (EPI == OCN_EPI_BIAS_RESID_F32) {
        ((float*)a.out)[o] = v + b + a.resid[o];
    } else if (EPI == OCN_EPI_DGELU) {
        ((bf16*)a.out)[o] = f2bf((v + b) * dgelu_f(bf2f(a.aux[o])));
    } else {
        ((float*)a.out)[o] = v * a.alpha + b;
    }
}

// Geometry: workgroup tile BM x BN, WR x WC waves, each wave (BM/WR) x (BN/WC) = TI x TJ blocks of 32x32.
//   <128,128,2,2>: 4 waves, 64 KiB ring, 2 workgroups / CU (small problems, ragged edges)
//   <256,256,2,4>: 8 waves, 128 KiB ring, 1 workgroup / CU, wave tile 128x64 (less LDS traffic per MFMA)
template <int EPI, int BM_, int BN_, int WR, int WC>
__global__ __launch_bounds__(WR* WC * 64, (WR * WC * 64 * ((BM_ + BN_) * 256 <= 65536 ? 2 : 1)) / 256)
void gemm_nt_kernel(GemmNtArgs a) {
    constexpr int NW = WR * WC;
    constexpr int TI = BM_ / WR / 32, TJ = BN_ / WC / 32;
    constexpr int A_BYTES = BM_ * 128, B_BYTES = BN_ * 128, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 * STAGE
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = xcd_remap(blockIdx.x, a.ntiles);
    const int m0 = (tile / a.tiles_n) * BM_, n0 = (tile % a.tiles_n) * BN_;
    const int wm = wave / WC, wn = wave % WC;
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = swz_nt(lr);

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_row = (wm * TI * 32 + lr) * 128, b_row = (wn * TJ * 32 + lr) * 128;
    const int nk = a.K / BK;

    stage_nt<BM_, NW>(a.A, a.lda, m0, a.M, 0, smem, wave, lane);
    stage_nt<BN_, NW>(a.B, a.ldb, n0, a.N, 0, smem + A_BYTES, wave, lane);
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * STAGE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my DMA pieces of tile kt have landed
        __syncthreads();                                  // everyone's have; everyone is done with tile kt-1
        if (kt + 1 < nk) {
            char* nxt = smem + ((kt + 1) & 1) * STAGE;
            stage_nt<BM_, NW>(a.A, a.lda, m0, a.M, (kt + 1) * BK, nxt, wave, lane);
            stage_nt<BN_, NW>(a.B, a.ldb, n0, a.N, (kt + 1) * BK, nxt + A_BYTES, wave, lane);
        }
        const char* sA = cur;
        const char* sB = cur + A_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int cpos = ((s * 2 + lh) ^ sw) << 4;
            bf16x8 af[TI], bq[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *(const bf16x8*)(sA + a_row + i * 32 * 128 + cpos);
#pragma unroll
            for (int j = 0; j < TJ; ++j) bq[j] = *(const bf16x8*)(sB + b_row + j * 32 * 128 + cpos);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = mfma32(af[i], bq[j], acc[i][j]);
        }
    }
    __syncthreads();  // all waves finished reading the ring; reuse it as per-wave C staging

    // stage 32 x (TJ*32) fp32 slabs of the wave's sub-tile through its private LDS region so that global
    // traffic is whole rows (full 128-/256-byte lines) for every epilogue operand
    constexpr int TW = TJ * 32;                      // columns of the wave tile
    float* sC = (float*)(smem + wave * (32 * TW * 4));  // <= 8 KiB per wave
    const bool vec_ok = ((a.N & 3) == 0) && ((a.ldc & 3) == 0);
    constexpr int LPR = TW / 4;                      // lanes per row (16 for TW=64)
    constexpr int RPI = 64 / LPR;                    // rows per iteration
    const int col = (lane % LPR) * 4;
    const int gn = n0 + wn * TW + col;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sC[mfma32_row(r, lane) * TW + j * 32 + lr] = acc[i][j][r];
#pragma unroll 4
        for (int it = 0; it < 32 / RPI; ++it) {
            const int row = it * RPI + lane / LPR;
            const int gm = m0 + (wm * TI + i) * 32 + row;
            const f32x4 v = *(const f32x4*)(sC + row * TW + col);
            if (gm < a.M) {
                if (vec_ok) {
                    if (gn < a.N) epilogue_store4<EPI>(a, gm, gn, v);
                } else {
#pragma unroll
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
struct GemmTnArgs {
    const bf16* A;
    const bf16* B;
    float* dW;
    float* dbias;
    int lda, ldb, ldw, M, N, K;
    float alpha;
    int tiles_n, tiles_k, chunk, nwg;
};

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

template <int EPI, int BM_, int BN_, int WR, int WC>
int launch_nt_geo(GemmNtArgs a, hipStream_t st) {
    constexpr int LDS = 2 * (BM_ + BN_) * 128;
    a.tiles_n = ocn_cdiv(a.N, BN_);
    a.ntiles = ocn_cdiv(a.M, BM_) * a.tiles_n;
    auto kern = gemm_nt_kernel<EPI, BM_, BN_, WR, WC>;
    if (LDS > 65536) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            attr_set = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(a.ntiles), dim3(WR * WC * 64), LDS, st, a);
    OCN_CHECK_LAUNCH("ocn_gemm_nt");
    return OCN_OK;
}

int g_nt_variant = 0;  // 0 = auto, 1 = 128x128, 2 = 256x256, 3 = 256x128

template <int EPI>
int launch_nt(const GemmNtArgs& a, hipStream_t st) {
    int v = g_nt_variant;
    if (v == 0) v = (a.M >= 2048 && a.N >= 256) ? 2 : 1;
    if (v == 2) return launch_nt_geo<EPI, 256, 256, 2, 4>(a, st);
    if (v == 3) return launch_nt_geo<EPI, 256, 128, 4, 2>(a, st);
    return launch_nt_geo<EPI, 128, 128, 2, 2>(a, st);
}

}  // namespace

extern "C" int ocn_gemm_nt(int epilogue, const void* A, int lda, const void* B, int ldb, void* out, int ldc, int M, int N,
                           int K, const float* bias, const float* resid, void* aux, float alpha, ocn_stream_t stream) {
    OCN_CHECK_ARG(A && B && out, "ocn_gemm_nt: null operand");
    OCN_CHECK_ARG(M > 0 && N > 0 && K > 0, "ocn_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
    OCN_CHECK_ARG(K % 64 == 0, "ocn_gemm_nt: K=%d must be a multiple of 64", K);
    OCN_CHECK_ARG(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0, "ocn_gemm_nt: bad leading dims");
    OCN_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)out & 15) == 0,
                  "ocn_gemm_nt: operands must be 16-byte aligned");
    OCN_CHECK_ARG(epilogue != OCN_EPI_BIAS_RESID_F32 || resid, "ocn_gemm_nt: residual epilogue needs resid");
    OCN_CHECK_ARG((epilogue != OCN_EPI_BIAS_GELU && epilogue != OCN_EPI_DGELU) || aux, "ocn_gemm_nt: gelu epilogues need aux");
    GemmNtArgs a;
    a.A = (const bf16*)A; a.B = (const bf16*)B; a.out = out; a.bias = bias; a.resid = resid; a.aux = (bf16*)aux;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.alpha = alpha;
    a.tiles_n = 0; a.ntiles = 0;
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case OCN_EPI_BF16: return launch_nt<OCN_EPI_BF16>(a, st);
        case OCN_EPI_BIAS_GELU: return launch_nt<OCN_EPI_BIAS_GELU>(a, st);
        case OCN_EPI_BIAS_RESID_F32: return launch_nt<OCN_EPI_BIAS_RESID_F32>(a, st);
        case OCN_EPI_DGELU: return launch_nt<OCN_EPI_DGELU>(a, st);
        case OCN_EPI_F32: return launch_nt<OCN_EPI_F32>(a, st);
    }
    ocn_set_error("ocn_gemm_nt: unknown epilogue %d", epilogue);
    return OCN_ERR_INVALID;
}

extern "C" int ocn_set_gemm_variant(int nt_variant) {
    g_nt_variant = nt_variant;
    return OCN_OK;
}

extern "C" int ocn_gemm_tn_accum(const void* A, int lda, const void* B, int ldb, float* dW, int ldw, int M, int N, int K,
                                 float* dbias, float alpha, ocn_stream_t stream) {
    OCN_CHECK_ARG(A && B && dW, "ocn_gemm_tn_accum: null operand");
    OCN_CHECK_ARG(M > 0 && N > 0 && K > 0, "ocn_gemm_tn_accum: bad shape M=%d N=%d K=%d", M, N, K);
    OCN_CHECK_ARG(N % 8 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "ocn_gemm_tn_accum: N, K, lda, ldb must be multiples of 8");
    OCN_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "ocn_gemm_tn_accum: operands must be 16-byte aligned");
    GemmTnArgs a;
    a.A = (const bf16*)A; a.B = (const bf16*)B; a.dW = dW; a.dbias = dbias;
    a.lda = lda; a.ldb = ldb; a.ldw = ldw; a.M = M; a.N = N; a.K = K; a.alpha = alpha;
    a.tiles_n = ocn_cdiv(N, 128);
    a.tiles_k = ocn_cdiv(K, 128);
    const int ntile = a.tiles_n * a.tiles_k;
    const int msteps = ocn_cdiv(M, 64);
    int splits = ocn_cdiv(1536, ntile);          // ~3 waves of workgroups over 256 CUs x 2
    if (splits > msteps) splits = msteps;
    if (splits < 1) splits = 1;
    a.chunk = ocn_cdiv(msteps, splits) * 64;
    splits = ocn_cdiv(M, a.chunk);
    a.nwg = splits * ntile;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(a.nwg), dim3(256), 0, (hipStream_t)stream, a);
    OCN_CHECK_LAUNCH("ocn_gemm_tn_accum");
    return OCN_OK;
}
