// Generic multi-head self-attention core: any head_dim D in {64, 80, 96, 128} (D % 16 == 0) and any L whose two resident
// operand images fit the LDS -- the path of the ViT-H-14 image tower (head_dim 80, 257 tokens; BASELINE config 5), which the
// specialised kernels of attention.hip (head_dim 64, everything of a head resident: <= 320 tokens) reject.
//
// Same arithmetic as attention.hip (swapped products S^T = K Q^T, O^T = V^T P^T so that softmax statistics are lane-local,
// fp32 softmax in the exp2 domain, causal mask as a predicate, backward recomputes P from the saved LSE), different residency:
//   * a workgroup = 4 waves = 4 blocks of 32 queries (forward, dQ) or of 32 keys (dK / dV) of ONE (batch, head); grid.y walks
//     the groups of 4 blocks;
//   * only the two operands that every wave needs in full are staged in LDS -- K, V (forward and dQ kernel) or Q, dO (dK / dV
//     kernel) -- as row-major [rows padded to 32][DP] bf16 images, DP = D rounded up to 32 (zero padded), no swizzle; the
//     wave's own rows (Q, dO, O resp. K, V) go straight from global memory into MFMA fragments;
//   * d-contiguous operands are read with ds_read_b128, transposed ones (V^T, K^T, Q^T, dO^T) with ds_read_b64_tr_b16, both
//     from the same image;
//   * the backward is two launches (dQ, then dK / dV) that exchange delta[q] = sum_d dO O through a caller-provided fp32
//     workspace [B, H, L].
// This is the parity path of the non-headline configurations: correct and MFMA-based, not tuned (plain staging loads, no
// LDS-DMA, bank conflicts accepted).
#include "ocn_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

OCN_DEV f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

OCN_DEV bf16x8 pack8(const f32x16& p, int t) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(p[8 * t + e]);
    return o;
}

// stage rows [0, LPAD) x [0, DP) of one head's column block (row stride `rs` elements, D valid columns, `rows` valid rows)
template <int D, int DP>
OCN_DEV void stage_image(const bf16* __restrict__ base, size_t rs, int rows, int lpad, char* sT) {
    constexpr int CH = DP / 8;  // 16-byte chunks per image row
    const bf16x8 zero = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    for (int idx = threadIdx.x; idx < lpad * CH; idx += blockDim.x) {
        const int r = idx / CH, c = idx % CH;
        bf16x8 v = zero;
        if (r < rows && c * 8 < D) v = *(const bf16x8*)(base + (size_t)r * rs + c * 8);
        *(bf16x8*)(sT + (size_t)r * (DP * 2) + c * 16) = v;
    }
}

// d-contiguous operand: row `row` of the image, k-step s (16 of the d's), this lane's 8 d's
template <int DP>
OCN_DEV bf16x8 frag_rows_p(const char* sT, int row, int s, int lane) {
    return *(const bf16x8*)(sT + (size_t)row * (DP * 2) + ((s * 2 + (lane >> 5)) << 4));
}

// the same fragment taken straight from global memory (row-major, `rs` elements per row)
OCN_DEV bf16x8 frag_rows_g(const bf16* __restrict__ base, size_t rs, int row, int s, int lane) {
    return *(const bf16x8*)(base + (size_t)row * rs + (s * 2 + (lane >> 5)) * 8);
}

// transposed operand: A[i = d (dblk*32 + lane&31)][k-slots <-> rows rbase + 16t + 8(e>>2) + 4h + (e&3)]  (cf. attention.hip)
template <int DP>
OCN_DEV bf16x8 frag_cols_p(const char* sT, int rbase, int t, int dblk, int lane) {
    const int i = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
    const int chunk = dblk * 4 + g * 2 + ((i & 3) >> 1);
    const int r0 = rbase + 16 * t + 4 * h + (i >> 2);
    const int r1 = r0 + 8;
    const s16x4 lo = lds_read_tr16((const OCN_LDS void*)(sT + (size_t)r0 * (DP * 2) + (chunk << 4) + (i & 1) * 8));
    const s16x4 hi = lds_read_tr16((const OCN_LDS void*)(sT + (size_t)r1 * (DP * 2) + (chunk << 4) + (i & 1) * 8));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// a wave's [32 rows] x [DP] result (acc[dblk]: lane <-> row lr, registers 4*q4.. <-> d = dblk*32 + 8*q4 + 4*lh ..) -> global
template <int D, int DB>
OCN_DEV void store_rows(bf16* __restrict__ base, size_t rs, int row, int rows, int lane, const f32x16 (&acc)[DB], float mul) {
    if (row >= rows) return;
    const int lh = lane >> 5;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int d0 = db * 32 + 8 * q4 + 4 * lh;
            if (d0 < D) {
                const bf16x4 v = {f2bf(acc[db][4 * q4] * mul), f2bf(acc[db][4 * q4 + 1] * mul), f2bf(acc[db][4 * q4 + 2] * mul),
                                  f2bf(acc[db][4 * q4 + 3] * mul)};
                *(bf16x4*)(base + (size_t)row * rs + d0) = v;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// forward: grid (B*H, groups of 4 query blocks), 256 threads
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_g_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, float* __restrict__ lse,
                                                         int L, int H, int causal, float scale) {
    constexpr int DP = (D + 31) / 32 * 32, DB = DP / 32, KS = D / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / H, hd = blockIdx.x % H;
    const int C = H * D;
    const size_t rs = (size_t)3 * C;
    const int nblk = (L + 31) / 32, LPAD = nblk * 32;
    const bf16* qbase = qkv + (size_t)b * L * rs + hd * D;
    char* sK = smem;
    char* sV = smem + (size_t)LPAD * DP * 2;
    stage_image<D, DP>(qbase + C, rs, L, LPAD, sK);
    stage_image<D, DP>(qbase + 2 * C, rs, L, LPAD, sV);
    const int qb = blockIdx.y * 4 + wave;
    const int lr = lane & 31, lh = lane >> 5;
    const int query = qb * 32 + lr;
    const int qrow = query < L ? query : L - 1;
    bf16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = frag_rows_g(qbase, rs, qrow, s, lane);
    __syncthreads();
    if (qb >= nblk) return;

    const float sc = scale * LOG2E;
    float m = -1e30f, l = 0.f;
    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) o[db] = zero16();
    const int nkb = causal ? qb + 1 : nblk;
    for (int kb = 0; kb < nkb; ++kb) {
        f32x16 st = zero16();
#pragma unroll
        for (int s = 0; s < KS; ++s) st = mfma32(frag_rows_p<DP>(sK, kb * 32 + lr, s, lane), qf[s], st);
        float mx = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + mfma32_row(r, lane);
            const bool ok = key < L && (!causal || key <= query);
            st[r] = ok ? st[r] * sc : -INFINITY;
            mx = fmaxf(mx, st[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = exp2f(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = exp2f(st[r] - mn);
            ps += st[r];
        }
        l = l * alpha + ps;
        m = mn;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 pf = pack8(st, t);
#pragma unroll
            for (int db = 0; db < DB; ++db) o[db] = mfma32(frag_cols_p<DP>(sV, kb * 32, t, db, lane), pf, o[db]);
        }
    }
    l += __shfl_xor(l, 32, 64);
    store_rows<D, DB>(out + (size_t)b * L * C + hd * D, (size_t)C, query, L, lane, o, 1.0f / l);
    if (query < L && lh == 0) lse[((size_t)b * H + hd) * L + query] = (m + log2f(l)) * LN2;
}

// ------------------------------------------------------------------------------------------------
// backward, kernel 1: dQ for 4 query blocks; writes delta[q]
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_g_bwd_dq_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                            const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                            bf16* __restrict__ dqkv, float* __restrict__ delta, int L, int H,
                                                            int causal, float scale) {
    constexpr int DP = (D + 31) / 32 * 32, DB = DP / 32, KS = D / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / H, hd = blockIdx.x % H;
    const int C = H * D;
    const size_t rs = (size_t)3 * C;
    const int nblk = (L + 31) / 32, LPAD = nblk * 32;
    const bf16* qbase = qkv + (size_t)b * L * rs + hd * D;
    const bf16* dobase = dout + (size_t)b * L * C + hd * D;
    const bf16* obase = out + (size_t)b * L * C + hd * D;
    char* sK = smem;
    char* sV = smem + (size_t)LPAD * DP * 2;
    stage_image<D, DP>(qbase + C, rs, L, LPAD, sK);
    stage_image<D, DP>(qbase + 2 * C, rs, L, LPAD, sV);
    const int qb = blockIdx.y * 4 + wave;
    const int lr = lane & 31, lh = lane >> 5;
    const int query = qb * 32 + lr;
    const int qrow = query < L ? query : L - 1;
    bf16x8 qf[KS], dof[KS];
    float delta_q = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        qf[s] = frag_rows_g(qbase, rs, qrow, s, lane);
        dof[s] = frag_rows_g(dobase, (size_t)C, qrow, s, lane);
        const bf16x8 of = frag_rows_g(obase, (size_t)C, qrow, s, lane);
#pragma unroll
        for (int e = 0; e < 8; ++e) delta_q += bf2f(dof[s][e]) * bf2f(of[e]);
    }
    delta_q += __shfl_xor(delta_q, 32, 64);
    const float lse_q = lse[((size_t)b * H + hd) * L + qrow] * LOG2E;
    if (qb < nblk && query < L && lh == 0) delta[((size_t)b * H + hd) * L + query] = delta_q;
    __syncthreads();
    if (qb >= nblk) return;

    const float sc = scale * LOG2E;
    f32x16 dq[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) dq[db] = zero16();
    const int nkb = causal ? qb + 1 : nblk;
    for (int kb = 0; kb < nkb; ++kb) {
        f32x16 st = zero16(), dp = zero16();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            st = mfma32(frag_rows_p<DP>(sK, kb * 32 + lr, s, lane), qf[s], st);
            dp = mfma32(frag_rows_p<DP>(sV, kb * 32 + lr, s, lane), dof[s], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + mfma32_row(r, lane);
            const bool ok = key < L && query < L && (!causal || key <= query);
            const float p = ok ? exp2f(st[r] * sc - lse_q) : 0.f;
            st[r] = p * (dp[r] - delta_q) * scale;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 dsf = pack8(st, t);
#pragma unroll
            for (int db = 0; db < DB; ++db) dq[db] = mfma32(frag_cols_p<DP>(sK, kb * 32, t, db, lane), dsf, dq[db]);
        }
    }
    store_rows<D, DB>(dqkv + (size_t)b * L * rs + hd * D, rs, query, L, lane, dq, 1.0f);
}

// ------------------------------------------------------------------------------------------------
// backward, kernel 2: dK, dV for 4 key blocks (reads delta written by kernel 1)
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_g_bwd_dkv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             bf16* __restrict__ dqkv, int L, int H, int causal, float scale) {
    constexpr int DP = (D + 31) / 32 * 32, DB = DP / 32, KS = D / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / H, hd = blockIdx.x % H;
    const int C = H * D;
    const size_t rs = (size_t)3 * C;
    const int nblk = (L + 31) / 32, LPAD = nblk * 32;
    const bf16* qbase = qkv + (size_t)b * L * rs + hd * D;
    const bf16* dobase = dout + (size_t)b * L * C + hd * D;
    char* sQ = smem;
    char* sdO = smem + (size_t)LPAD * DP * 2;
    float* sLse = (float*)(smem + (size_t)2 * LPAD * DP * 2);
    float* sDelta = sLse + LPAD;
    stage_image<D, DP>(qbase, rs, L, LPAD, sQ);
    stage_image<D, DP>(dobase, (size_t)C, L, LPAD, sdO);
    for (int q = threadIdx.x; q < LPAD; q += blockDim.x) {
        const int qq = q < L ? q : L - 1;
        sLse[q] = lse[((size_t)b * H + hd) * L + qq] * LOG2E;
        sDelta[q] = delta[((size_t)b * H + hd) * L + qq];
    }
    const int kb = blockIdx.y * 4 + wave;
    const int lr = lane & 31;
    const int key = kb * 32 + lr;
    const int krow = key < L ? key : L - 1;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        kf[s] = frag_rows_g(qbase + C, rs, krow, s, lane);
        vf[s] = frag_rows_g(qbase + 2 * C, rs, krow, s, lane);
    }
    __syncthreads();
    if (kb >= nblk) return;

    const float sc = scale * LOG2E;
    f32x16 dk[DB], dv[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) {
        dk[db] = zero16();
        dv[db] = zero16();
    }
    for (int qb = causal ? kb : 0; qb < nblk; ++qb) {
        f32x16 st = zero16(), dp = zero16();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            st = mfma32(frag_rows_p<DP>(sQ, qb * 32 + lr, s, lane), kf[s], st);
            dp = mfma32(frag_rows_p<DP>(sdO, qb * 32 + lr, s, lane), vf[s], dp);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int query = qb * 32 + mfma32_row(r, lane);
            const bool ok = key < L && query < L && (!causal || key <= query);
            const float p = ok ? exp2f(st[r] * sc - sLse[query]) : 0.f;
            dp[r] = p * (dp[r] - sDelta[query]) * scale;
            st[r] = p;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 pf = pack8(st, t), dsf = pack8(dp, t);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                dv[db] = mfma32(frag_cols_p<DP>(sdO, qb * 32, t, db, lane), pf, dv[db]);
                dk[db] = mfma32(frag_cols_p<DP>(sQ, qb * 32, t, db, lane), dsf, dk[db]);
            }
        }
    }
    bf16* dbase = dqkv + (size_t)b * L * rs + hd * D;
    store_rows<D, DB>(dbase + C, rs, key, L, lane, dk, 1.0f);
    store_rows<D, DB>(dbase + 2 * C, rs, key, L, lane, dv, 1.0f);
}

template <int D>
int lds_bytes(int L, bool with_stats) {
    constexpr int DP = (D + 31) / 32 * 32;
    const int lpad = (L + 31) / 32 * 32;
    return 2 * lpad * DP * 2 + (with_stats ? 2 * lpad * 4 : 0);
}

template <int D>
int launch_fwd(const bf16* qkv, bf16* out, float* lse, int B, int L, int H, int causal, float scale, hipStream_t st) {
    const int lds = lds_bytes<D>(L, false);
    if (lds > 160 * 1024) return 1;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_g_fwd_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int groups = ocn_cdiv(ocn_cdiv(L, 32), 4);
    hipLaunchKernelGGL(attn_g_fwd_kernel<D>, dim3(B * H, groups), dim3(256), lds, st, qkv, out, lse, L, H, causal, scale);
    return 0;
}

template <int D>
int launch_bwd(const bf16* qkv, const bf16* out, const bf16* dout, const float* lse, bf16* dqkv, float* delta, int B, int L, int H,
               int causal, float scale, hipStream_t st) {
    const int lds = lds_bytes<D>(L, true);
    if (lds > 160 * 1024) return 1;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_g_bwd_dq_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_g_bwd_dkv_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int groups = ocn_cdiv(ocn_cdiv(L, 32), 4);
    hipLaunchKernelGGL(attn_g_bwd_dq_kernel<D>, dim3(B * H, groups), dim3(256), lds_bytes<D>(L, false), st, qkv, out, dout, lse, dqkv,
                       delta, L, H, causal, scale);
    hipLaunchKernelGGL(attn_g_bwd_dkv_kernel<D>, dim3(B * H, groups), dim3(256), lds, st, qkv, dout, lse, delta, dqkv, L, H, causal,
                       scale);
    return 0;
}

}  // namespace

// returns 0 = launched, 1 = shape not supported by this path
int ocn_launch_attn_generic_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int D, int causal, float scale,
                                hipStream_t st) {
    switch (D) {
        case 64: return launch_fwd<64>((const bf16*)qkv, (bf16*)out, lse, B, L, H, causal, scale, st);
        case 80: return launch_fwd<80>((const bf16*)qkv, (bf16*)out, lse, B, L, H, causal, scale, st);
        case 96: return launch_fwd<96>((const bf16*)qkv, (bf16*)out, lse, B, L, H, causal, scale, st);
        case 128: return launch_fwd<128>((const bf16*)qkv, (bf16*)out, lse, B, L, H, causal, scale, st);
    }
    return 1;
}

int ocn_launch_attn_generic_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta, int B,
                                int L, int H, int D, int causal, float scale, hipStream_t st) {
    switch (D) {
        case 64: return launch_bwd<64>((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, delta, B, L, H, causal, scale, st);
        case 80: return launch_bwd<80>((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, delta, B, L, H, causal, scale, st);
        case 96: return launch_bwd<96>((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, delta, B, L, H, causal, scale, st);
        case 128: return launch_bwd<128>((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, delta, B, L, H, causal, scale, st);
    }
    return 1;
}
