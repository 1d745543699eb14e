// Generic multi-head self-attention core: any head_dim D in {64, 80, 88, 96, 104, 112, 128} and ANY sequence length -- the path of the
// ViT-H-14 image tower (head_dim 80, 257 tokens; BASELINE config 5), of every head_dim-64 tower beyond 128 tokens (ViT-L-14's 257: config
// 4) and of ViT-g-14 / ViT-bigG-14 / ViT-e-14 (head_dim 88 / 104 / 112), which the head-resident kernels of attention.hip (head_dim 64, a
// whole head in LDS) serve badly or not at all.
//
// Same arithmetic as attention.hip (swapped products S^T = K Q^T, O^T = V^T P^T so that softmax statistics are lane-local, fp32
// softmax in the exp2 domain, causal mask as a predicate, backward recomputes P from the saved LSE).  Residency is what differs
// (round 3: the round-2 version kept both operands of a head resident -- 110 KB of LDS, ONE 4-wave workgroup per CU, every load
// serialised in front of the arithmetic: 69 TFLOP/s on ViT-H-14's forward, 32 % of that model's step):
//   * a workgroup = 4 waves = 4 blocks of 32 queries (forward, dQ) or 32 keys (dK / dV) of ONE (batch, head); its own rows go straight
//     from global memory into MFMA fragments and stay in registers;
//   * the operand every wave needs in full -- K, V (forward, dQ) or Q, dO (+ their LSE / delta) (dK / dV) -- is STREAMED in chunks of 64
//     rows through a two-slot LDS ring: the global loads of chunk c + 1 are issued before the arithmetic on chunk c and land in
//     registers while the MFMAs run, then go to the other slot (one barrier per chunk).  18-35 KB of LDS per workgroup instead of
//     110: three to four workgroups per CU, each overlapping its own loads with its own arithmetic;
//   * LDS rows are padded by 16 bytes (pitch 2 D + 16): the ds_read_b128 fragment reads of 16 consecutive rows hit 16 distinct
//     bank quads for every supported D; transposed operands (V^T, K^T, Q^T, dO^T) come from the same image by ds_read_b64_tr_b16;
//   * D = 80 needs no padding of the contraction (5 MFMA k-steps of 16); as an OUTPUT dimension it is covered by three 32-wide blocks
//     whose last 16 columns multiply whatever follows the row in LDS and are never stored;
//   * D = 88 / 104 (8 short of a k-step): the contraction runs over DP = 96 / 112 with the missing 8 d's ZERO on both sides -- the
//     16-byte piece behind every LDS row is zeroed once (the chunk writes never touch it), and a wave's own rows, which come straight
//     from global memory, get a zero fragment half instead of the neighbouring head's columns;
//   * workgroups of one head are adjacent in the grid (they share K / V resp. Q / dO through the Infinity Cache);
//   * the backward is three launches (dQ, dV, dK) that exchange delta[q] = sum_d dO O through a caller-provided fp32 workspace.
#include "ocn_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int CH = 64;  // rows per streamed chunk (two 32-row blocks)

template <int D>
struct Geo {
    static_assert(D % 8 == 0, "head_dim must be a multiple of 8 (16-byte pieces)");
    static constexpr int DP = (D + 15) / 16 * 16;   // contraction length: D rounded up to whole MFMA k-steps (zero padded)
    static constexpr bool PADDED = DP != D;
    static constexpr int CPR = D / 8;               // 16-byte pieces per row (the ones that exist in global memory)
    static constexpr int PITCH = DP * 2 + 16;       // LDS row pitch in bytes
    static constexpr int DB = (D + 31) / 32;        // 32-wide output blocks over d
    static constexpr int KS = DP / 16;              // MFMA k-steps over d
    static constexpr int PIECES = CH * CPR;         // 16-byte pieces per chunk image
    static constexpr int PPT = (PIECES + 255) / 256;  // pieces per thread
    static constexpr int BUF = CH * PITCH;          // bytes per chunk image
};

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

// Marks a fragment that was loaded from global memory BEFORE the chunk loop as complete: without it hipcc's wait-count pass, which
// merges the first iteration (fragment loads still outstanding) with the later ones, puts `s_waitcnt vmcnt(4..0)` in front of the first
// MFMAs of EVERY iteration -- and vmcnt retires in order, so those waits also drain the chunk prefetch that was issued a few
// instructions earlier (seen in the ISA of the first build: every load serialised in front of the arithmetic again).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
OCN_DEV void settle(bf16x8& f) {
    u32x4_t v = __builtin_bit_cast(u32x4_t, f);
    asm volatile("" : "+v"(v));
    f = __builtin_bit_cast(bf16x8, v);
}
OCN_DEV void settle(float& x) { asm volatile("" : "+v"(x)); }

// rows [row0, row0 + 64) x D of one head's column block (row stride `rs` elements) -> registers; rows beyond `rows` are copies of the
// last valid row (finite data; the arithmetic masks them).  STRAIGHT-LINE code on purpose: with the loads inside exec-masked branches
// (a guard per 16-byte piece, an `if (more)` around the prefetch) hipcc puts `s_waitcnt vmcnt(0)` in front of every single load -- six
// serialised memory round trips per chunk.  A thread whose piece index lies beyond the image loads (and later writes) the LAST piece
// again: the same bytes to the same address as their owner, which is harmless.
template <int D>
OCN_DEV void load_chunk(const bf16* __restrict__ base, size_t rs, int row0, int rows, bf16x8 (&r)[Geo<D>::PPT]) {
    using G = Geo<D>;
#pragma unroll
    for (int j = 0; j < G::PPT; ++j) {
        int i = threadIdx.x + j * 256;
        if (G::PIECES % 256 != 0) i = i < G::PIECES ? i : G::PIECES - 1;
        const int row = i / G::CPR, c = i - row * G::CPR;
        int gr = row0 + row;
        gr = gr < rows ? gr : rows - 1;
        r[j] = *(const bf16x8*)(base + (size_t)gr * rs + c * 8);
    }
}
template <int D>
OCN_DEV void write_chunk(char* sT, const bf16x8 (&r)[Geo<D>::PPT]) {
    using G = Geo<D>;
#pragma unroll
    for (int j = 0; j < G::PPT; ++j) {
        int i = threadIdx.x + j * 256;
        if (G::PIECES % 256 != 0) i = i < G::PIECES ? i : G::PIECES - 1;
        const int row = i / G::CPR, c = i - row * G::CPR;
        *(bf16x8*)(sT + row * G::PITCH + c * 16) = r[j];
    }
}

// D = 88 / 104: the 16-byte piece behind the D columns of every row of `nimg` chunk images (`stride` bytes apart) <- 0, once per kernel
template <int D>
OCN_DEV void zero_pad_pieces(char* img0, int stride, int nimg) {
    using G = Geo<D>;
    if constexpr (G::PADDED) {
        const bf16x8 z = __builtin_bit_cast(bf16x8, (u32x4_t){0u, 0u, 0u, 0u});
        for (int i = threadIdx.x; i < nimg * CH; i += 256) *(bf16x8*)(img0 + (i / CH) * stride + (i % CH) * G::PITCH + G::CPR * 16) = z;
    }
}

// d-contiguous operand: row `row` of a chunk image, k-step s (16 of the d's), this lane's 8 d's
template <int PITCH>
OCN_DEV bf16x8 frag_rows_p(const char* sT, int row, int s, int lane) {
    return *(const bf16x8*)(sT + row * PITCH + ((s * 2 + (lane >> 5)) << 4));
}

// the same fragment taken straight from global memory (row-major, `rs` elements per row).  D = 88 / 104: the upper half of the last
// k-step lies behind the head's columns -- zero, loaded from the last piece that exists (never past the row)
template <int D>
OCN_DEV bf16x8 frag_rows_g(const bf16* __restrict__ base, size_t rs, int row, int s, int lane) {
    using G = Geo<D>;
    int piece = s * 2 + (lane >> 5);
    if constexpr (G::PADDED) {
        const bool beyond = piece >= G::CPR;
        piece = beyond ? G::CPR - 1 : piece;
        const bf16x8 v = *(const bf16x8*)(base + (size_t)row * rs + piece * 8);
        const bf16x8 z = __builtin_bit_cast(bf16x8, (u32x4_t){0u, 0u, 0u, 0u});
        return beyond ? z : v;
    }
    return *(const bf16x8*)(base + (size_t)row * rs + piece * 8);
}

// transposed operand: A[i = d (dblk*32 + lane&31)][k-slots <-> rows rbase + 16t + 8(e>>2) + 4h + (e&3)]  (cf. attention.hip)
template <int PITCH>
OCN_DEV bf16x8 frag_cols_p(const char* sT, int rbase, int t, int dblk, int lane) {
    const int i = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
    const int chunk = dblk * 4 + g * 2 + ((i & 3) >> 1);
    const int r0 = rbase + 16 * t + 4 * h + (i >> 2);
    const int r1 = r0 + 8;
    const s16x4 lo = lds_read_tr16((const OCN_LDS void*)(sT + r0 * PITCH + (chunk << 4) + (i & 1) * 8));
    const s16x4 hi = lds_read_tr16((const OCN_LDS void*)(sT + r1 * PITCH + (chunk << 4) + (i & 1) * 8));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// Workgroup id -> (head, group of 4 blocks).  The `groups` workgroups of one head stream the SAME K / V (resp. Q / dO): they get ids that
// are congruent modulo 8 -- hardware places block b on XCD b % 8, so they share that XCD's L2 -- and lie within 8 * groups of each
// other, so they run at the same time.  Heads are taken eight at a time (the grid is rounded up; surplus workgroups leave).
OCN_DEV bool head_group(int bid, int groups, int BH, int& bh, int& grp) {
    const int per8 = groups * 8;
    const int super = bid / per8, within = bid - super * per8;
    grp = within >> 3;
    bh = super * 8 + (within & 7);
    return bh >= BH;
}

// a wave's [32 rows] x [D] result (acc[dblk]: lane <-> row lr, registers 4*q4.. <-> d = dblk*32 + 8*q4 + 4*lh ..) -> global
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
// forward: grid = B*H heads x groups of 4 query blocks (groups fastest), 256 threads
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 2) void attn_s_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, float* __restrict__ lse,
                                                         int L, int H, int BH, int groups, int causal, float scale) {
    using G = Geo<D>;
    constexpr int DB = G::DB, KS = G::KS, P = G::PITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 slots][K | V][CH][PITCH]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bh, grp;
    if (head_group(blockIdx.x, groups, BH, bh, grp)) return;  // (before any barrier)
    const int b = bh / H, hd = bh - b * H;
    const int C = H * D;
    const size_t rs = (size_t)3 * C;
    const int nblk = (L + 31) / 32;
    const bf16* qbase = qkv + (size_t)b * L * rs + hd * D;
    const bf16 *kbase = qbase + C, *vbase = qbase + 2 * C;
    const int qb = grp * 4 + wave;
    const int lr = lane & 31, lh = lane >> 5;
    const int query = qb * 32 + lr;
    const int qrow = query < L ? query : L - 1;
    bf16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = frag_rows_g<D>(qbase, rs, qrow, s, lane);
    // key blocks this WORKGROUP walks (causal: up to its last query block) and this wave needs
    const int wg_nkb = causal ? min(nblk, grp * 4 + 4) : nblk;
    const int my_nkb = qb >= nblk ? 0 : (causal ? qb + 1 : nblk);
    const int nch = (wg_nkb + 1) / 2;
    bf16x8 rk[G::PPT], rv[G::PPT];
    load_chunk<D>(kbase, rs, 0, L, rk);
    load_chunk<D>(vbase, rs, 0, L, rv);
    zero_pad_pieces<D>(smem, G::BUF, 4);
    write_chunk<D>(smem, rk);
    write_chunk<D>(smem + G::BUF, rv);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) settle(qf[s]);

    const float sc = scale * LOG2E;
    float m = -1e30f, l = 0.f;
    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) o[db] = zero16();
    for (int c = 0; c < nch; ++c) {
        const char* sK = smem + (c & 1) * 2 * G::BUF;
        const char* sV = sK + G::BUF;
        // the next chunk (the last one once more at the end: unconditional, see load_chunk): in flight under the arithmetic below
        const int cn = c + 1 < nch ? c + 1 : c;
        load_chunk<D>(kbase, rs, cn * CH, L, rk);
        load_chunk<D>(vbase, rs, cn * CH, L, rv);
#pragma unroll
        for (int kbi = 0; kbi < 2; ++kbi) {
            const int kb = c * 2 + kbi;
            if (kb < my_nkb) {
                f32x16 st = zero16();
#pragma unroll
                for (int s = 0; s < KS; ++s) st = mfma32(frag_rows_p<P>(sK, kbi * 32 + lr, s, lane), qf[s], st);
                float mx = -1e30f;
                // only the last key block (keys >= L) and, under the causal mask, the diagonal block have anything to mask: every other
                // block skips the three compares and the select per element (wave-uniform branch)
                const bool edge = (kb == nblk - 1 && (L & 31) != 0) || (causal && kb == qb);
                if (edge) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kb * 32 + mfma32_row(r, lane);
                        const bool ok = key < L && (!causal || key <= query);
                        st[r] = ok ? st[r] * sc : -INFINITY;
                        mx = fmaxf(mx, st[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        st[r] *= sc;
                        mx = fmaxf(mx, st[r]);
                    }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mn = fmaxf(m, mx);
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[r] = fast_exp2(st[r] - mn);
                    ps += st[r];
                }
                // the running maximum moves in the first blocks and rarely afterwards: when it stayed put for every query of the wave the
                // rescale factor is exactly 1 and the 16 * DB multiplies of the output accumulators are skipped
                if (__builtin_amdgcn_ballot_w64(mn != m) != 0) {
                    const float alpha = fast_exp2(m - mn);
                    l *= alpha;
#pragma unroll
                    for (int db = 0; db < DB; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
                }
                l += ps;
                m = mn;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 pf = pack8(st, t);
#pragma unroll
                    for (int db = 0; db < DB; ++db) o[db] = mfma32(frag_cols_p<P>(sV, kbi * 32, t, db, lane), pf, o[db]);
                }
            }
        }
        {  // the other slot: every wave finished reading it before the barrier that ended the previous iteration
            char* nK = smem + ((c + 1) & 1) * 2 * G::BUF;
            write_chunk<D>(nK, rk);
            write_chunk<D>(nK + G::BUF, rv);
        }
        __syncthreads();
    }
    if (qb >= nblk) return;
    l += __shfl_xor(l, 32, 64);
    store_rows<D, DB>(out + (size_t)b * L * C + hd * D, (size_t)C, query, L, lane, o, 1.0f / l);
    if (query < L && lh == 0) lse[((size_t)b * H + hd) * L + query] = (m + log2f(l)) * LN2;
}

// ------------------------------------------------------------------------------------------------
// backward, kernel 1: dQ for 4 query blocks (K, V streamed); writes delta[q]
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 2) void attn_s_bwd_dq_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                            const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                            bf16* __restrict__ dqkv, float* __restrict__ delta, int L, int H, int BH, int groups,
                                                            int causal, float scale) {
    using G = Geo<D>;
    constexpr int DB = G::DB, KS = G::KS, P = G::PITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bh, grp;
    if (head_group(blockIdx.x, groups, BH, bh, grp)) return;  // (before any barrier)
    const int b = bh / H, hd = bh - b * H;
    const int C = H * D;
    const size_t rs = (size_t)3 * C;
    const int nblk = (L + 31) / 32;
    const bf16* qbase = qkv + (size_t)b * L * rs + hd * D;
    const bf16 *kbase = qbase + C, *vbase = qbase + 2 * C;
    const bf16* dobase = dout + (size_t)b * L * C + hd * D;
    const bf16* obase = out + (size_t)b * L * C + hd * D;
    const int qb = grp * 4 + wave;
    const int lr = lane & 31, lh = lane >> 5;
    const int query = qb * 32 + lr;
    const int qrow = query < L ? query : L - 1;
    bf16x8 qf[KS], dof[KS];
    float delta_q = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        qf[s] = frag_rows_g<D>(qbase, rs, qrow, s, lane);
        dof[s] = frag_rows_g<D>(dobase, (size_t)C, qrow, s, lane);
        const bf16x8 of = frag_rows_g<D>(obase, (size_t)C, qrow, s, lane);
#pragma unroll
        for (int e = 0; e < 8; ++e) delta_q += bf2f(dof[s][e]) * bf2f(of[e]);
    }
    delta_q += __shfl_xor(delta_q, 32, 64);
    const float lse_q = lse[((size_t)b * H + hd) * L + qrow] * LOG2E;
    if (qb < nblk && query < L && lh == 0) delta[((size_t)b * H + hd) * L + query] = delta_q;
    const int wg_nkb = causal ? min(nblk, grp * 4 + 4) : nblk;
    const int my_nkb = qb >= nblk ? 0 : (causal ? qb + 1 : nblk);
    const int nch = (wg_nkb + 1) / 2;
    bf16x8 rk[G::PPT], rv[G::PPT];
    load_chunk<D>(kbase, rs, 0, L, rk);
    load_chunk<D>(vbase, rs, 0, L, rv);
    zero_pad_pieces<D>(smem, G::BUF, 4);
    write_chunk<D>(smem, rk);
    write_chunk<D>(smem + G::BUF, rv);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        settle(qf[s]);
        settle(dof[s]);
    }
    float lse_s = lse_q, delta_s = delta_q;
    settle(lse_s);
    settle(delta_s);

    const float sc = scale * LOG2E;
    f32x16 dq[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) dq[db] = zero16();
    for (int c = 0; c < nch; ++c) {
        const char* sK = smem + (c & 1) * 2 * G::BUF;
        const char* sV = sK + G::BUF;
        const int cn = c + 1 < nch ? c + 1 : c;
        load_chunk<D>(kbase, rs, cn * CH, L, rk);
        load_chunk<D>(vbase, rs, cn * CH, L, rv);
#pragma unroll
        for (int kbi = 0; kbi < 2; ++kbi) {
            const int kb = c * 2 + kbi;
            if (kb < my_nkb) {
                f32x16 st = zero16(), dp = zero16();
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    st = mfma32(frag_rows_p<P>(sK, kbi * 32 + lr, s, lane), qf[s], st);
                    dp = mfma32(frag_rows_p<P>(sV, kbi * 32 + lr, s, lane), dof[s], dp);
                }
                // masks only where they can bite: the last key block (keys >= L are copies of the last row) and the causal diagonal; rows of
                // queries >= L are dropped by the store
                const bool edge = (kb == nblk - 1 && (L & 31) != 0) || (causal && kb == qb);
                if (edge) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kb * 32 + mfma32_row(r, lane);
                        const bool ok = key < L && (!causal || key <= query);
                        const float e = fast_exp2(st[r] * sc - lse_s);
                        st[r] = (ok ? e : 0.f) * (dp[r] - delta_s) * scale;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[r] = fast_exp2(st[r] * sc - lse_s) * (dp[r] - delta_s) * scale;
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 dsf = pack8(st, t);
#pragma unroll
                    for (int db = 0; db < DB; ++db) dq[db] = mfma32(frag_cols_p<P>(sK, kbi * 32, t, db, lane), dsf, dq[db]);
                }
            }
        }
        {
            char* nK = smem + ((c + 1) & 1) * 2 * G::BUF;
            write_chunk<D>(nK, rk);
            write_chunk<D>(nK + G::BUF, rv);
        }
        __syncthreads();
    }
    store_rows<D, DB>(dqkv + (size_t)b * L * rs + hd * D, rs, query, L, lane, dq, 1.0f);
}

// ------------------------------------------------------------------------------------------------
// backward, kernels 2 and 3: dV (WHAT = 1) resp. dK (WHAT = 2) for 4 key blocks (Q, dO and their LSE / delta streamed; delta written by
// kernel 1).  One launch each: with both accumulator sets (2 x DB x 16 registers) next to the K and V fragments and the staging registers
// a wave needs the whole 256-register budget (one 4-wave workgroup per CU; spills at head_dim 96 / 128) -- apart, the dV kernel fits three
// workgroups per CU and the dK kernel two.  The price: S = K Q^T is evaluated in both (5 of 27 MFMAs per block pair at head_dim 80) and
// Q / dO are streamed twice (mostly out of the L2 / Infinity Cache: the workgroups of a head run together on one XCD).
// ------------------------------------------------------------------------------------------------
template <int D, int WHAT>
__global__ __launch_bounds__(256, 2) void attn_s_bwd_dkv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                bf16* __restrict__ dqkv, int L, int H, int BH, int groups, int causal, float scale) {
    using G = Geo<D>;
    constexpr int DB = G::DB, KS = G::KS, P = G::PITCH;
    constexpr bool DV = WHAT == 1;
    constexpr int SLOT = 2 * G::BUF + 2 * CH * 4;  // Q | dO | lse[64] | delta[64]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bh, grp;
    if (head_group(blockIdx.x, groups, BH, bh, grp)) return;  // (before any barrier)
    const int b = bh / H, hd = bh - b * H;
    const int C = H * D;
    const size_t rs = (size_t)3 * C;
    const int nblk = (L + 31) / 32;
    const bf16* qbase = qkv + (size_t)b * L * rs + hd * D;
    const bf16* dobase = dout + (size_t)b * L * C + hd * D;
    const float* lse_h = lse + ((size_t)b * H + hd) * L;
    const float* delta_h = delta + ((size_t)b * H + hd) * L;
    const int kb = grp * 4 + wave;
    const int lr = lane & 31;
    const int key = kb * 32 + lr;
    const int krow = key < L ? key : L - 1;
    bf16x8 kf[KS], vf[DV ? 1 : KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        kf[s] = frag_rows_g<D>(qbase + C, rs, krow, s, lane);
        if constexpr (!DV) vf[s] = frag_rows_g<D>(qbase + 2 * C, rs, krow, s, lane);
    }
    // query chunks this workgroup walks: causal -> from its first key block on
    const int c0 = causal ? (grp * 4) / 2 : 0;
    const int nch = (nblk + 1) / 2;
    bf16x8 rq[G::PPT], rd[G::PPT];
    float rs_stat = 0.f;  // thread t: (t & 64) == 0 -> lse of query (t & 63) of the chunk, else its delta (the upper 128 threads duplicate)
    const float* stat_h = (threadIdx.x & CH) ? delta_h : lse_h;
    const float stat_mul = (threadIdx.x & CH) ? 1.0f : LOG2E;
    auto load_stats = [&](int row0) {
        int q = row0 + (threadIdx.x & (CH - 1));
        q = q < L ? q : L - 1;
        rs_stat = stat_h[q] * stat_mul;
    };
    auto write_stats = [&](char* slot) { ((float*)(slot + 2 * G::BUF))[threadIdx.x & (2 * CH - 1)] = rs_stat; };
    load_chunk<D>(qbase, rs, c0 * CH, L, rq);
    load_chunk<D>(dobase, (size_t)C, c0 * CH, L, rd);
    load_stats(c0 * CH);
    zero_pad_pieces<D>(smem, G::BUF, 2);
    zero_pad_pieces<D>(smem + SLOT, G::BUF, 2);
    write_chunk<D>(smem, rq);
    write_chunk<D>(smem + G::BUF, rd);
    write_stats(smem);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        settle(kf[s]);
        if constexpr (!DV) settle(vf[s]);
    }

    const float sc = scale * LOG2E;
    f32x16 acc[DB];  // dV or dK
#pragma unroll
    for (int db = 0; db < DB; ++db) acc[db] = zero16();
    for (int c = c0; c < nch; ++c) {
        const char* slot = smem + ((c - c0) & 1) * SLOT;
        const char* sQ = slot;
        const char* sdO = slot + G::BUF;
        const float* sLse = (const float*)(slot + 2 * G::BUF);
        const float* sDelta = sLse + CH;
        const int cn = c + 1 < nch ? c + 1 : c;
        load_chunk<D>(qbase, rs, cn * CH, L, rq);
        load_chunk<D>(dobase, (size_t)C, cn * CH, L, rd);
        load_stats(cn * CH);
#pragma unroll
        for (int qbi = 0; qbi < 2; ++qbi) {
            const int qb = c * 2 + qbi;
            if (kb < nblk && qb < nblk && (!causal || qb >= kb)) {
                f32x16 st = zero16(), dp = zero16();
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    st = mfma32(frag_rows_p<P>(sQ, qbi * 32 + lr, s, lane), kf[s], st);
                    if constexpr (!DV) dp = mfma32(frag_rows_p<P>(sdO, qbi * 32 + lr, s, lane), vf[s], dp);
                }
                // masks only where they can bite: the last query block (queries >= L are copies of the last row) and the causal diagonal; rows
                // of keys >= L are dropped by the store.  The statistics are read unconditionally (under `ok ? ... : 0` hipcc branches around
                // every element).  st <- P (dV) or dS = P (dP - delta) scale (dK)
                const bool edge = (qb == nblk - 1 && (L & 31) != 0) || (causal && qb == kb);
                if (edge) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ql = qbi * 32 + mfma32_row(r, lane);  // query inside the chunk
                        const int query = c * CH + ql;
                        const bool ok = query < L && (!causal || key <= query);
                        const float e = fast_exp2(st[r] * sc - sLse[ql]);
                        const float p = ok ? e : 0.f;
                        st[r] = DV ? p : p * (dp[r] - sDelta[ql]) * scale;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ql = qbi * 32 + mfma32_row(r, lane);
                        const float p = fast_exp2(st[r] * sc - sLse[ql]);
                        st[r] = DV ? p : p * (dp[r] - sDelta[ql]) * scale;
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 f = pack8(st, t);
#pragma unroll
                    for (int db = 0; db < DB; ++db) acc[db] = mfma32(frag_cols_p<P>(DV ? sdO : sQ, qbi * 32, t, db, lane), f, acc[db]);
                }
            }
        }
        {
            char* nslot = smem + ((c + 1 - c0) & 1) * SLOT;
            write_chunk<D>(nslot, rq);
            write_chunk<D>(nslot + G::BUF, rd);
            write_stats(nslot);
        }
        __syncthreads();
    }
    if (kb >= nblk) return;
    bf16* dbase = dqkv + (size_t)b * L * rs + hd * D;
    store_rows<D, DB>(dbase + (DV ? 2 * C : C), rs, key, L, lane, acc, 1.0f);
}

template <int D>
int launch_fwd(const bf16* qkv, bf16* out, float* lse, int B, int L, int H, int causal, float scale, hipStream_t st) {
    using G = Geo<D>;
    const int lds = 4 * G::BUF + 64;  // + slack: the transposed reads of the last row's unused d-columns reach 16 bytes past an image
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_s_fwd_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int groups = ocn_cdiv(ocn_cdiv(L, 32), 4);
    const long grid = (long)ocn_cdiv((long)B * H, 8) * 8 * groups;
    if (grid > 0x7fffffffL) return 1;
    hipLaunchKernelGGL(attn_s_fwd_kernel<D>, dim3((unsigned)grid), dim3(256), lds, st, qkv, out, lse, L, H, B * H, groups, causal, scale);
    return 0;
}

template <int D>
int launch_bwd(const bf16* qkv, const bf16* out, const bf16* dout, const float* lse, bf16* dqkv, float* delta, int B, int L, int H,
               int causal, float scale, hipStream_t st) {
    using G = Geo<D>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_s_bwd_dq_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_s_bwd_dkv_kernel<D, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_s_bwd_dkv_kernel<D, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int groups = ocn_cdiv(ocn_cdiv(L, 32), 4);
    const long grid = (long)ocn_cdiv((long)B * H, 8) * 8 * groups;
    if (grid > 0x7fffffffL) return 1;
    hipLaunchKernelGGL(attn_s_bwd_dq_kernel<D>, dim3((unsigned)grid), dim3(256), 4 * G::BUF + 64, st, qkv, out, dout, lse, dqkv, delta, L, H, B * H, groups,
                       causal, scale);
    hipLaunchKernelGGL((attn_s_bwd_dkv_kernel<D, 1>), dim3((unsigned)grid), dim3(256), 2 * (2 * G::BUF + 2 * CH * 4) + 64, st, qkv, dout, lse, delta, dqkv, L,
                       H, B * H, groups, causal, scale);
    hipLaunchKernelGGL((attn_s_bwd_dkv_kernel<D, 2>), dim3((unsigned)grid), dim3(256), 2 * (2 * G::BUF + 2 * CH * 4) + 64, st, qkv, dout, lse, delta, dqkv, L,
                       H, B * H, groups, causal, scale);
    return 0;
}

}  // namespace

// returns 0 = launched, 1 = shape not supported by this path
int ocn_launch_attn_generic_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int D, int causal, float scale,
                                hipStream_t st) {
    switch (D) {
        case 64: return launch_fwd<64>((const bf16*)qkv, (bf16*)out, lse, B, L, H, causal, scale, st);
        case 80: return launch_fwd<80>((const bf16*)qkv, (bf16*)out, lse, B, L, H, causal, scale, st);
        case 88: return launch_fwd<88>((const bf16*)qkv, (bf16*)out, lse, B, L, H, causal, scale, st);
        case 104: return launch_fwd<104>((const bf16*)qkv, (bf16*)out, lse, B, L, H, causal, scale, st);
        case 112: return launch_fwd<112>((const bf16*)qkv, (bf16*)out, lse, B, L, H, causal, scale, st);
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
        case 88: return launch_bwd<88>((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, delta, B, L, H, causal, scale, st);
        case 104: return launch_bwd<104>((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, delta, B, L, H, causal, scale, st);
        case 112: return launch_bwd<112>((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, delta, B, L, H, causal, scale, st);
        case 96: return launch_bwd<96>((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, delta, B, L, H, causal, scale, st);
        case 128: return launch_bwd<128>((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, delta, B, L, H, causal, scale, st);
    }
    return 1;
}
