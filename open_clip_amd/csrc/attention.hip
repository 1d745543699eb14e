// Multi-head self-attention core for the CLIP towers (sequence 50 / 77 / 257, head_dim 64) on gfx950.
//
// One workgroup per (batch, head); one wave per 32-query block.  K and V (and, in the backward, Q and dO)
// of the head are staged once into LDS by LDS-DMA as row-major [L_pad][64] bf16 images (128-byte rows,
// chunk-swizzled, see swz_nt) and are consumed two ways from the same image:
//   * d-contiguous MFMA operands by ds_read_b128   (K in S^T = K Q^T,  V in dP^T = V dO^T, ...)
//   * key-/query-contiguous (transposed) operands by ds_read_b64_tr_b16   (V^T in O^T = V^T P^T, K^T, Q^T, dO^T)
// All products are computed "swapped" (S^T = K Q^T, O^T = V^T P^T) so that a lane owns one query (forward,
// dQ) or one key (dK/dV) and the softmax statistics stay lane-local: the row max / sum need one cross-half
// exchange, and the C-layout registers of S^T are directly the B operand of the second product
// (k-slot (h,e) of step t <-> accumulator register 8t+e <-> key 32kb + 16t + 8(e>>2) + 4h + (e&3)).
// Softmax in fp32 (exp2 domain); the causal mask of the text tower is a predicate, not a tensor.
// Global traffic is full 128-byte lines both ways: every input tile arrives by LDS-DMA, and every output tile (O, dQ, dK,
// dV) is transposed through LDS (the wave's own rows of an input image it no longer needs) and leaves as 16-byte stores
// covering 8 whole rows per instruction -- a row-per-lane epilogue of 8-byte pieces is store-ISSUE bound (32 partial
// lines per instruction; MI355X_MICROARCH.md "attention epilogue store tail").
#include "ocn_common.h"

namespace {

OCN_DEV int swz_nt(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

}  // namespace
extern int g_ocn_tuning[16];
namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// Rows of sequence b inside the [rows, 3C] qkv matrix.  Dense batches: Lmax rows each.  Packed ("varlen") batches: only the first
// seq_off[b+1] - seq_off[b] tokens of a sequence exist (the text tower drops everything behind the pooled EOT token: under the
// causal mask those positions cannot influence it -- reference transformer.py:941-944 pools x[argmax], :1716-1722 is the mask).
struct SeqSpan {
    size_t row0;
    int len;
};
OCN_DEV SeqSpan seq_span(const int32_t* __restrict__ seq_off, int b, int Lmax) {
    if (!seq_off) return {(size_t)b * Lmax, Lmax};
    const int o = seq_off[b];
    return {(size_t)o, seq_off[b + 1] - o};
}

// Packed batches are launched in BUCKETS of equal block count (ocn_seq_pack_plan's `order`: sequence ids grouped by ceil(len / 32)):
// a workgroup then has exactly as many waves and LDS rows as its sequence needs, instead of the longest sequence's -- these kernels
// are latency-bound, their rate is the number of resident workgroups.  order == nullptr: workgroup i handles sequence i.
OCN_DEV int seq_index(const int32_t* __restrict__ order, int order_off, int i) { return order ? order[order_off + i] : i; }

// stage rows [0, LP) (clamped to L-1) of one head's 64-wide column block into LDS; all waves cooperate.  LP = the sequence's own
// length rounded up to 32: a packed (varlen) batch never touches the 32-row blocks behind it.
// NTL: non-temporal policy for the head's rows (read by this workgroup only) -- measured no faster than the default policy
// (profiles/r01_attn_cache_policy.txt), kept as developer knob 9 = 2 of the forward
template <bool NTL = false>
OCN_DEV void stage_head(const bf16* __restrict__ base, size_t row_stride, int L, int LP, char* sT, int wave, int nwaves, int lane) {
    for (int seg = wave; seg < LP / 8; seg += nwaves) {
        const int r = seg * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz_nt(r);
        const int gr = r < L ? r : L - 1;
        __builtin_amdgcn_global_load_lds((const OCN_GLB void*)(base + (size_t)gr * row_stride + c * 8), (OCN_LDS void*)(sT + seg * 1024), 16, 0,
                                         NTL ? 2 : 0);
    }
}

// d-contiguous operand: row `row` of the image, k-step s (16 of the 64 d's), this lane's 8 d's
OCN_DEV bf16x8 frag_rows(const char* sT, int row, int s, int lane) {
    const int c = (s * 2 + (lane >> 5)) ^ swz_nt(row);
    return *(const bf16x8*)(sT + row * 128 + (c << 4));
}

// transposed operand: A[i = d (dblk*32 + lane&31)][k-slots <-> rows rbase + 16t + 8(e>>2) + 4h + (e&3)]
OCN_DEV bf16x8 frag_cols(const char* sT, int rbase, int t, int dblk, int lane) {
    const int i = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
    const int chunk = dblk * 4 + g * 2 + ((i & 3) >> 1);
    const int r0 = rbase + 16 * t + 4 * h + (i >> 2);
    const int r1 = r0 + 8;
    const s16x4 lo = lds_read_tr16((const OCN_LDS void*)(sT + r0 * 128 + ((chunk ^ swz_nt(r0)) << 4) + (i & 1) * 8));
    const s16x4 hi = lds_read_tr16((const OCN_LDS void*)(sT + r1 * 128 + ((chunk ^ swz_nt(r1)) << 4) + (i & 1) * 8));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

OCN_DEV bf16x8 pack8(const f32x16& p, int t) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(p[8 * t + e]);
    return o;
}

OCN_DEV f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// A wave's 32 x 64 result tile (a0: d 0..31, a1: d 32..63; lane <-> row lr, registers 4*q4.. <-> d = 8*q4 + 4*lh ..) goes to
// rows [row0, row0+32) of the LDS image `img` (16-byte chunks XOR-swizzled by row & 7), then out as whole rows.
// Only this wave touches those image rows from here on.
OCN_DEV void stage_tile(char* img, int row0, int lane, const f32x16& a0, const f32x16& a1, float mul) {
    const int lr = lane & 31, lh = lane >> 5;
    char* rowp = img + (row0 + lr) * 128 + lh * 8;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const bf16x4 v0 = {f2bf(a0[4 * q4] * mul), f2bf(a0[4 * q4 + 1] * mul), f2bf(a0[4 * q4 + 2] * mul), f2bf(a0[4 * q4 + 3] * mul)};
        const bf16x4 v1 = {f2bf(a1[4 * q4] * mul), f2bf(a1[4 * q4 + 1] * mul), f2bf(a1[4 * q4 + 2] * mul), f2bf(a1[4 * q4 + 3] * mul)};
        *(bf16x4*)(rowp + ((q4 ^ (lr & 7)) << 4)) = v0;
        *(bf16x4*)(rowp + (((4 + q4) ^ (lr & 7)) << 4)) = v1;
    }
}
OCN_DEV void flush_tile(const char* img, int row0, int lane, bf16* base, size_t row_stride, int L) {
    const int r8 = lane >> 3, c = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = row0 + it * 8 + r8;
        const bf16x8 v = *(const bf16x8*)(img + row * 128 + ((c ^ (row & 7)) << 4));
        if (row < L) *(bf16x8*)(base + (size_t)row * row_stride + c * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// (three waves per SIMD asked for: with the whole 512-register file on offer hipcc keeps the MFMA accumulators in AGPRs and moves every one of
// them to a VGPR and back around the softmax arithmetic -- 144 v_accvgpr moves in this kernel)
template <int MAXT, bool NTL>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(3))) void attn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                         float* __restrict__ lse, const int32_t* __restrict__ seq_off, int Lmax, int H,
                                                         int causal, float scale, const int32_t* __restrict__ order, int order_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int LP = nwaves * 32;
    const int hd = blockIdx.x % H;
    const int b = seq_index(order, order_off, blockIdx.x / H);
    const int C = H * 64;
    const size_t rs = (size_t)3 * C;
    const SeqSpan sp = seq_span(seq_off, b, Lmax);  // dense: rows b*Lmax .. +Lmax; packed: rows seq_off[b] .. seq_off[b+1]
    const int L = sp.len, nb = (L + 31) >> 5, LPe = nb * 32;
    const bf16* qbase = qkv + sp.row0 * rs + hd * 64;
    char* sQ = smem;
    char* sK = smem + LP * 128;
    char* sV = smem + 2 * LP * 128;
    stage_head<NTL>(qbase, rs, L, LPe, sQ, wave, nwaves, lane);
    stage_head<NTL>(qbase + C, rs, L, LPe, sK, wave, nwaves, lane);
    stage_head<NTL>(qbase + 2 * C, rs, L, LPe, sV, wave, nwaves, lane);

    const int qb = wave;
    const int lr = lane & 31, lh = lane >> 5;
    const int query = qb * 32 + lr;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (qb >= nb) return;  // a query block behind the end of a short packed sequence (no barrier follows)
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = frag_rows(sQ, query, s, lane);  // rows >= L are copies of row L-1 (stage_head clamps)

    const float sc = scale * LOG2E;
    float m = -1e30f, l = 0.f;
    f32x16 o0 = zero16(), o1 = zero16();
    const int nkb = causal ? qb + 1 : nb;
    for (int kb = 0; kb < nkb; ++kb) {
        f32x16 st = zero16();
#pragma unroll
        for (int s = 0; s < 4; ++s) st = mfma32(frag_rows(sK, kb * 32 + lr, s, lane), qf[s], st);
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
        const float alpha = fast_exp2(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = fast_exp2(st[r] - mn);
            ps += st[r];
        }
        l = l * alpha + ps;
        m = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0[r] *= alpha;
            o1[r] *= alpha;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 pf = pack8(st, t);
            o0 = mfma32(frag_cols(sV, kb * 32, t, 0, lane), pf, o0);
            o1 = mfma32(frag_cols(sV, kb * 32, t, 1, lane), pf, o1);
        }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // O goes out through this wave's own (already consumed) rows of the Q image
    stage_tile(sQ, qb * 32, lane, o0, o1, inv);
    flush_tile(sQ, qb * 32, lane, out + sp.row0 * C + hd * 64, (size_t)C, L);
    if (query < L && lh == 0) lse[((size_t)b * H + hd) * Lmax + query] = (m + log2f(l)) * LN2;  // lse stays [B, H, Lmax]
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// WPE = waves per SIMD the register allocation is held to (2: up to 256 VGPRs; 3: 168 -- a handful of spills, but a third
// 3-wave workgroup fits on a CU at the text tower's L = 77, where LDS would allow three and 236 VGPRs allow two)
// TWO_PASS: dV and dK in two sub-passes over the query blocks (S is recomputed in the second: +4 of 16 MFMAs per block) so that only ONE
// pair of 32 x 64 accumulators is live at a time and each result leaves through the wave's own rows of the V image as soon as it is
// complete -- the live set fits 168 registers (three waves per SIMD, six workgroups of two waves per CU instead of four) without spills.
// DELTA_P (round 6; sequences of at most two key blocks: the image tower's 50 tokens, the short buckets of the packed text tower): the softmax
// backward's row term delta_q = sum_d dO[q,d] O[q,d] is taken as sum_j P[q,j] dP[q,j] -- the same number (O = P V, dP = dO V^T), formed from the P and dP
// tiles phase A has in registers anyway.  O is then not read at all: 2C of the kernel's 16C bytes per token, and the wave's per-lane global loads of its
// O rows, go away; P enters unrounded instead of through O's bf16 rounding.
template <int MAXT, int WPE, bool TWO_PASS = false, bool DELTA_P = false>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void attn_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                         const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                         bf16* __restrict__ dqkv, const int32_t* __restrict__ seq_off, int Lmax, int H, int causal,
                                                         float scale, int ablate, const int32_t* __restrict__ order, int order_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int LP = nwaves * 32;
    const int hd = blockIdx.x % H;
    const int b = seq_index(order, order_off, blockIdx.x / H);
    const int C = H * 64;
    const size_t rs = (size_t)3 * C;
    const SeqSpan sp = seq_span(seq_off, b, Lmax);
    const int L = sp.len, nb = (L + 31) >> 5, LPe = nb * 32;
    const bool active = wave < nb;  // waves behind the end of a short packed sequence only take part in the barriers
    const bf16* qbase = qkv + sp.row0 * rs + hd * 64;
    const bf16* dobase = dout + sp.row0 * C + hd * 64;
    const bf16* obase = out + sp.row0 * C + hd * 64;
    // Three images instead of four: the arithmetic of this kernel is latency-bound and its rate scales with the number of
    // resident workgroups (measured: profiles/r01_attn_bwd_occupancy.txt), and LDS is what caps that number.  Pass 1 holds K,
    // V and dO (dQ needs all keys; the wave's own Q and O rows come straight from global memory into fragments); before pass 2
    // (dK / dV need all queries) Q is staged over K while dQ leaves through the V image -- the wave's own K / V fragments are
    // taken from the images first -- so the only load a workgroup waits for in mid-flight is one image, issued together with
    // the dQ stores.
    char* sA = smem;                  // K, then Q, then the transposing buffer of dK
    char* sB = smem + LP * 128;       // V, then the transposing buffer of dQ
    char* sC = smem + 2 * LP * 128;   // dO, then the transposing buffer of dV
    float* sLse = (float*)(smem + 3 * LP * 128);
    float* sDelta = sLse + LP;
    const int lr = lane & 31, lh = lane >> 5;
    const float sc = scale * LOG2E;
    if (!(ablate & 1)) {
        stage_head(qbase + C, rs, L, LPe, sA, wave, nwaves, lane);
        stage_head(qbase + 2 * C, rs, L, LPe, sB, wave, nwaves, lane);
        stage_head(dobase, (size_t)C, L, LPe, sC, wave, nwaves, lane);
    }

    // ---- phase A: dQ for query block `wave` (lane <-> query, registers <-> keys) ----
    f32x16 dq0 = zero16(), dq1 = zero16();
    {
        const int qb = wave;
        const int query = qb * 32 + lr;
        const int qrow = query < L ? query : L - 1;  // rows >= L: copies of row L-1 (masked below)
        bf16x8 qf[4], dof[4], of[4];
        float lse_q = 0.f, delta_q = 0.f;
        if (!(ablate & 1)) {
            // this lane's 8 d's of k-step s are columns (2s + lh)*8 .. +7 of its query's row
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                qf[s] = *(const bf16x8*)(qbase + (size_t)qrow * rs + (s * 2 + lh) * 8);
                if constexpr (!DELTA_P) of[s] = *(const bf16x8*)(obase + (size_t)qrow * C + (s * 2 + lh) * 8);
            }
            lse_q = lse[((size_t)b * H + hd) * Lmax + qrow] * LOG2E;  // exp2 units
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                qf[s] = frag_rows(sA, active ? query : lr, s, lane);
                if constexpr (!DELTA_P) of[s] = qf[s];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // delta[q] = sum_d dO[q,d] * O[q,d] (each half-wave holds 32 of the 64 d's)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            dof[s] = frag_rows(sC, active ? query : lr, s, lane);
            if constexpr (!DELTA_P) {
#pragma unroll
                for (int e = 0; e < 8; ++e) delta_q += bf2f(dof[s][e]) * bf2f(of[s][e]);
            }
        }
        const int nkb = ((ablate & 2) || !active) ? 0 : (causal ? qb + 1 : nb);
        if constexpr (DELTA_P) {
            // every key block's P and dP first (at most two: 64 registers), delta from them, then dS and the dQ products
            f32x16 pk[2], dpk[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                pk[kb] = zero16();
                dpk[kb] = zero16();
                if (kb < nkb) {
                    f32x16 st = zero16(), dp = zero16();
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        st = mfma32(frag_rows(sA, kb * 32 + lr, s, lane), qf[s], st);
                        dp = mfma32(frag_rows(sB, kb * 32 + lr, s, lane), dof[s], dp);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kb * 32 + mfma32_row(r, lane);
                        const bool ok = key < L && query < L && (!causal || key <= query);
                        const float e = fast_exp2(st[r] * sc - lse_q);
                        const float p = ok ? e : 0.f;
                        pk[kb][r] = p;
                        dpk[kb][r] = dp[r];
                        delta_q = fmaf(p, dp[r], delta_q);
                    }
                }
            }
            delta_q += __shfl_xor(delta_q, 32, 64);  // a query's keys sit in both half-waves
            if (lh == 0) {
                sLse[query] = lse_q;
                sDelta[query] = delta_q;
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb < nkb) {
                    f32x16 st;
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[r] = pk[kb][r] * (dpk[kb][r] - delta_q) * scale;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bf16x8 dsf = pack8(st, t);
                        dq0 = mfma32(frag_cols(sA, kb * 32, t, 0, lane), dsf, dq0);
                        dq1 = mfma32(frag_cols(sA, kb * 32, t, 1, lane), dsf, dq1);
                    }
                }
            }
        } else {
        delta_q += __shfl_xor(delta_q, 32, 64);
        if (lh == 0) {  // phase B (behind the next barrier) reads these for every query
            sLse[query] = lse_q;
            sDelta[query] = delta_q;
        }
        for (int kb = 0; kb < nkb; ++kb) {
            f32x16 st = zero16(), dp = zero16();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                st = mfma32(frag_rows(sA, kb * 32 + lr, s, lane), qf[s], st);
                dp = mfma32(frag_rows(sB, kb * 32 + lr, s, lane), dof[s], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + mfma32_row(r, lane);
                const bool ok = key < L && query < L && (!causal || key <= query);
                const float e = fast_exp2(st[r] * sc - lse_q);
                    const float p = ok ? e : 0.f;
                st[r] = p * (dp[r] - delta_q) * scale;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 dsf = pack8(st, t);
                dq0 = mfma32(frag_cols(sA, kb * 32, t, 0, lane), dsf, dq0);
                dq1 = mfma32(frag_cols(sA, kb * 32, t, 1, lane), dsf, dq1);
            }
        }
        }
    }

    // ---- phase B: dK, dV for key block `wave` (lane <-> key, registers <-> queries) ----
    {
        const int kb = wave;
        const int key = kb * 32 + lr;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = frag_rows(sA, active ? key : lr, s, lane);
            vf[s] = frag_rows(sB, active ? key : lr, s, lane);
        }
        // every wave has finished phase A and holds its K / V fragments: Q is staged over K while dQ leaves through the V
        // image (keeping dQ in registers across phase B would cost occupancy)
        __syncthreads();
        bf16* dbase = dqkv + sp.row0 * rs + hd * 64;
        if (!(ablate & 1)) stage_head(qbase, rs, L, LPe, sA, wave, nwaves, lane);
        stage_tile(sB, wave * 32, lane, dq0, dq1, 1.0f);
        if (!(ablate & 4)) flush_tile(sB, wave * 32, lane, dbase, rs, L);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int qb0 = ((ablate & 2) || !active) ? nb : (causal ? kb : 0);
        if constexpr (TWO_PASS) {
            // the wave's rows of the V image carried dQ out; their reads are complete (flush_tile), so dV and then dK leave the same way,
            // with no barrier: nobody else touches these rows, and the Q / dO images stay intact for the other waves
            {
                f32x16 dv0 = zero16(), dv1 = zero16();
                for (int qb = qb0; qb < nb; ++qb) {
                    f32x16 st = zero16();
#pragma unroll
                    for (int s = 0; s < 4; ++s) st = mfma32(frag_rows(sA, qb * 32 + lr, s, lane), kf[s], st);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int query = qb * 32 + mfma32_row(r, lane);
                        const bool ok = key < L && query < L && (!causal || key <= query);
                        const float e = fast_exp2(st[r] * sc - sLse[query]);
                        st[r] = ok ? e : 0.f;
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bf16x8 pf = pack8(st, t);
                        dv0 = mfma32(frag_cols(sC, qb * 32, t, 0, lane), pf, dv0);
                        dv1 = mfma32(frag_cols(sC, qb * 32, t, 1, lane), pf, dv1);
                    }
                }
                stage_tile(sB, wave * 32, lane, dv0, dv1, 1.0f);
                if (!(ablate & 4)) flush_tile(sB, wave * 32, lane, dbase + 2 * C, rs, L);
            }
            {
                f32x16 dk0 = zero16(), dk1 = zero16();
                for (int qb = qb0; qb < nb; ++qb) {
                    f32x16 st = zero16(), dp = zero16();
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        st = mfma32(frag_rows(sA, qb * 32 + lr, s, lane), kf[s], st);
                        dp = mfma32(frag_rows(sC, qb * 32 + lr, s, lane), vf[s], dp);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int query = qb * 32 + mfma32_row(r, lane);
                        const bool ok = key < L && query < L && (!causal || key <= query);
                        const float ls = sLse[query], dl = sDelta[query];
                        const float e = fast_exp2(st[r] * sc - ls);
                        const float p = ok ? e : 0.f;
                        dp[r] = p * (dp[r] - dl) * scale;
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bf16x8 dsf = pack8(dp, t);
                        dk0 = mfma32(frag_cols(sA, qb * 32, t, 0, lane), dsf, dk0);
                        dk1 = mfma32(frag_cols(sA, qb * 32, t, 1, lane), dsf, dk1);
                    }
                }
                stage_tile(sB, wave * 32, lane, dk0, dk1, 1.0f);
                if (!(ablate & 4)) flush_tile(sB, wave * 32, lane, dbase + C, rs, L);
            }
        } else {
        f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
        for (int qb = qb0; qb < nb; ++qb) {
            f32x16 st = zero16(), dp = zero16();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                st = mfma32(frag_rows(sA, qb * 32 + lr, s, lane), kf[s], st);
                dp = mfma32(frag_rows(sC, qb * 32 + lr, s, lane), vf[s], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int query = qb * 32 + mfma32_row(r, lane);
                const bool ok = key < L && query < L && (!causal || key <= query);
                const float ls = sLse[query], dl = sDelta[query];  // read unconditionally: under `ok ? ... : 0` hipcc branches around every element
                const float e = fast_exp2(st[r] * sc - ls);
                const float p = ok ? e : 0.f;
                dp[r] = p * (dp[r] - dl) * scale;
                st[r] = p;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 pf = pack8(st, t), dsf = pack8(dp, t);
                dv0 = mfma32(frag_cols(sC, qb * 32, t, 0, lane), pf, dv0);
                dv1 = mfma32(frag_cols(sC, qb * 32, t, 1, lane), pf, dv1);
                dk0 = mfma32(frag_cols(sA, qb * 32, t, 0, lane), dsf, dk0);
                dk1 = mfma32(frag_cols(sA, qb * 32, t, 1, lane), dsf, dk1);
            }
        }
        // every wave is done with the Q / dO images: they become the transposing buffers of dK / dV
        __syncthreads();
        stage_tile(sA, wave * 32, lane, dk0, dk1, 1.0f);
        stage_tile(sC, wave * 32, lane, dv0, dv1, 1.0f);
        if (!(ablate & 4)) {
            flush_tile(sA, wave * 32, lane, dbase + C, rs, L);
            flush_tile(sC, wave * 32, lane, dbase + 2 * C, rs, L);
        }
        }
    }
}

// Causal variant (text tower).  With the causal mask the work of a wave is triangular: query block w sees w + 1 key blocks
// (phase A), key block w is seen by nwaves - w query blocks (phase B) -- balanced in total, but a barrier between the phases
// makes every wave wait for the longest of each.  Here Q, K, V and dO all stay resident and the outputs leave through a fifth
// buffer (only the wave's own rows of it), so there is no barrier after the one that publishes the images and a wave runs its
// phase A and phase B back to back.  5 images are affordable exactly where this matters: at L = 77 the register budget, not LDS,
// caps the text tower at two workgroups per CU.
template <int MAXT, int WPE>
__global__ __launch_bounds__(MAXT) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void attn_bwd_causal_kernel(
    const bf16* __restrict__ qkv, const bf16* __restrict__ out, const bf16* __restrict__ dout, const float* __restrict__ lse,
    bf16* __restrict__ dqkv, const int32_t* __restrict__ seq_off, int Lmax, int H, float scale, const int32_t* __restrict__ order,
    int order_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int LP = nwaves * 32;
    const int hd = blockIdx.x % H;
    const int b = seq_index(order, order_off, blockIdx.x / H);
    const int C = H * 64;
    const size_t rs = (size_t)3 * C;
    const SeqSpan sp = seq_span(seq_off, b, Lmax);
    const int L = sp.len, nb = (L + 31) >> 5, LPe = nb * 32;
    const bf16* qbase = qkv + sp.row0 * rs + hd * 64;
    const bf16* dobase = dout + sp.row0 * C + hd * 64;
    const bf16* obase = out + sp.row0 * C + hd * 64;
    char* sQ = smem;
    char* sK = smem + LP * 128;
    char* sV = smem + 2 * LP * 128;
    char* sdO = smem + 3 * LP * 128;
    char* sT = smem + 4 * LP * 128;  // transposing buffer of dQ, dK, dV (a wave only touches its own 32 rows)
    float* sLse = (float*)(smem + 5 * LP * 128);
    float* sDelta = sLse + LP;
    const int lr = lane & 31, lh = lane >> 5;
    const float sc = scale * LOG2E;
    stage_head(qbase, rs, L, LPe, sQ, wave, nwaves, lane);
    stage_head(qbase + C, rs, L, LPe, sK, wave, nwaves, lane);
    stage_head(qbase + 2 * C, rs, L, LPe, sV, wave, nwaves, lane);
    stage_head(dobase, (size_t)C, L, LPe, sdO, wave, nwaves, lane);
    const int qb = wave, query = qb * 32 + lr;
    const int qrow = query < L ? query : L - 1;
    bf16x8 of[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) of[s] = *(const bf16x8*)(obase + (size_t)qrow * C + (s * 2 + lh) * 8);
    const float lse_q = lse[((size_t)b * H + hd) * Lmax + qrow] * LOG2E;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16* dbase = dqkv + sp.row0 * rs + hd * 64;
    const bool active = wave < nb;  // waves behind the end of a short packed sequence only take part in the barriers

    // every wave first publishes lse / delta of its query block (phase B of OTHER waves reads them) ...
    bf16x8 qf[4], dof[4];
    float delta_q = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = frag_rows(sQ, active ? query : lr, s, lane);
        dof[s] = frag_rows(sdO, active ? query : lr, s, lane);
#pragma unroll
        for (int e = 0; e < 8; ++e) delta_q += bf2f(dof[s][e]) * bf2f(of[s][e]);
    }
    delta_q += __shfl_xor(delta_q, 32, 64);
    if (lh == 0) {
        sLse[query] = lse_q;
        sDelta[query] = delta_q;
    }
    __syncthreads();
    if (!active) return;  // no barrier follows

    // ---- phase A: dQ for query block `wave` (keys 0 .. query block) ----
    {
        f32x16 dq0 = zero16(), dq1 = zero16();
        for (int kb = 0; kb <= qb; ++kb) {
            f32x16 st = zero16(), dp = zero16();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                st = mfma32(frag_rows(sK, kb * 32 + lr, s, lane), qf[s], st);
                dp = mfma32(frag_rows(sV, kb * 32 + lr, s, lane), dof[s], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + mfma32_row(r, lane);
                const bool ok = key < L && query < L && key <= query;
                const float e = fast_exp2(st[r] * sc - lse_q);
                    const float p = ok ? e : 0.f;
                st[r] = p * (dp[r] - delta_q) * scale;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 dsf = pack8(st, t);
                dq0 = mfma32(frag_cols(sK, kb * 32, t, 0, lane), dsf, dq0);
                dq1 = mfma32(frag_cols(sK, kb * 32, t, 1, lane), dsf, dq1);
            }
        }
        stage_tile(sT, wave * 32, lane, dq0, dq1, 1.0f);
        flush_tile(sT, wave * 32, lane, dbase, rs, L);
    }

    // ---- phase B: dK, dV for key block `wave` (queries from the key block on) ----
    {
        const int kb = wave;
        const int key = kb * 32 + lr;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = frag_rows(sK, key, s, lane);
            vf[s] = frag_rows(sV, key, s, lane);
        }
        f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
        for (int q2 = kb; q2 < nb; ++q2) {
            f32x16 st = zero16(), dp = zero16();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                st = mfma32(frag_rows(sQ, q2 * 32 + lr, s, lane), kf[s], st);
                dp = mfma32(frag_rows(sdO, q2 * 32 + lr, s, lane), vf[s], dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = q2 * 32 + mfma32_row(r, lane);
                const bool ok = key < L && qq < L && key <= qq;
                const float ls = sLse[qq], dl = sDelta[qq];
                const float e = fast_exp2(st[r] * sc - ls);
                const float p = ok ? e : 0.f;
                dp[r] = p * (dp[r] - dl) * scale;
                st[r] = p;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 pf = pack8(st, t), dsf = pack8(dp, t);
                dv0 = mfma32(frag_cols(sdO, q2 * 32, t, 0, lane), pf, dv0);
                dv1 = mfma32(frag_cols(sdO, q2 * 32, t, 1, lane), pf, dv1);
                dk0 = mfma32(frag_cols(sQ, q2 * 32, t, 0, lane), dsf, dk0);
                dk1 = mfma32(frag_cols(sQ, q2 * 32, t, 1, lane), dsf, dk1);
            }
        }
        // the wave's rows of the transposing buffer: dQ's reads are complete (flush_tile consumed them), dK then dV follow
        stage_tile(sT, wave * 32, lane, dk0, dk1, 1.0f);
        flush_tile(sT, wave * 32, lane, dbase + C, rs, L);
        stage_tile(sT, wave * 32, lane, dv0, dv1, 1.0f);
        flush_tile(sT, wave * 32, lane, dbase + 2 * C, rs, L);
    }
}

#ifndef OCN_ATTN_BWD_TWO_PASS_DEFAULT
#define OCN_ATTN_BWD_TWO_PASS_DEFAULT true  // up to 128 tokens (profiles/r03_attention_backward_two_pass.txt: -4 % at 50 tokens; slower at 257)
#endif

int check_attn(const char* name, int B, int L, int H) {
    OCN_CHECK_ARG(B > 0 && H > 0 && L > 0, "%s: bad shape B=%d L=%d H=%d", name, B, L, H);
    OCN_CHECK_ARG(L <= 320, "%s: L=%d > 320 unsupported", name, L);
    return OCN_OK;
}

}  // namespace

namespace {
// One launch of `nseq` sequences whose workgroups have `nw` waves (32 rows each).  Dense batches and un-bucketed packed batches: one
// launch with nw = ceil(L / 32); bucketed packed batches: one launch per non-empty bucket (see seq_index).
int attn_fwd_launch(const void* qkv, void* out, float* lse, const int32_t* seq_off, const int32_t* order, int order_off, int nseq, int nw,
                    int L, int H, int causal, float scale, hipStream_t st) {
    const int lds = 3 * nw * 32 * 128;
    const bool ntl = g_ocn_tuning[9] == 2;
    if (nw <= 4) {
        if (ntl) hipLaunchKernelGGL((attn_fwd_kernel<256, true>), dim3(nseq * H), dim3(nw * 64), lds, st, (const bf16*)qkv, (bf16*)out, lse, seq_off, L, H, causal, scale, order, order_off);
        else hipLaunchKernelGGL((attn_fwd_kernel<256, false>), dim3(nseq * H), dim3(nw * 64), lds, st, (const bf16*)qkv, (bf16*)out, lse, seq_off, L, H, causal, scale, order, order_off);
    } else {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<640, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL((attn_fwd_kernel<640, false>), dim3(nseq * H), dim3(nw * 64), lds, st, (const bf16*)qkv, (bf16*)out, lse, seq_off, L, H, causal, scale, order, order_off);
    }
    OCN_CHECK_LAUNCH("ocn_attn_fwd");
    return OCN_OK;
}

int check_buckets(const char* fn, const int32_t* bucket_counts, int B, int L) {
    long n = 0;
    for (int k = 0; k < ocn_cdiv(L, 32); ++k) {
        OCN_CHECK_ARG(bucket_counts[k] >= 0, "%s: negative bucket count", fn);
        n += bucket_counts[k];
    }
    OCN_CHECK_ARG(n == B, "%s: bucket counts sum to %ld, not to B = %d", fn, n, B);
    return OCN_OK;
}

int attn_fwd_impl(const void* qkv, void* out, float* lse, const int32_t* seq_off, const int32_t* order, const int32_t* bucket_counts, int B,
                  int L, int H, int causal, float scale, ocn_stream_t stream) {
    OCN_CHECK_ARG(qkv && out && lse, "ocn_attn_fwd: null operand");
    if (int e = check_attn("ocn_attn_fwd", B, L, H)) return e;
    const int nw = ocn_cdiv(L, 32);
    if (!order) return attn_fwd_launch(qkv, out, lse, seq_off, nullptr, 0, B, nw, L, H, causal, scale, (hipStream_t)stream);
    if (int e = check_buckets("ocn_attn_fwd_varlen", bucket_counts, B, L)) return e;
    for (int k = 1, off = 0; k <= nw; off += bucket_counts[k - 1], ++k)
        if (bucket_counts[k - 1] > 0)
            if (int e = attn_fwd_launch(qkv, out, lse, seq_off, order, off, bucket_counts[k - 1], k, L, H, causal, scale, (hipStream_t)stream)) return e;
    return OCN_OK;
}

int attn_bwd_launch(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, const int32_t* seq_off,
                    const int32_t* order, int order_off, int nseq, int nw, int L, int H, int causal, float scale, hipStream_t st) {
    int lds = 3 * nw * 32 * 128 + 2 * nw * 32 * 4;
    OCN_CHECK_ARG(lds <= 160 * 1024, "ocn_attn_bwd: L=%d needs %d bytes of LDS (> 160 KiB)", L, lds);
    if (g_ocn_tuning[5] > 0 && lds + g_ocn_tuning[5] * 1024 <= 160 * 1024) lds += g_ocn_tuning[5] * 1024;  // developer knob: lower the occupancy
    const int lds5 = 5 * nw * 32 * 128 + 2 * nw * 32 * 4;
    if (causal && nw >= 2 && nw <= 3 && g_ocn_tuning[6] != 1 && g_ocn_tuning[1] == 0) {  // developer knob 6 = 1: use the generic kernel
        static bool cattr_set = false;
        if (!cattr_set) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_causal_kernel<256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            cattr_set = true;
        }
        hipLaunchKernelGGL((attn_bwd_causal_kernel<256, 2>), dim3(nseq * H), dim3(nw * 64), lds5, st, (const bf16*)qkv,
                           (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, seq_off, L, H, scale, order, order_off);
        OCN_CHECK_LAUNCH("ocn_attn_bwd");
        return OCN_OK;
    }
    // two-pass dK / dV build (168 registers, three waves per SIMD, no spills): developer knob 2 = 4 selects it, 2 = 5 forces the one-pass
    // build; default: see OCN_ATTN_BWD_TWO_PASS_DEFAULT
    const bool two_pass = g_ocn_tuning[2] == 4 || (g_ocn_tuning[2] != 5 && OCN_ATTN_BWD_TWO_PASS_DEFAULT && nw <= 4);
    // The CALL's longest sequence has at most two key blocks (the image tower's 50 tokens): delta from P and dP, O is not read.  Decided per call, not per
    // bucket: results must not depend on the bucketing, and the packed text tower (Lmax = 77) must stay bit-identical to the dense one, whose kernels
    // read O.  (developer knob 2 = 6: the O-reading form, A/B)
    if (ocn_cdiv(L, 32) <= 2 && two_pass && g_ocn_tuning[2] != 6) {
        hipLaunchKernelGGL((attn_bwd_kernel<256, 3, true, true>), dim3(nseq * H), dim3(nw * 64), lds, st, (const bf16*)qkv,
                           (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, seq_off, L, H, causal, scale, g_ocn_tuning[1], order, order_off);
    } else if (nw <= 4 && two_pass) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<256, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL((attn_bwd_kernel<256, 3, true>), dim3(nseq * H), dim3(nw * 64), lds, st, (const bf16*)qkv,
                           (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, seq_off, L, H, causal, scale, g_ocn_tuning[1], order, order_off);
    } else if (nw <= 4) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL((attn_bwd_kernel<256, 2>), dim3(nseq * H), dim3(nw * 64), lds, st, (const bf16*)qkv,
                           (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, seq_off, L, H, causal, scale, g_ocn_tuning[1], order, order_off);
    } else {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<640, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<640, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        // up to 10 waves per workgroup: three per SIMD, 168 registers -- the one-pass build spills 17 of them there, the two-pass one none
        if (two_pass) hipLaunchKernelGGL((attn_bwd_kernel<640, 1, true>), dim3(nseq * H), dim3(nw * 64), lds, st, (const bf16*)qkv,
                           (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, seq_off, L, H, causal, scale, g_ocn_tuning[1], order, order_off);
        else hipLaunchKernelGGL((attn_bwd_kernel<640, 1>), dim3(nseq * H), dim3(nw * 64), lds, st, (const bf16*)qkv,
                           (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, seq_off, L, H, causal, scale, g_ocn_tuning[1], order, order_off);
    }
    OCN_CHECK_LAUNCH("ocn_attn_bwd");
    return OCN_OK;
}

int attn_bwd_impl(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, const int32_t* seq_off,
                  const int32_t* order, const int32_t* bucket_counts, int B, int L, int H, int causal, float scale, ocn_stream_t stream) {
    OCN_CHECK_ARG(qkv && out && dout && lse && dqkv, "ocn_attn_bwd: null operand");
    if (int e = check_attn("ocn_attn_bwd", B, L, H)) return e;
    const int nw = ocn_cdiv(L, 32);
    if (!order) return attn_bwd_launch(qkv, out, dout, lse, dqkv, seq_off, nullptr, 0, B, nw, L, H, causal, scale, (hipStream_t)stream);
    if (int e = check_buckets("ocn_attn_bwd_varlen", bucket_counts, B, L)) return e;
    for (int k = 1, off = 0; k <= nw; off += bucket_counts[k - 1], ++k)
        if (bucket_counts[k - 1] > 0)
            if (int e = attn_bwd_launch(qkv, out, dout, lse, dqkv, seq_off, order, off, bucket_counts[k - 1], k, L, H, causal, scale, (hipStream_t)stream)) return e;
    return OCN_OK;
}
}  // namespace

extern "C" int ocn_attn_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int causal, float scale, ocn_stream_t stream) {
    return attn_fwd_impl(qkv, out, lse, nullptr, nullptr, nullptr, B, L, H, causal, scale, stream);
}

extern "C" int ocn_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int L, int H,
                            int causal, float scale, ocn_stream_t stream) {
    return attn_bwd_impl(qkv, out, dout, lse, dqkv, nullptr, nullptr, nullptr, B, L, H, causal, scale, stream);
}

// Packed ("varlen") batches, head_dim 64: sequence b owns rows seq_off[b] .. seq_off[b+1] of qkv / out / dout / dqkv
// (1 <= length <= Lmax, seq_off = B + 1 ascending int32 on the device); lse keeps the dense [B, H, Lmax] layout.
// Optional bucketing: `order` (device, B sequence ids grouped by ceil(len / 32) ascending, from ocn_seq_pack_plan) + `bucket_counts`
// (HOST array of ceil(Lmax / 32) ints summing to B): one launch per non-empty bucket with workgroups sized for that bucket.
extern "C" int ocn_attn_fwd_varlen(const void* qkv, void* out, float* lse, const int32_t* seq_off, const int32_t* order,
                                   const int32_t* bucket_counts, int B, int Lmax, int H, int causal, float scale, ocn_stream_t stream) {
    OCN_CHECK_ARG(seq_off, "ocn_attn_fwd_varlen: null seq_off");
    OCN_CHECK_ARG((order == nullptr) == (bucket_counts == nullptr), "ocn_attn_fwd_varlen: order and bucket_counts go together");
    return attn_fwd_impl(qkv, out, lse, seq_off, order, bucket_counts, B, Lmax, H, causal, scale, stream);
}

extern "C" int ocn_attn_bwd_varlen(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, const int32_t* seq_off,
                                   const int32_t* order, const int32_t* bucket_counts, int B, int Lmax, int H, int causal, float scale,
                                   ocn_stream_t stream) {
    OCN_CHECK_ARG(seq_off, "ocn_attn_bwd_varlen: null seq_off");
    OCN_CHECK_ARG((order == nullptr) == (bucket_counts == nullptr), "ocn_attn_bwd_varlen: order and bucket_counts go together");
    return attn_bwd_impl(qkv, out, dout, lse, dqkv, seq_off, order, bucket_counts, B, Lmax, H, causal, scale, stream);
}

// ---- explicit head_dim: dispatch between the specialised (head_dim 64, head resident) and the generic kernels ---------------
int ocn_launch_attn_generic_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int D, int causal, float scale, hipStream_t st);
int ocn_launch_attn_generic_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta, int B,
                                int L, int H, int D, int causal, float scale, hipStream_t st);

// head_dim 64: the head-resident kernels above (one workgroup per head) up to OCN_ATTN_RESIDENT_MAX_L tokens, the streamed kernels of
// attention_generic.hip beyond (at 257 tokens a resident head is 110 KB of LDS: one workgroup per CU, loads serialised in front of the
// arithmetic); developer knob 7: 1 = always streamed, 2 = always resident (L <= 320)
// measured at ViT-L-14's 257 tokens (profiles/r03_attention_long_sequences.txt): forward 2.13 ms streamed vs 2.49 resident, backward 6.3 vs 5.7
#ifndef OCN_ATTN_RESIDENT_MAX_L_FWD
#define OCN_ATTN_RESIDENT_MAX_L_FWD 128
#endif
#ifndef OCN_ATTN_RESIDENT_MAX_L_BWD
#define OCN_ATTN_RESIDENT_MAX_L_BWD 320
#endif
static bool attn_use_resident(int head_dim, int L, int max_l) {
    if (head_dim != 64 || L > 320 || g_ocn_tuning[7] == 1) return false;
    return g_ocn_tuning[7] == 2 || L <= max_l;
}

extern "C" int ocn_attn_fwd_hd(const void* qkv, void* out, float* lse, int B, int L, int H, int head_dim, int causal, float scale,
                               ocn_stream_t stream) {
    if (attn_use_resident(head_dim, L, OCN_ATTN_RESIDENT_MAX_L_FWD)) return ocn_attn_fwd(qkv, out, lse, B, L, H, causal, scale, stream);
    OCN_CHECK_ARG(qkv && out && lse && B > 0 && L > 0 && H > 0, "ocn_attn_fwd_hd: bad arguments");
    const int rc = ocn_launch_attn_generic_fwd(qkv, out, lse, B, L, H, head_dim, causal, scale, (hipStream_t)stream);
    if (rc == 1) {
        ocn_set_error("ocn_attn_fwd_hd: head_dim=%d, L=%d unsupported (head_dim in {64,80,88,96,104,112,128}; K and V of a head must fit 160 KiB of LDS)", head_dim, L);
        return OCN_ERR_UNSUPPORTED;
    }
    OCN_CHECK_LAUNCH("ocn_attn_fwd_hd");
    return OCN_OK;
}

extern "C" int ocn_attn_bwd_hd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta_ws, int B,
                               int L, int H, int head_dim, int causal, float scale, ocn_stream_t stream) {
    if (attn_use_resident(head_dim, L, OCN_ATTN_RESIDENT_MAX_L_BWD)) return ocn_attn_bwd(qkv, out, dout, lse, dqkv, B, L, H, causal, scale, stream);
    OCN_CHECK_ARG(qkv && out && dout && lse && dqkv && delta_ws && B > 0 && L > 0 && H > 0, "ocn_attn_bwd_hd: bad arguments");
    const int rc = ocn_launch_attn_generic_bwd(qkv, out, dout, lse, dqkv, delta_ws, B, L, H, head_dim, causal, scale, (hipStream_t)stream);
    if (rc == 1) {
        ocn_set_error("ocn_attn_bwd_hd: head_dim=%d, L=%d unsupported (head_dim in {64,80,88,96,104,112,128}; Q and dO of a head must fit 160 KiB of LDS)", head_dim, L);
        return OCN_ERR_UNSUPPORTED;
    }
    OCN_CHECK_LAUNCH("ocn_attn_bwd_hd");
    return OCN_OK;
}
