// Single-query attention for the LAST residual block of a tower (head_dim 64).
//
// Both poolers read ONE row per sequence of the last block's output (`x[:, 0]` behind ln_post, transformer.py:829-831; `x[arange,
// text.argmax(-1)]` behind ln_final, :941-944), and the only place where rows of a residual block mix is the attention: the pooled row's
// QUERY reads the keys / values of its sequence.  So of the last block's attention only one query per (sequence, head) is ever needed:
//
//   forward   s_j = scale <q, k_j>,  p = softmax_j(s),  out = sum_j p_j v_j                                      j over the keys the query sees
//   backward  dP_j = <dout, v_j>,  delta = <dout, out>,  dS_j = p_j (dP_j - delta)
//             dq = scale sum_j dS_j k_j        dk_j = scale dS_j q        dv_j = p_j dout                        (rank-1 in every key row)
//
// which is 2 L d flops per head instead of 4 L^2 d, reads K and V once and never touches Q of the other rows (the caller does not even
// project it: model.py::_pooled_block_forward).  The kernels are HBM-bound streams over the [M, 2C] K | V matrix: one wave per (sequence,
// head); a lane owns 16 bytes (8 dims) of one key row, 8 lanes cover a 128-byte head row, 8 key rows per wave-load; softmax statistics are
// kept per lane group (keys j = g mod 8) online and merged across the 8 groups at the end, all arithmetic in fp32.
//
// Keys of sequence b: rows [begin, begin + len) of kv, begin = seq_off[b] (packed text rows) or b*L; the query is the pooled row rows[b]
// and, under the causal mask (transformer.py:1716-1722), sees the keys up to and including its own row.  The backward writes EVERY key row
// of the sequence (zeros behind the pooled row), so dkv needs no clearing.
#include "ocn_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

OCN_DEV void unpack8(u32x4_t v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, v[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xffff0000u);
    }
}
OCN_DEV u32x4_t pack8(const float (&f)[8]) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bf16x2 p = {f2bf(f[2 * i]), f2bf(f[2 * i + 1])};
        v[i] = __builtin_bit_cast(unsigned, p);
    }
    return v;
}
OCN_DEV float group8_sum(float v) {  // over the 8 lanes that share a key row (lane & 7)
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}
OCN_DEV float across_groups_sum(float v) {  // over the 8 lane groups (lane >> 3)
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
OCN_DEV float across_groups_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 8, 64));
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

struct PooledArgs {
    const bf16* q;      // [B, C]
    const bf16* kv;     // [M, 2C]: K | V column blocks, heads contiguous inside each
    const bf16* out;    // [B, C]   (forward: written)
    const bf16* dout;   // [B, C]
    float* lse;         // [B*H] natural log
    bf16* dq;           // [B, C]
    bf16* dkv;          // [M, 2C]
    const int* seq_off; // [B+1] or null
    const int* rows;    // [B] absolute row of the pooled token
    int B, L, H, causal;
    float scale;
};

OCN_DEV void key_range(const PooledArgs& a, int b, int& begin, int& seq_end, int& vis_end) {
    begin = a.seq_off ? a.seq_off[b] : b * a.L;
    seq_end = a.seq_off ? a.seq_off[b + 1] : begin + a.L;
    vis_end = a.causal ? a.rows[b] + 1 : seq_end;
    if (vis_end > seq_end) vis_end = seq_end;
    if (vis_end < begin) vis_end = begin;  // a pooled row in front of its sequence: an EMPTY visible range (output 0, no NaN), never a negative one
}

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

__global__ __launch_bounds__(256) void attn_pooled_fwd_kernel(PooledArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= a.B * a.H) return;
    const int b = bh / a.H, h = bh - b * a.H;
    const int g = lane >> 3, c = lane & 7;
    const int C = a.H * 64;
    int begin, seq_end, vis_end;
    key_range(a, b, begin, seq_end, vis_end);
    float q[8];
    unpack8(*(const u32x4_t*)(a.q + (size_t)b * C + h * 64 + c * 8), q);
    const float sc = a.scale * LOG2E;  // exp2 domain
    const bf16* kbase = a.kv + h * 64 + c * 8;
    float m = -1e30f, l = 0.f, acc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = 0.f;
    // two key rows per lane in flight
    for (int j0 = begin; j0 < vis_end; j0 += 16) {
        u32x4_t kr[2], vr[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = j0 + u * 8 + g;
            ok[u] = j < vis_end;
            const bf16* p = kbase + (size_t)(ok[u] ? j : begin) * (2 * C);
            kr[u] = *(const u32x4_t*)p;
            vr[u] = *(const u32x4_t*)(p + C);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float k[8], v[8];
            unpack8(kr[u], k);
            unpack8(vr[u], v);
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) s = fmaf(q[d], k[d], s);
            s = group8_sum(s) * sc;
            if (!ok[u]) s = -INFINITY;
            const float mn = fmaxf(m, s);
            const float corr = fast_exp2(m - mn), p = fast_exp2(s - mn);
            l = fmaf(l, corr, p);
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[d] = fmaf(acc[d], corr, p * v[d]);
            m = mn;
        }
    }
    // merge the 8 lane groups' partial softmaxes
    const float mt = across_groups_max(m);
    const float w = fast_exp2(m - mt);
    const float lt = across_groups_sum(l * w);
    // an empty visible key range (zero-length sequence, pooled row outside its sequence): lt = 0 -- the output is 0 and the statistic a finite
    // very negative number instead of 0 * inf = NaN that the backward would spread into dkv, dq and the weight gradients
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    float o[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = across_groups_sum(acc[d] * w) * inv;
    if (g == 0) *(u32x4_t*)(const_cast<bf16*>(a.out) + (size_t)b * C + h * 64 + c * 8) = pack8(o);
    if (lane == 0) a.lse[bh] = lt > 0.f ? (mt + __log2f(lt)) * LN2 : -1e30f;
}

__global__ __launch_bounds__(256) void attn_pooled_bwd_kernel(PooledArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= a.B * a.H) return;
    const int b = bh / a.H, h = bh - b * a.H;
    const int g = lane >> 3, c = lane & 7;
    const int C = a.H * 64;
    int begin, seq_end, vis_end;
    key_range(a, b, begin, seq_end, vis_end);
    const size_t prow = (size_t)b * C + h * 64 + c * 8;
    float q[8], o[8], dO[8];
    unpack8(*(const u32x4_t*)(a.q + prow), q);
    unpack8(*(const u32x4_t*)(a.out + prow), o);
    unpack8(*(const u32x4_t*)(a.dout + prow), dO);
    float delta = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) delta = fmaf(dO[d], o[d], delta);
    delta = group8_sum(delta);
    const float sc = a.scale * LOG2E, lse2 = a.lse[bh] * LOG2E;
    const bf16* kbase = a.kv + h * 64 + c * 8;
    bf16* dbase = a.dkv + h * 64 + c * 8;
    float dq[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) dq[d] = 0.f;
    for (int j0 = begin; j0 < seq_end; j0 += 16) {
        u32x4_t kr[2], vr[2];
        int jj[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            jj[u] = j0 + u * 8 + g;
            const bf16* p = kbase + (size_t)(jj[u] < vis_end ? jj[u] : begin) * (2 * C);
            kr[u] = *(const u32x4_t*)p;
            vr[u] = *(const u32x4_t*)(p + C);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float k[8], v[8];
            unpack8(kr[u], k);
            unpack8(vr[u], v);
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                s = fmaf(q[d], k[d], s);
                dp = fmaf(dO[d], v[d], dp);
            }
            s = group8_sum(s);
            dp = group8_sum(dp);
            const bool vis = jj[u] < vis_end;
            const float p = vis ? fast_exp2(fmaf(s, sc, -lse2)) : 0.f;
            const float ds = p * (dp - delta) * a.scale;
            float dk[8], dv[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                dq[d] = fmaf(ds, k[d], dq[d]);
                dk[d] = ds * q[d];
                dv[d] = p * dO[d];
            }
            if (jj[u] < seq_end) {  // rows behind the pooled row get zeros (p = 0): every key row of the sequence is written
                bf16* dp_ = dbase + (size_t)jj[u] * (2 * C);
                *(u32x4_t*)dp_ = pack8(dk);
                *(u32x4_t*)(dp_ + C) = pack8(dv);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) dq[d] = across_groups_sum(dq[d]);
    if (g == 0) *(u32x4_t*)(a.dq + prow) = pack8(dq);
}

int check(const PooledArgs& a, const char* name) {
    OCN_CHECK_ARG(a.q && a.kv && a.out && a.lse && a.rows, "%s: null operand", name);
    OCN_CHECK_ARG(a.B > 0 && a.L > 0 && a.H > 0, "%s: bad shape B=%d L=%d H=%d", name, a.B, a.L, a.H);
    OCN_CHECK_ARG((((uintptr_t)a.q | (uintptr_t)a.kv | (uintptr_t)a.out) & 15) == 0, "%s: operands must be 16-byte aligned", name);
    OCN_CHECK_ARG(((uintptr_t)a.lse & 3) == 0, "%s: lse must be 4-byte aligned", name);
    return OCN_OK;
}

}  // namespace

extern "C" int ocn_attn_pooled_fwd(const void* q, const void* kv, void* out, float* lse, const int32_t* seq_off, const int32_t* rows, int B, int L,
                                   int H, int causal, float scale, ocn_stream_t stream) {
    PooledArgs a{(const bf16*)q, (const bf16*)kv, (const bf16*)out, nullptr, lse, nullptr, nullptr, seq_off, rows, B, L, H, causal, scale};
    if (int e = check(a, "ocn_attn_pooled_fwd")) return e;
    hipLaunchKernelGGL(attn_pooled_fwd_kernel, dim3((B * H + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    OCN_CHECK_LAUNCH("ocn_attn_pooled_fwd");
    return OCN_OK;
}

extern "C" int ocn_attn_pooled_bwd(const void* q, const void* kv, const void* out, const void* dout, const float* lse, void* dq, void* dkv,
                                   const int32_t* seq_off, const int32_t* rows, int B, int L, int H, int causal, float scale, ocn_stream_t stream) {
    PooledArgs a{(const bf16*)q, (const bf16*)kv, (const bf16*)out, (const bf16*)dout, const_cast<float*>(lse), (bf16*)dq, (bf16*)dkv, seq_off, rows, B, L, H, causal, scale};
    if (int e = check(a, "ocn_attn_pooled_bwd")) return e;
    OCN_CHECK_ARG(dout && dq && dkv && ((((uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dkv) & 15) == 0), "ocn_attn_pooled_bwd: null or misaligned gradient operand");
    hipLaunchKernelGGL(attn_pooled_bwd_kernel, dim3((B * H + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    OCN_CHECK_LAUNCH("ocn_attn_pooled_bwd");
    return OCN_OK;
}
