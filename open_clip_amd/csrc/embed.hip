// HBM-bound embedding / pooling / normalisation kernels around the two transformer stacks.
#include "ocn_common.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

// ---- patchify: image [B,3,H,W] -> patches bf16 [B*gh*gw, Kpad], column = c*P*P + i*P + j ------
template <typename T, int VEC>
__global__ void patchify_kernel(const T* __restrict__ img, bf16* __restrict__ out, int B, int H, int W, int P, int Kpad,
                                long total) {
    const int gh = H / P, gw = W / P, KP = 3 * P * P, kv = Kpad / VEC;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / kv;
        const int k = (int)(idx % kv) * VEC;
        bf16* o = out + row * Kpad + k;
        if (k >= KP) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = (bf16)0.f;
            continue;
        }
        const int c = k / (P * P), rem = k % (P * P), i = rem / P, j = rem % P;
        const int b = (int)(row / (gh * gw)), g = (int)(row % (gh * gw)), py = g / gw, px = g % gw;
        const T* s = img + (((size_t)b * 3 + c) * H + py * P + i) * W + px * P + j;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = f2bf((float)s[e]);
    }
}

// uint8 pixels (the decoded, resized image before ToTensor / Normalize, transform.py:367-510): the kernel applies
// x/255, - mean[c], / std[c] (constants.py:1-2) while it builds the bf16 patch matrix -- 4x fewer bytes over PCIe and HBM
// than the fp32 tensor prepare_batch moves (base_task.py:135-157).  hwc = 1: [B,H,W,3] (PIL / decoder order), 0: [B,3,H,W].
struct NormC {
    float scale[3], shift[3];  // y = x * scale[c] + shift[c]
};
__global__ void patchify_u8_kernel(const unsigned char* __restrict__ img, bf16* __restrict__ out, int B, int H, int W, int P, int Kpad,
                                   long total, int hwc, NormC nc) {
    const int gh = H / P, gw = W / P, KP = 3 * P * P, kv = Kpad / 2;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / kv;
        const int k = (int)(idx % kv) * 2;
        bf16* o = out + row * Kpad + k;
        if (k >= KP) {
            o[0] = (bf16)0.f;
            o[1] = (bf16)0.f;
            continue;
        }
        const int c = k / (P * P), rem = k % (P * P), i = rem / P, j = rem % P;
        const int b = (int)(row / (gh * gw)), g = (int)(row % (gh * gw)), py = g / gw, px = g % gw;
        const int y = py * P + i, x = px * P + j;
        float v0, v1;
        if (hwc) {
            const unsigned char* s = img + (((size_t)b * H + y) * W + x) * 3 + c;
            v0 = (float)s[0];
            v1 = (float)s[3];
        } else {
            const unsigned char* s = img + (((size_t)b * 3 + c) * H + y) * W + x;
            v0 = (float)s[0];
            v1 = (float)s[1];
        }
        o[0] = f2bf(fmaf(v0, nc.scale[c], nc.shift[c]));
        o[1] = f2bf(fmaf(v1, nc.scale[c], nc.shift[c]));
    }
}

// Fast path of the same for the decoder layout [B,H,W,3] with P % 16 == 0 (ViT-B-32 / B-16): a patch row is 3P contiguous bytes,
// so a workgroup pulls one whole patch (P rows x 3P bytes) into LDS with 16-byte loads and writes its 3P^2 bf16 values as 16-byte
// stores (8 consecutive k = one colour plane, one patch row, 8 consecutive pixels: bytes 3 apart in the staged row).  The generic
// kernel above moves 2 bytes in and 4 out per thread (4096 x 224 x 224: 4 ms instead of 0.5).
__global__ __launch_bounds__(256) void patchify_u8_hwc_kernel(const unsigned char* __restrict__ img, bf16* __restrict__ out, int H, int W, int P,
                                                              int Kpad, long npatch, NormC nc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pix[];
    const int gh = H / P, gw = W / P, rowb = 3 * P, n16 = P * (rowb >> 4), per_row = rowb >> 4, KP = 3 * P * P, nq = Kpad >> 3;
    for (long patch = blockIdx.x; patch < npatch; patch += gridDim.x) {
        const int b = (int)(patch / (gh * gw)), g = (int)(patch % (gh * gw)), py = g / gw, px = g % gw;
        const unsigned char* base = img + (((size_t)b * H + (size_t)py * P) * W + (size_t)px * P) * 3;
        __syncthreads();  // the previous patch has been read out of LDS
        for (int t = threadIdx.x; t < n16; t += blockDim.x) {
            const int i = t / per_row, c16 = t - i * per_row;
            *(uint4*)(pix + i * rowb + c16 * 16) = *(const uint4*)(base + (size_t)i * W * 3 + c16 * 16);
        }
        __syncthreads();
        bf16* o = out + patch * Kpad;
        for (int q = threadIdx.x; q < nq; q += blockDim.x) {
            const int k0 = q << 3;
            bf16x8 v;
            if (k0 >= KP) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
            } else {
                const int c = k0 / (P * P), rem = k0 - c * P * P, i = rem / P, j0 = rem - i * P;
                const unsigned char* sp = pix + i * rowb + j0 * 3 + c;
                const float sc = nc.scale[c], sh = nc.shift[c];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = f2bf(fmaf((float)sp[3 * e], sc, sh));
            }
            *(bf16x8*)(o + k0) = v;
        }
    }
}

// ---- class token + positional embedding (transformer.py:799-801) --------------------------------
__global__ void embed_assemble_fwd_kernel(const float* __restrict__ po, const float* __restrict__ cls,
                                          const float* __restrict__ pos, float* __restrict__ emb, int B, int G, int C) {
    const int T = G + 1, c4n = C / 4;
    const long total = (long)B * T * c4n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const long bt = idx / c4n;
        const int t = (int)(bt % T);
        const long b = bt / T;
        const f32x4 p = *(const f32x4*)(pos + (size_t)t * C + c);
        const f32x4 v = (t == 0) ? *(const f32x4*)(cls + c) : *(const f32x4*)(po + ((size_t)b * G + t - 1) * C + c);
        *(f32x4*)(emb + (size_t)bt * C + c) = v + p;
    }
}

// grid.x covers (t, c4); grid.y = batch chunks.  dpos/dcls by fp32 atomics (one per chunk per element).
__global__ void embed_assemble_bwd_kernel(const float* __restrict__ demb, bf16* __restrict__ dpatch,
                                          float* __restrict__ dpos, float* __restrict__ dcls, int B, int G, int C, int bchunk) {
    const int T = G + 1, c4n = C / 4;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * c4n) return;
    const int t = idx / c4n, c = (idx % c4n) * 4;
    const int b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int b = b0; b < b1; ++b) {
        const f32x4 v = *(const f32x4*)(demb + ((size_t)b * T + t) * C + c);
        acc = acc + v;
        if (t > 0) {
            bf16x4 o4 = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            *(bf16x4*)(dpatch + ((size_t)b * G + t - 1) * C + c) = o4;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsafeAtomicAdd(dpos + (size_t)t * C + c + e, acc[e]);
        if (t == 0) unsafeAtomicAdd(dcls + c + e, acc[e]);
    }
}

// ---- token embedding (model.py:399-401) ----------------------------------------------------------
__global__ void token_embed_fwd_kernel(const int64_t* __restrict__ text, const float* __restrict__ table,
                                       const float* __restrict__ pos, float* __restrict__ x, int B, int L, int C, int vocab) {
    const int c4n = C / 4;
    const long total = (long)B * L * c4n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const long bl = idx / c4n;
        const int l = (int)(bl % L);
        long tok = text[bl];
        tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
        *(f32x4*)(x + (size_t)bl * C + c) = *(const f32x4*)(table + (size_t)tok * C + c) + *(const f32x4*)(pos + (size_t)l * C + c);
    }
}

__global__ void token_embed_bwd_kernel(const int64_t* __restrict__ text, const float* __restrict__ dx,
                                       float* __restrict__ dtable, float* __restrict__ dpos, int B, int L, int C, int vocab,
                                       int bchunk) {
    // thread <-> (position l, ONE column), loops over a chunk of the batch: a wave's atomic instruction then covers 256
    // contiguous bytes of one table row (2 cache lines, fully used) instead of 4 bytes out of every 16 over 8 lines.
    // Rows that share a token at the same position (SOT at l = 0, the zero padding behind EOT: about half of all
    // tokens) are run-length combined in registers before the fp32 atomic, which removes the hot-address serialisation.
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L * C) return;
    const int l = idx / C, c = idx % C;
    const int b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
    float acc = 0.f, run = 0.f;
    long run_tok = -1;
    for (int b = b0; b < b1; ++b) {
        const float v = dx[((size_t)b * L + l) * C + c];
        acc += v;
        long tok = text[(size_t)b * L + l];
        tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
        if (tok != run_tok) {
            if (run_tok >= 0) unsafeAtomicAdd(dtable + (size_t)run_tok * C + c, run);
            run_tok = tok;
            run = v;
        } else {
            run += v;
        }
    }
    if (run_tok >= 0) unsafeAtomicAdd(dtable + (size_t)run_tok * C + c, run);
    unsafeAtomicAdd(dpos + (size_t)l * C + c, acc);
}

// The same gradient from SORTED token ids (segment reduce instead of one fp32 atomic per (token, column)): `keys` = the flat token ids
// in ascending order, `order[i]` = the flat (b*L + l) row of dx that keys[i] came from.  A workgroup walks CH consecutive entries with
// one float4 of columns per thread; a run of equal ids that starts and ends inside the chunk (and is bounded by different ids on both
// sides) is complete and is STORED (dtable arrives zeroed), only runs that cross a chunk boundary -- the zero padding, SOT / EOT --
// are added atomically, once per chunk instead of once per occurrence.  0.29 ms instead of 1.43 ms at B = 4096 (+ the sort).
template <typename T>
OCN_DEV f32x4 load4f(const T* p);
template <>
OCN_DEV f32x4 load4f<float>(const float* p) { return *(const f32x4*)p; }
template <>
OCN_DEV f32x4 load4f<bf16>(const bf16* p) {
    const bf16x4 v = *(const bf16x4*)p;
    return (f32x4){bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
}

template <typename T>
__global__ __launch_bounds__(128) void token_embed_bwd_sorted_kernel(const int64_t* __restrict__ keys, const int64_t* __restrict__ order,
                                                                     const T* __restrict__ dx, float* __restrict__ dtable, long n, int C,
                                                                     int vocab, int CH, int det) {
    const long i0 = (long)blockIdx.x * CH, i1 = min(n, i0 + CH);
    // a run is identified by the CLAMPED id (ids outside the vocabulary share row 0 / vocab - 1), so whether a run continues into the
    // neighbouring chunk is decided on clamped ids too: two chunks must never both take one row for a complete run and plain-store it
    auto clamped = [&](long i) -> long {
        const long t = keys[i];
        return t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
    };
    if (det) {
        // reproducible form (NativeCLIP(deterministic=True)): a run of equal ids is summed by ONE workgroup -- the one whose chunk holds the run's
        // first entry walks it to its end, wherever that is, in sorted (stable) order, and plain-stores the row; a chunk that begins inside a run
        // skips it.  No atomics; long runs (SOT / EOT: B entries; the zero padding of a dense batch) serialise on one workgroup.
        for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {
            long i = i0;
            if (i0 > 0) {
                const long t0 = clamped(i0);
                if (clamped(i0 - 1) == t0)
                    while (i < n && clamped(i) == t0) ++i;
            }
            while (i < i1) {
                const long tok = clamped(i);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (; i < n && clamped(i) == tok; ++i) acc += load4f<T>(dx + (size_t)order[i] * C + c);
                *(f32x4*)(dtable + (size_t)tok * C + c) = acc;
            }
        }
        return;
    }
    for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        long cur = -1, seg_start = i0;
        for (long i = i0; i <= i1; ++i) {
            const long tok = i < i1 ? clamped(i) : -2;
            if (tok != cur) {
                if (cur >= 0) {
                    const bool closed_l = seg_start > i0 || i0 == 0 || clamped(i0 - 1) != clamped(i0);
                    const bool closed_r = i < i1 || i1 == n || clamped(i1) != clamped(i1 - 1);
                    float* d = dtable + (size_t)cur * C + c;
                    if (closed_l && closed_r) {
                        *(f32x4*)d = acc;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(d + e, acc[e]);
                    }
                }
                cur = tok;
                seg_start = i;
                acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (i < i1) acc += load4f<T>(dx + (size_t)order[i] * C + c);
        }
    }
}

// dpos[l, :] += sum_b dx[b, l, :]   (one float4 of columns per thread, a chunk of the batch per workgroup row)
template <typename T>
__global__ void pos_grad_kernel(const T* __restrict__ dx, float* __restrict__ dpos, int B, int L, int C, int bchunk) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, c4n = C / 4;
    if (idx >= L * c4n) return;
    const int l = idx / c4n, c = (idx % c4n) * 4;
    const int b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int b = b0; b < b1; ++b) acc += load4f<T>(dx + ((size_t)b * L + l) * C + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dpos + (size_t)l * C + c + e, acc[e]);
}

// ---- packed ("varlen") text batches ----------------------------------------------------------------------------------------------
// The text tower pools x[b, argmax(text[b])] (transformer.py:941-944) under a causal mask (:1716-1722): tokens behind the pooled
// one cannot influence the feature, so only the first eot[b] + 1 tokens of each sequence are kept.  seq_off = exclusive scan of those
// lengths (B + 1 entries), last_row[b] = seq_off[b+1] - 1 = the pooled row.  One workgroup (B is a few thousand).
__global__ __launch_bounds__(1024) void seq_plan_kernel(const int32_t* __restrict__ eot, int32_t* __restrict__ seq_off,
                                                        int32_t* __restrict__ last_row, int B) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) {
        carry_s = 0;
        seq_off[0] = 0;
    }
    __syncthreads();
    for (int base = 0; base < B; base += 1024) {
        const int i = base + tid;
        int v = i < B ? eot[i] + 1 : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(v, o, 64);
            if (lane >= o) v += t;
        }
        if (lane == 63) wsum[w] = v;
        __syncthreads();
        int pre = carry_s;
        for (int k = 0; k < w; ++k) pre += wsum[k];
        v += pre;  // inclusive prefix over the whole batch
        if (i < B) {
            seq_off[i + 1] = v;
            last_row[i] = v - 1;
        }
        __syncthreads();
        if (tid == 1023) carry_s = v;
        __syncthreads();
    }
}

// tokens[r] / posidx[r] of packed row r = seq_off[b] + l  (l <= eot[b])
__global__ void seq_pack_rows_kernel(const int64_t* __restrict__ text, const int32_t* __restrict__ seq_off, int64_t* __restrict__ tokens,
                                     int32_t* __restrict__ posidx, int B, int L) {
    const long total = (long)B * L;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / L), l = (int)(i % L);
        const int o = seq_off[b];
        if (l < seq_off[b + 1] - o) {
            tokens[o + l] = text[i];
            posidx[o + l] = l;
        }
    }
}

__global__ void token_embed_fwd_rows_kernel(const int64_t* __restrict__ tokens, const int32_t* __restrict__ posidx,
                                            const float* __restrict__ table, const float* __restrict__ pos, float* __restrict__ x, long M,
                                            int C, int vocab) {
    const int c4n = C / 4;
    const long total = M * c4n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        const long r = idx / c4n;
        long tok = tokens[r];
        tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
        *(f32x4*)(x + (size_t)r * C + c) = *(const f32x4*)(table + (size_t)tok * C + c) + *(const f32x4*)(pos + (size_t)posidx[r] * C + c);
    }
}

// dpos[l, :] += sum over the sequences that HAVE a position l of dx[seq_off[b] + l, :]
template <typename T>
__global__ void pos_grad_varlen_kernel(const T* __restrict__ dx, float* __restrict__ dpos, const int32_t* __restrict__ seq_off, int B, int L,
                                       int C, int bchunk) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, c4n = C / 4;
    if (idx >= L * c4n) return;
    const int l = idx / c4n, c = (idx % c4n) * 4;
    const int b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bool any = false;
    for (int b = b0; b < b1; ++b) {
        const int o = seq_off[b];
        if (l < seq_off[b + 1] - o) {
            acc += load4f<T>(dx + ((size_t)o + l) * C + c);
            any = true;
        }
    }
    if (!any) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dpos + (size_t)l * C + c + e, acc[e]);
}

// ---- pooling ---------------------------------------------------------------------------------------
__global__ void argmax_rows_kernel(const int64_t* __restrict__ text, int32_t* __restrict__ idx, int B, int L) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= B) return;
    long best = INT64_MIN;
    int bi = 0x7fffffff;
    for (int l = lane; l < L; l += 64) {
        const long v = text[(size_t)row * L + l];
        if (v > best) {  // ascending l per lane: keeps the first index of this lane's max
            best = v;
            bi = l;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if (lane == 0) idx[row] = bi;
}

// X16: x is bf16 (the image tower's bf16 residual stream); the gathered rows are fp32 either way
template <bool X16>
__global__ void gather_rows_kernel(const void* __restrict__ x, const int32_t* __restrict__ idx, float* __restrict__ out,
                                   int B, int L, int C) {
    const int c4n = C / 4;
    const long total = (long)B * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const long b = i / c4n;
        const int t = idx ? idx[b] : 0;
        const size_t o = ((size_t)b * L + t) * C + c;
        if constexpr (X16) {
            const bf16x4 v = *(const bf16x4*)((const bf16*)x + o);
            *(f32x4*)(out + (size_t)b * C + c) = (f32x4){bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
        } else {
            *(f32x4*)(out + (size_t)b * C + c) = *(const f32x4*)((const float*)x + o);
        }
    }
}

__global__ void scatter_rows_kernel(const float* __restrict__ d, const int32_t* __restrict__ idx, float* __restrict__ dx,
                                    bf16* __restrict__ dx16, int B, int L, int C) {
    const int c4n = C / 4;
    const long total = (long)B * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const long b = i / c4n;
        const int t = idx ? idx[b] : 0;
        const f32x4 v = *(const f32x4*)(d + (size_t)b * C + c);
        const size_t o = ((size_t)b * L + t) * C + c;
        if (dx) *(f32x4*)(dx + o) = v;
        if (dx16) {
            bf16x4 o4 = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            *(bf16x4*)(dx16 + o) = o4;
        }
    }
}

// dx[row_b] += d[b]; dx16[row_b] = bf16 of the sum: the pooled rows' share of a gradient added into the all-row result (rows are distinct)
__global__ void scatter_add_rows_kernel(const float* __restrict__ d, const int32_t* __restrict__ idx, float* __restrict__ dx,
                                        bf16* __restrict__ dx16, int B, int L, int C) {
    const int c4n = C / 4;
    const long total = (long)B * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const long b = i / c4n;
        const int t = idx ? idx[b] : 0;
        const size_t o = ((size_t)b * L + t) * C + c;
        f32x4 v = *(const f32x4*)(d + (size_t)b * C + c);
        if (dx) {
            v = v + *(const f32x4*)(dx + o);
            *(f32x4*)(dx + o) = v;
        } else {  // bf16 gradient stream: the sum is formed on the bf16 value (rounded once more, on the B pooled rows only)
            const bf16x4 h = *(const bf16x4*)(dx16 + o);
            v = v + (f32x4){bf2f(h[0]), bf2f(h[1]), bf2f(h[2]), bf2f(h[3])};
        }
        if (dx16) {
            bf16x4 o4 = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            *(bf16x4*)(dx16 + o) = o4;
        }
    }
}

// ---- F.normalize ---------------------------------------------------------------------------------------
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, bf16* __restrict__ y16,
                                  float* __restrict__ inv_norm, int B, int E, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= B) return;
    float s = 0.f;
    for (int c = lane; c < E; c += 64) {
        const float v = x[(size_t)row * E + c];
        s += v * v;
    }
    s = wave_sum(s);
    const float inv = 1.0f / fmaxf(sqrtf(s), eps);
    if (lane == 0) inv_norm[row] = inv;
    for (int c = lane; c < E; c += 64) {
        const float v = x[(size_t)row * E + c] * inv;
        y[(size_t)row * E + c] = v;
        if (y16) y16[(size_t)row * E + c] = f2bf(v);
    }
}

__global__ void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ inv_norm,
                                  float* __restrict__ dx, int B, int E) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= B) return;
    float s = 0.f;
    for (int c = lane; c < E; c += 64) s += dy[(size_t)row * E + c] * y[(size_t)row * E + c];
    s = wave_sum(s);
    const float inv = inv_norm[row];
    for (int c = lane; c < E; c += 64) dx[(size_t)row * E + c] = (dy[(size_t)row * E + c] - y[(size_t)row * E + c] * s) * inv;
}

int grid_for(long items, int block) {
    long g = (items + block - 1) / block;
    return (int)(g < 8192 ? (g > 0 ? g : 1) : 8192);
}

}  // namespace

extern "C" int ocn_patchify(const void* image, int image_is_bf16, void* patches, int B, int H, int W, int P, int Kpad,
                            ocn_stream_t stream) {
    OCN_CHECK_ARG(image && patches, "ocn_patchify: null operand");
    OCN_CHECK_ARG(B > 0 && P > 0 && H % P == 0 && W % P == 0 && P % 2 == 0, "ocn_patchify: bad geometry H=%d W=%d P=%d", H, W, P);
    OCN_CHECK_ARG(Kpad >= 3 * P * P && Kpad % 4 == 0, "ocn_patchify: Kpad=%d too small / not a multiple of 4", Kpad);
    hipStream_t st = (hipStream_t)stream;
    const long rows = (long)B * (H / P) * (W / P);
    if (P % 4 == 0) {
        const long total = rows * (Kpad / 4);
        if (image_is_bf16) hipLaunchKernelGGL((patchify_kernel<bf16, 4>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const bf16*)image, (bf16*)patches, B, H, W, P, Kpad, total);
        else hipLaunchKernelGGL((patchify_kernel<float, 4>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const float*)image, (bf16*)patches, B, H, W, P, Kpad, total);
    } else {
        const long total = rows * (Kpad / 2);
        if (image_is_bf16) hipLaunchKernelGGL((patchify_kernel<bf16, 2>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const bf16*)image, (bf16*)patches, B, H, W, P, Kpad, total);
        else hipLaunchKernelGGL((patchify_kernel<float, 2>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const float*)image, (bf16*)patches, B, H, W, P, Kpad, total);
    }
    OCN_CHECK_LAUNCH("ocn_patchify");
    return OCN_OK;
}

extern "C" int ocn_patchify_u8(const void* image_u8, int hwc, const float* mean3, const float* std3, void* patches, int B, int H, int W,
                               int P, int Kpad, ocn_stream_t stream) {
    OCN_CHECK_ARG(image_u8 && patches && mean3 && std3, "ocn_patchify_u8: null operand");
    OCN_CHECK_ARG(B > 0 && P > 0 && H % P == 0 && W % P == 0 && P % 2 == 0, "ocn_patchify_u8: bad geometry H=%d W=%d P=%d", H, W, P);
    OCN_CHECK_ARG(Kpad >= 3 * P * P && Kpad % 4 == 0, "ocn_patchify_u8: Kpad=%d too small / not a multiple of 4", Kpad);
    NormC nc;
    for (int c = 0; c < 3; ++c) {  // mean3 / std3 are HOST pointers (three floats each)
        OCN_CHECK_ARG(std3[c] > 0.f, "ocn_patchify_u8: std must be positive");
        nc.scale[c] = 1.0f / (255.0f * std3[c]);
        nc.shift[c] = -mean3[c] / std3[c];
    }
    if (hwc && P % 16 == 0 && Kpad % 8 == 0 && ((uintptr_t)image_u8 & 15) == 0 && ((uintptr_t)patches & 15) == 0) {
        const long npatch = (long)B * (H / P) * (W / P);
        hipLaunchKernelGGL(patchify_u8_hwc_kernel, dim3((unsigned)(npatch < 16384 ? npatch : 16384)), dim3(256), 3 * P * P, (hipStream_t)stream,
                           (const unsigned char*)image_u8, (bf16*)patches, H, W, P, Kpad, npatch, nc);
        OCN_CHECK_LAUNCH("ocn_patchify_u8");
        return OCN_OK;
    }
    const long total = (long)B * (H / P) * (W / P) * (Kpad / 2);
    hipLaunchKernelGGL(patchify_u8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)image_u8,
                       (bf16*)patches, B, H, W, P, Kpad, total, hwc, nc);
    OCN_CHECK_LAUNCH("ocn_patchify_u8");
    return OCN_OK;
}

extern "C" int ocn_embed_assemble_fwd(const float* patch_out, const float* cls, const float* pos, float* emb, int B, int G,
                                      int C, ocn_stream_t stream) {
    OCN_CHECK_ARG(patch_out && cls && pos && emb, "ocn_embed_assemble_fwd: null operand");
    OCN_CHECK_ARG(B > 0 && G > 0 && C % 4 == 0, "ocn_embed_assemble_fwd: bad shape");
    const long total = (long)B * (G + 1) * (C / 4);
    hipLaunchKernelGGL(embed_assemble_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, patch_out, cls, pos, emb, B, G, C);
    OCN_CHECK_LAUNCH("ocn_embed_assemble_fwd");
    return OCN_OK;
}

extern "C" int ocn_embed_assemble_bwd(const float* demb, void* dpatch_bf16, float* dpos, float* dcls, int B, int G, int C, int deterministic,
                                      ocn_stream_t stream) {
    OCN_CHECK_ARG(demb && dpatch_bf16 && dpos && dcls, "ocn_embed_assemble_bwd: null operand");
    OCN_CHECK_ARG(B > 0 && G > 0 && C % 4 == 0, "ocn_embed_assemble_bwd: bad shape");
    const int bchunk = deterministic ? B : 32;  // deterministic: ONE batch chunk -- every element of dpos / dcls has a single writer summing in batch order
    dim3 grid(ocn_cdiv((long)(G + 1) * (C / 4), 256), ocn_cdiv(B, bchunk));
    hipLaunchKernelGGL(embed_assemble_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, demb, (bf16*)dpatch_bf16, dpos, dcls, B, G, C, bchunk);
    OCN_CHECK_LAUNCH("ocn_embed_assemble_bwd");
    return OCN_OK;
}

extern "C" int ocn_token_embed_fwd(const int64_t* text, const float* table, const float* pos, float* x, int B, int L, int C,
                                   int vocab, ocn_stream_t stream) {
    OCN_CHECK_ARG(text && table && pos && x, "ocn_token_embed_fwd: null operand");
    OCN_CHECK_ARG(B > 0 && L > 0 && C % 4 == 0 && vocab > 0, "ocn_token_embed_fwd: bad shape");
    const long total = (long)B * L * (C / 4);
    hipLaunchKernelGGL(token_embed_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, text, table, pos, x, B, L, C, vocab);
    OCN_CHECK_LAUNCH("ocn_token_embed_fwd");
    return OCN_OK;
}

extern "C" int ocn_token_embed_bwd(const int64_t* text, const float* dx, float* dtable, float* dpos, int B, int L, int C,
                                   int vocab, ocn_stream_t stream) {
    OCN_CHECK_ARG(text && dx && dtable && dpos, "ocn_token_embed_bwd: null operand");
    OCN_CHECK_ARG(B > 0 && L > 0 && C % 4 == 0 && vocab > 0, "ocn_token_embed_bwd: bad shape");
    const int bchunk = 64;
    dim3 grid(ocn_cdiv((long)L * C, 256), ocn_cdiv(B, bchunk));
    hipLaunchKernelGGL(token_embed_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, text, dx, dtable, dpos, B, L, C, vocab, bchunk);
    OCN_CHECK_LAUNCH("ocn_token_embed_bwd");
    return OCN_OK;
}

extern "C" int ocn_token_embed_bwd_sorted(const int64_t* sorted_tokens, const int64_t* order, const void* dx, int dx_is_bf16, float* dtable, float* dpos,
                                          int B, int L, int C, int vocab, int deterministic, ocn_stream_t stream) {
    OCN_CHECK_ARG(sorted_tokens && order && dx && dtable && dpos, "ocn_token_embed_bwd_sorted: null operand");
    OCN_CHECK_ARG(B > 0 && L > 0 && C % 4 == 0 && vocab > 0, "ocn_token_embed_bwd_sorted: bad shape");
    OCN_CHECK_ARG(((uintptr_t)dx & 15) == 0 && ((uintptr_t)dtable & 15) == 0, "ocn_token_embed_bwd_sorted: operands must be 16-byte aligned");
    const long n = (long)B * L;
    const int CH = 64, bchunk = deterministic ? B : 128;  // deterministic: one batch chunk (a single writer per element of dpos) and whole runs per workgroup
    const dim3 g1((unsigned)ocn_cdiv(n, CH)), g2(ocn_cdiv((long)L * (C / 4), 256), ocn_cdiv(B, bchunk));
    hipStream_t st = (hipStream_t)stream;
    if (dx_is_bf16) {
        hipLaunchKernelGGL(token_embed_bwd_sorted_kernel<bf16>, g1, dim3(128), 0, st, sorted_tokens, order, (const bf16*)dx, dtable, n, C, vocab, CH, deterministic);
        hipLaunchKernelGGL(pos_grad_kernel<bf16>, g2, dim3(256), 0, st, (const bf16*)dx, dpos, B, L, C, bchunk);
    } else {
        hipLaunchKernelGGL(token_embed_bwd_sorted_kernel<float>, g1, dim3(128), 0, st, sorted_tokens, order, (const float*)dx, dtable, n, C, vocab, CH, deterministic);
        hipLaunchKernelGGL(pos_grad_kernel<float>, g2, dim3(256), 0, st, (const float*)dx, dpos, B, L, C, bchunk);
    }
    OCN_CHECK_LAUNCH("ocn_token_embed_bwd_sorted");
    return OCN_OK;
}

// nn.Embedding raises on ids outside [0, vocab) (model.py:399 -> torch embedding); the embedding kernels here clamp so that a bad id can
// never read outside the table, and this one-workgroup pass COUNTS such ids so that the host can raise like the reference does
// (open_clip_amd/model.py::_TextPack reads the count back with the packed row count it waits for anyway).
__global__ __launch_bounds__(1024) void token_range_kernel(const int64_t* __restrict__ text, long n, int vocab, int32_t* __restrict__ bad) {
    __shared__ int wsum[16];
    int c = 0;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const long t = text[i];
        c += (t < 0 || t >= vocab) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += wsum[w];
        *bad = t;
    }
}

extern "C" int ocn_token_range_check(const int64_t* text, long n, int vocab, int32_t* bad_count, ocn_stream_t stream) {
    OCN_CHECK_ARG(text && bad_count && n > 0 && vocab > 0, "ocn_token_range_check: bad arguments");
    hipLaunchKernelGGL(token_range_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, text, n, vocab, bad_count);
    OCN_CHECK_LAUNCH("ocn_token_range_check");
    return OCN_OK;
}

// Buckets of a packed batch for the attention launches (attention.hip::seq_index): order = the B sequence ids grouped by
// nb = ceil(len / 32) ascending, counts[k] = how many sequences have nb == k + 1.  One workgroup; the position of a sequence INSIDE its
// bucket comes from an LDS atomic (any order is correct: a workgroup's result does not depend on which workgroup computes it).
__global__ __launch_bounds__(1024) void seq_bucket_kernel(const int32_t* __restrict__ seq_off, int32_t* __restrict__ order,
                                                          int32_t* __restrict__ counts, int B, int nbuckets) {
    __shared__ int cnt[16], cur[16];
    if (threadIdx.x < 16) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += 1024) atomicAdd(&cnt[(seq_off[b + 1] - seq_off[b] + 31) / 32 - 1], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int k = 0; k < nbuckets; ++k) {
            counts[k] = cnt[k];
            cur[k] = run;
            run += cnt[k];
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += 1024) order[atomicAdd(&cur[(seq_off[b + 1] - seq_off[b] + 31) / 32 - 1], 1)] = b;
}

extern "C" int ocn_seq_bucket_plan(const int32_t* seq_off, int32_t* order, int32_t* counts, int B, int Lmax, ocn_stream_t stream) {
    OCN_CHECK_ARG(seq_off && order && counts && B > 0 && Lmax > 0, "ocn_seq_bucket_plan: bad arguments");
    OCN_CHECK_ARG(Lmax <= 512, "ocn_seq_bucket_plan: Lmax = %d (at most 16 buckets of 32 rows)", Lmax);
    hipLaunchKernelGGL(seq_bucket_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, seq_off, order, counts, B, ocn_cdiv(Lmax, 32));
    OCN_CHECK_LAUNCH("ocn_seq_bucket_plan");
    return OCN_OK;
}

extern "C" int ocn_seq_pack_plan(const int64_t* text, int32_t* eot, int32_t* seq_off, int32_t* last_row, int B, int L, ocn_stream_t stream) {
    OCN_CHECK_ARG(text && eot && seq_off && last_row && B > 0 && L > 0, "ocn_seq_pack_plan: bad arguments");
    OCN_CHECK_ARG((long)B * L < 0x7fffffffL, "ocn_seq_pack_plan: B*L = %ld rows exceed int32 offsets", (long)B * L);
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(ocn_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, text, eot, B, L);
    hipLaunchKernelGGL(seq_plan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, eot, seq_off, last_row, B);
    OCN_CHECK_LAUNCH("ocn_seq_pack_plan");
    return OCN_OK;
}

extern "C" int ocn_seq_pack_rows(const int64_t* text, const int32_t* seq_off, int64_t* tokens, int32_t* posidx, int B, int L,
                                 ocn_stream_t stream) {
    OCN_CHECK_ARG(text && seq_off && tokens && posidx && B > 0 && L > 0, "ocn_seq_pack_rows: bad arguments");
    hipLaunchKernelGGL(seq_pack_rows_kernel, dim3(grid_for((long)B * L, 256)), dim3(256), 0, (hipStream_t)stream, text, seq_off, tokens, posidx, B, L);
    OCN_CHECK_LAUNCH("ocn_seq_pack_rows");
    return OCN_OK;
}

extern "C" int ocn_token_embed_fwd_rows(const int64_t* tokens, const int32_t* posidx, const float* table, const float* pos, float* x, long M,
                                        int C, int vocab, ocn_stream_t stream) {
    OCN_CHECK_ARG(tokens && posidx && table && pos && x, "ocn_token_embed_fwd_rows: null operand");
    OCN_CHECK_ARG(M > 0 && C % 4 == 0 && vocab > 0, "ocn_token_embed_fwd_rows: bad shape");
    hipLaunchKernelGGL(token_embed_fwd_rows_kernel, dim3(grid_for(M * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, tokens, posidx, table, pos, x, M, C, vocab);
    OCN_CHECK_LAUNCH("ocn_token_embed_fwd_rows");
    return OCN_OK;
}

extern "C" int ocn_token_embed_bwd_sorted_varlen(const int64_t* sorted_tokens, const int64_t* order, const void* dx, int dx_is_bf16, float* dtable,
                                                 float* dpos, const int32_t* seq_off, int B, int L, long M, int C, int vocab, int deterministic,
                                                 ocn_stream_t stream) {
    OCN_CHECK_ARG(sorted_tokens && order && dx && dtable && dpos && seq_off, "ocn_token_embed_bwd_sorted_varlen: null operand");
    OCN_CHECK_ARG(B > 0 && L > 0 && M > 0 && C % 4 == 0 && vocab > 0, "ocn_token_embed_bwd_sorted_varlen: bad shape");
    OCN_CHECK_ARG(((uintptr_t)dx & 15) == 0 && ((uintptr_t)dtable & 15) == 0, "ocn_token_embed_bwd_sorted_varlen: operands must be 16-byte aligned");
    const int CH = 64, bchunk = deterministic ? B : 128;
    const dim3 g1((unsigned)ocn_cdiv(M, CH)), g2(ocn_cdiv((long)L * (C / 4), 256), ocn_cdiv(B, bchunk));
    hipStream_t st = (hipStream_t)stream;
    if (dx_is_bf16) {
        hipLaunchKernelGGL(token_embed_bwd_sorted_kernel<bf16>, g1, dim3(128), 0, st, sorted_tokens, order, (const bf16*)dx, dtable, M, C, vocab, CH, deterministic);
        hipLaunchKernelGGL(pos_grad_varlen_kernel<bf16>, g2, dim3(256), 0, st, (const bf16*)dx, dpos, seq_off, B, L, C, bchunk);
    } else {
        hipLaunchKernelGGL(token_embed_bwd_sorted_kernel<float>, g1, dim3(128), 0, st, sorted_tokens, order, (const float*)dx, dtable, M, C, vocab, CH, deterministic);
        hipLaunchKernelGGL(pos_grad_varlen_kernel<float>, g2, dim3(256), 0, st, (const float*)dx, dpos, seq_off, B, L, C, bchunk);
    }
    OCN_CHECK_LAUNCH("ocn_token_embed_bwd_sorted_varlen");
    return OCN_OK;
}

extern "C" int ocn_argmax_rows(const int64_t* text, int32_t* idx, int B, int L, ocn_stream_t stream) {
    OCN_CHECK_ARG(text && idx && B > 0 && L > 0, "ocn_argmax_rows: bad arguments");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(ocn_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, text, idx, B, L);
    OCN_CHECK_LAUNCH("ocn_argmax_rows");
    return OCN_OK;
}

extern "C" int ocn_gather_rows(const void* x, int x_is_bf16, const int32_t* idx, float* out, int B, int L, int C, ocn_stream_t stream) {
    OCN_CHECK_ARG(x && out && B > 0 && L >= 0 && (L > 0 || idx) && C % 4 == 0, "ocn_gather_rows: bad arguments");
    if (x_is_bf16) hipLaunchKernelGGL(gather_rows_kernel<true>, dim3(grid_for((long)B * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, idx, out, B, L, C);
    else hipLaunchKernelGGL(gather_rows_kernel<false>, dim3(grid_for((long)B * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, idx, out, B, L, C);
    OCN_CHECK_LAUNCH("ocn_gather_rows");
    return OCN_OK;
}

// the same gather for a bf16 matrix (the attention output rows of the pooled tokens: A operand of the last block's out-proj)
__global__ void gather_rows_bf16_kernel(const bf16* __restrict__ x, const int32_t* __restrict__ idx, bf16* __restrict__ out, int B, int L, int C) {
    const int c8n = C / 8;
    const long total = (long)B * c8n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c8n) * 8;
        const long b = i / c8n;
        const int t = idx ? idx[b] : 0;
        *(bf16x8*)(out + (size_t)b * C + c) = *(const bf16x8*)(x + ((size_t)b * L + t) * C + c);
    }
}

extern "C" int ocn_gather_rows_bf16(const void* x, const int32_t* idx, void* out, int B, int L, int C, ocn_stream_t stream) {
    OCN_CHECK_ARG(x && out && B > 0 && L >= 0 && (L > 0 || idx) && C % 8 == 0, "ocn_gather_rows_bf16: bad arguments");
    hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(grid_for((long)B * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, idx,
                       (bf16*)out, B, L, C);
    OCN_CHECK_LAUNCH("ocn_gather_rows_bf16");
    return OCN_OK;
}

extern "C" int ocn_scatter_rows(const float* d, const int32_t* idx, float* dx, void* dx_bf16, int B, int L, int C,
                                ocn_stream_t stream) {
    OCN_CHECK_ARG(d && (dx || dx_bf16) && B > 0 && L >= 0 && (L > 0 || idx) && C % 4 == 0, "ocn_scatter_rows: bad arguments");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((long)B * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, d, idx, dx, (bf16*)dx_bf16, B, L, C);
    OCN_CHECK_LAUNCH("ocn_scatter_rows");
    return OCN_OK;
}

extern "C" int ocn_scatter_add_rows(const float* d, const int32_t* idx, float* dx, void* dx_bf16, int B, int L, int C, ocn_stream_t stream) {
    OCN_CHECK_ARG(d && (dx || dx_bf16) && B > 0 && L >= 0 && (L > 0 || idx) && C % 4 == 0, "ocn_scatter_add_rows: bad arguments");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for((long)B * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, d, idx, dx, (bf16*)dx_bf16, B, L, C);
    OCN_CHECK_LAUNCH("ocn_scatter_add_rows");
    return OCN_OK;
}

extern "C" int ocn_l2norm_fwd(const float* x, float* y, void* y_bf16, float* inv_norm, int B, int E, float eps,
                              ocn_stream_t stream) {
    OCN_CHECK_ARG(x && y && inv_norm && B > 0 && E > 0, "ocn_l2norm_fwd: bad arguments");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(ocn_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, x, y, (bf16*)y_bf16, inv_norm, B, E, eps);
    OCN_CHECK_LAUNCH("ocn_l2norm_fwd");
    return OCN_OK;
}

extern "C" int ocn_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int B, int E,
                              ocn_stream_t stream) {
    OCN_CHECK_ARG(dy && y && inv_norm && dx && B > 0 && E > 0, "ocn_l2norm_bwd: bad arguments");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(ocn_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, dy, y, inv_norm, dx, B, E);
    OCN_CHECK_LAUNCH("ocn_l2norm_bwd");
    return OCN_OK;
}
