// Hand-scheduled TN (weight-gradient) GEMM for gfx950:  dW[n,k] += alpha * sum_m A[m,n] * B[m,k]  (+ dbias[n] += alpha * sum_m A[m,n]).
//
// Same geometry as gemm_tn_ring_kernel (gemm.hip): a 256 (n) x 256 (k) tile of dW per workgroup, 8 waves (2 x 4, wave tile
// 128 x 64 = 4 x 2 MFMA blocks), 32 reduction rows per step, 4-slot LDS ring (4 x 32 KiB: [32 m][256 n] of A and
// [32 m][256 k] of B as 512-byte rows, 16-byte chunks XOR-swizzled by (row & 3) << 2 on the SOURCE address), split over M so
// that one launch fills the chip; fp32 atomics into dW.  What changed, and why (disassembly of the ring kernel):
//   * with compiler-visible global_load_lds + ds_read_tr builtins hipcc waits vmcnt(0) before the fragment reads of every
//     step (it cannot prove the reads do not alias an in-flight LDS-DMA), so the "3 stages ahead" ring never ran ahead.
//     Here the LDS-DMA is an inline-asm "buffer_load_dwordx4 ... lds" and the transposing fragment reads are inline-asm
//     ds_read_b64_tr_b16, with ONE counted vmcnt(4) and one raw s_barrier per step;
//   * buffer descriptors are based at the split's first row with num_records = the split's bytes: reduction rows beyond
//     m_end and (offset forced out of range) columns beyond N / K read as ZERO in hardware -> no per-DMA select, no tail code
//     (steps beyond the split's last row multiply zeros);
//   * the bias gradient (column sums of A by an MFMA against ones) is time-sliced over the tiles_k workgroups that share an
//     A strip: workgroup tk does it on steps kt == tk (mod tiles_k), so every workgroup carries the same 1/(8 tiles_k) extra
//     MFMA work instead of one workgroup in tiles_k carrying 1/8.
// Step kt (stage kt lives in ring slot kt & 3; X = fragments of sub-step 0, Y = sub-step 1):
//   wait vmcnt(4) [own DMAs of stage kt+1 landed] -> s_barrier [everybody's landed; everybody finished reading stage kt-1]
//   issue reads Y(kt) | 8 MFMAs on X(kt) with the 4 DMA pieces of stage kt+3 (into the slot of stage kt-1) between them
//   wait lgkmcnt(0) | issue reads X(kt+1) | 8 MFMAs on Y(kt) | wait lgkmcnt(0)
#include "gemm_args.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int STAGE = 32768, HALF = 16384, LDS_BYTES = 4 * STAGE;

struct Frag {
    u32x2 lo, hi;  // m rows {0..3} and {4..7} of this lane's 8 reduction slots
};
struct FragSet {
    Frag a[4], b[2];
};

OCN_DEV bf16x8 cat(const Frag& f) {
    const u32x4 v = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}

#define SB() __builtin_amdgcn_sched_barrier(0)
// two transposing reads (rows r..r+3 and r+4..r+7) of one 32-wide block; early-clobber: the address is read twice
#define TRR(F, ADDR, OFF)                                                                              \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"          \
                 : "=&v"((F).lo), "=&v"((F).hi)                                                        \
                 : "v"(ADDR), "i"(OFF), "i"((OFF) + 2048))
// the wait names every destination "+v": the compiler cannot copy / combine a fragment before its data has landed
#define WAIT_SET(S)                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                  \
                 : "+v"((S).a[0].lo), "+v"((S).a[0].hi), "+v"((S).a[1].lo), "+v"((S).a[1].hi), "+v"((S).a[2].lo),          \
                   "+v"((S).a[2].hi), "+v"((S).a[3].lo), "+v"((S).a[3].hi), "+v"((S).b[0].lo), "+v"((S).b[0].hi),          \
                   "+v"((S).b[1].lo), "+v"((S).b[1].hi))

// RESCUE (ocn_set_tile_rescue, the multi-GPU form; protocol: gemm_nt5.hip): a workgroup's share is its M-chunk in a.pieces pieces of a.piece_rows rows.
// The owner adds RESCUE_BIG to its counter when it starts and takes the chunk from the first piece no finisher has claimed to the end (one epilogue);
// a finisher claims ONE piece of a chunk nobody has started and adds its partial tile with the same fp32 atomics.  The bias slices stay consistent:
// step s of a chunk (counted from the chunk's first row) belongs to the workgroup with tk == s (mod tiles_k), whoever computes it.
constexpr int RESCUE_BIG = 1 << 20;
template <bool BIAS, bool RESCUE = false>
__global__ __launch_bounds__(512, 2) void gemm_tn5_kernel(GemmTnArgs a_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int unit = xcd_remap(blockIdx.x, a_in.nwg);  // the (tile, M-split) unit this pass works on
    int piece0 = 0;                              // RESCUE: its first piece
    int* const box = (int*)(smem + LDS_BYTES);   // RESCUE: workgroup-wide mailbox behind the ring (the launch asks for 64 bytes more)
    // the owner's claim is issued here and read behind the first prologue, whose operand fetches cover its latency (gemm_nt5.hip)
    int claim = 0;
    bool fresh = RESCUE;
    if constexpr (RESCUE) if (threadIdx.x == 0) {
        // inline asm: hipcc's own atomicAdd is consumed on the spot (its wave-reduction wrapper reads the result back at once)
        const int* cptr = a_in.rescue + unit * OCN_RESCUE_STRIDE;
        asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "=v"(claim) : "v"(0), "v"(RESCUE_BIG), "s"(cptr) : "memory");
    }
  for (int pass = 0;; ++pass) {  // one pass without RESCUE; with it: the own chunk, then one rescued piece per pass
    GemmTnArgs a = a_in;
    const int wid = unit;
    const int ntile = a.ntile_all;
    const int split = wid / ntile;
    int tile = wid % ntile;
    // Two problems in one launch (ocn_gemm_tn_accum2: the out-proj and QKV wgrads of a block share their rows and their K): a small dW
    // alone needs 28-64 M-splits to fill the chip and then spends 26-37 % of its time in 28-64-way contended atomics
    // (profiles/r01_tn5_epilogue_ablation.txt); together with its bigger sibling the pair has 36 tiles and 7 splits.
    if (tile >= a.ntile1) {
        tile -= a.ntile1;
        a.A = a.A2; a.B = a.B2; a.dW = a.dW2; a.dbias = a.dbias2;
        a.lda = a.lda2; a.ldb = a.ldb2; a.ldw = a.ldw2; a.N = a.N2;
    }
    const int tn = tile / a.tiles_k, tk = tile % a.tiles_k;
    const int n0 = tn * 256, k0 = tk * 256;
    int m_begin = split * a.chunk;
    int m_end = min(a.M, m_begin + a.chunk);
    if constexpr (RESCUE) {
        m_begin += piece0 * a.piece_rows;
        if (pass > 0) m_end = min(m_end, m_begin + a.piece_rows);
    }
    const int rows = max(m_end - m_begin, 0);
    const int nk = (rows + 31) / 32;
   if (!RESCUE || rows > 0) {
    const int wn = wave >> 2, wk = wave & 3;
    const unsigned lds_base = (unsigned)(size_t)(OCN_LDS char*)smem;

    // ---- descriptors: based at the split's first row; everything past its last row reads as zero -------------------------
    auto make_desc = [&](const bf16* base, int ld) -> u32x4 {
        const unsigned long long p = (unsigned long long)(base + (size_t)m_begin * ld);
        const long bytes = (long)rows * ld * 2;  // < 2^31 (checked by the launcher)
        u32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
        r[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu);
        r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);
        r[3] = 0x00020000u;
        return r;
    };
    const u32x4 dA = make_desc(a.A, a.lda), dB = make_desc(a.B, a.ldb);

    // ---- DMA: wave w moves rows {2w, 2w+1} (piece 0) and {16+2w, 17+2w} (piece 1) of each operand's [32][256] stage image ---
    unsigned voA[2], voB[2];  // per-lane byte offsets into the descriptors (advance by 32 rows per step)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave + j * 8) * 2 + (lane >> 5);
        const int c = ((lane & 31) ^ ((r & 3) << 2)) * 8;  // source-side chunk swizzle
        voA[j] = (n0 + c) < a.N ? (unsigned)(r * a.lda + n0 + c) * 2u : 0x80000000u;  // out-of-range columns stay out of range
        voB[j] = (k0 + c) < a.K ? (unsigned)(r * a.ldb + k0 + c) * 2u : 0x80000000u;
    }
    const unsigned stepA = (unsigned)(32 * a.lda * 2), stepB = (unsigned)(32 * a.ldb * 2);
    const unsigned wave_dst = lds_base + wave * 1024;
    // one LDS-DMA piece: 64 lanes x 16 B from desc[voff] to LDS m0 + lane*16 (m0 is compiler-reserved: save / restore)
#define DMA(DESC, VOFF, DST)                                                                                              \
    {                                                                                                                     \
        unsigned keep_;                                                                                                   \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_)                                                                                       \
                     : "v"(VOFF), "s"(DESC), "s"(DST)                                                                     \
                     : "memory");                                                                                         \
    }
#define DMA_A(J, SLOT) DMA(dA, voA[J], wave_dst + (SLOT) * STAGE + (J) * 8192)
#define DMA_B(J, SLOT) DMA(dB, voB[J], wave_dst + (SLOT) * STAGE + HALF + (J) * 8192)
#define DMA_ADV()      \
    {                  \
        voA[0] += stepA; \
        voA[1] += stepA; \
        voB[0] += stepB; \
        voB[1] += stepB; \
    }

    // ---- fragment read addresses -----------------------------------------------------------------------------------------
    // 16-lane group: 4 rows x 16 columns; lane i passes the address of columns (i&3)*4.. of row i>>2 and receives column i.
    // byte = row * 512 + (chunk ^ ((row & 3) << 2)) * 16 + (i & 1) * 8, row = s*16 + h*8 + (i>>2) (+4), chunk = col/8 + (i&3)>>1.
    // acc index ib of this wave is A block (ib + wk) & 3 (so that block wk, the one this wave sums for dbias, is a[0]).
    unsigned vaL[4], vbL[2], vaH[4], vbH[2];  // ring slots 0-1 / 2-3 (ds offsets are 16 bits)
    {
        const int i = lane & 15, g = (lane >> 4) & 1, h = lane >> 5, i2 = i >> 2;
        const unsigned lane_off = (unsigned)((h * 8 + i2) * 512 + (2 * g + ((i & 3) >> 1)) * 16 + (i & 1) * 8);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const int blk = (ib + wk) & 3;
            vaL[ib] = lds_base + lane_off + (unsigned)((wn * 16 + ((blk ^ i2) << 2)) * 16);
            vaH[ib] = vaL[ib] + 65536u;
        }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            vbL[jb] = lds_base + lane_off + (unsigned)((((wk * 2 + jb) ^ i2) << 2) * 16);
            vbH[jb] = vbL[jb] + 65536u;
        }
    }
#define READ_SET(S, SLOT, SUB)                                                                       \
    {                                                                                                \
        constexpr int o_ = ((SLOT) & 1) * STAGE + (SUB) * 8192;                                       \
        TRR((S).a[0], ((SLOT) < 2 ? vaL[0] : vaH[0]), o_);                                            \
        TRR((S).b[0], ((SLOT) < 2 ? vbL[0] : vbH[0]), o_ + HALF);                                     \
        TRR((S).b[1], ((SLOT) < 2 ? vbL[1] : vbH[1]), o_ + HALF);                                     \
        TRR((S).a[1], ((SLOT) < 2 ? vaL[1] : vaH[1]), o_);                                            \
        TRR((S).a[2], ((SLOT) < 2 ? vaL[2] : vaH[2]), o_);                                            \
        TRR((S).a[3], ((SLOT) < 2 ? vaL[3] : vaH[3]), o_);                                            \
    }

    f32x16 acc[4][2], accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
    int bc = tk;  // steps until this workgroup's next bias step
    if constexpr (RESCUE) {
        bc = tk - (piece0 * (a.piece_rows >> 5)) % a.tiles_k;  // the pass starts at step piece0 * piece_rows / 32 of the chunk
        bc += bc < 0 ? a.tiles_k : 0;
    }

    // ---- prologue: stages 0, 1, 2 ------------------------------------------------------------------------------------------
    DMA_A(0, 0) DMA_A(1, 0) DMA_B(0, 0) DMA_B(1, 0)
    DMA_ADV()
    DMA_A(0, 1) DMA_A(1, 1) DMA_B(0, 1) DMA_B(1, 1)
    DMA_ADV()
    DMA_A(0, 2) DMA_A(1, 2) DMA_B(0, 2) DMA_B(1, 2)
    DMA_ADV()
    FragSet X, Y;
#if (OCN_PRIO_MODE & 2)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    if constexpr (RESCUE) if (fresh) {
        fresh = false;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(claim)::"memory");  // the claim (issued first) and, with it, the three stages
        if (threadIdx.x == 0) box[0] = claim;
        __syncthreads();
        const int lo = min(__builtin_amdgcn_readfirstlane(box[0]), a_in.pieces);
        if (lo > 0) {  // finishers took the head of this chunk before the workgroup got a CU: start again behind their pieces
            piece0 = lo;
            __syncthreads();
            --pass;
            continue;
        }
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    SB();
    READ_SET(X, 0, 0)
    WAIT_SET(X);
    SB();

// OCN_PRIO_MODE (compile-time experiment, profiles/r02_setprio_experiment.txt): 1 = s_setprio 1 over the MFMA work of a step,
// 2 = static priority for waves 4..7, 3 = both
#ifndef OCN_PRIO_MODE
#define OCN_PRIO_MODE 0
#endif
#if (OCN_PRIO_MODE & 1)
#define PRIO_ON() __builtin_amdgcn_s_setprio(1);
#define PRIO_OFF() __builtin_amdgcn_s_setprio(0);
#else
#define PRIO_ON()
#define PRIO_OFF()
#endif
#define MM(S, IB, JB) acc[IB][JB] = mfma32(cat((S).a[IB]), cat((S).b[JB]), acc[IB][JB]);
#define STEP(SLOT)                                                                                   \
    {                                                                                                \
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                             \
        __builtin_amdgcn_s_barrier();                                                                \
        SB();                                                                                        \
        READ_SET(Y, SLOT, 1)                                                                         \
        SB(); PRIO_ON()                                                                              \
        const bool bias_now = BIAS && (bc == 0);                                                     \
        MM(X, 0, 0) MM(X, 0, 1)                                                                      \
        SB();                                                                                        \
        DMA_A(0, ((SLOT) + 3) & 3)                                                                   \
        SB();                                                                                        \
        MM(X, 1, 0) MM(X, 1, 1)                                                                      \
        SB();                                                                                        \
        DMA_A(1, ((SLOT) + 3) & 3)                                                                   \
        SB();                                                                                        \
        MM(X, 2, 0) MM(X, 2, 1)                                                                      \
        SB();                                                                                        \
        DMA_B(0, ((SLOT) + 3) & 3)                                                                   \
        SB();                                                                                        \
        MM(X, 3, 0) MM(X, 3, 1)                                                                      \
        SB();                                                                                        \
        DMA_B(1, ((SLOT) + 3) & 3)                                                                   \
        if (bias_now) accb = mfma32(cat(X.a[0]), ones, accb);                                        \
        PRIO_OFF() SB();                                                                             \
        WAIT_SET(Y);                                                                                 \
        SB();                                                                                        \
        READ_SET(X, ((SLOT) + 1) & 3, 0)                                                             \
        SB(); PRIO_ON()                                                                              \
        MM(Y, 0, 0) MM(Y, 0, 1) MM(Y, 1, 0) MM(Y, 1, 1)                                              \
        DMA_ADV()                                                                                    \
        MM(Y, 2, 0) MM(Y, 2, 1) MM(Y, 3, 0) MM(Y, 3, 1)                                              \
        if (bias_now) accb = mfma32(cat(Y.a[0]), ones, accb);                                        \
        if (BIAS) bc = (bc == 0 ? a.tiles_k : bc) - 1;                                               \
        PRIO_OFF() SB();                                                                             \
        WAIT_SET(X);                                                                                 \
        SB();                                                                                        \
    }
    // nk is rounded up to whole groups of 4 steps: the surplus steps see all-zero stages (reads past the split's last row)
    for (int kt = 0; kt < nk; kt += 4) {
        STEP(0)
        STEP(1)
        STEP(2)
        STEP(3)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (zero-fill) DMAs must land before the LDS is released
#undef STEP
#undef MM
#undef READ_SET
#undef DMA_ADV
#undef DMA_A
#undef DMA_B
#undef DMA

    // ---- epilogue: fp32 atomics (lanes of a half-wave hit 32 consecutive k = one 128-byte line) ---------------------------
    const int lr = lane & 31;
    if constexpr (!RESCUE) {
    if (a.ablate & 1) return;
    if (a.ws) {
        // reproducible form (ocn_gemm_tn_accum_det): every split stores its tile to its own slab and tn5_reduce_kernel sums the slabs
        // in split order -- no atomics, the same bits on every run (also what removes the 20..60-way contended atomics of the small
        // out-proj shapes: -5 % there, nothing on the step, profiles/r01_tn5_two_stage_epilogue.txt)
        float* slab = a.ws + (size_t)split * a.N * a.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int blk = (i + wk) & 3;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int gk = k0 + wk * 64 + j * 32 + lr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gn = n0 + wn * 128 + blk * 32 + mfma32_row(r, lane);
                    if (gn < a.N && gk < a.K) __builtin_nontemporal_store(acc[i][j][r], slab + (size_t)gn * a.K + gk);
                }
            }
        }
        // bias partials: the tiles_k workgroups of a split that share an A strip each summed the steps kt == tk (mod tiles_k)
        if (BIAS && lr == 0) {
            float* bslab = a.ws + (size_t)a.nsplit * a.N * a.K + ((size_t)split * a.tiles_k + tk) * a.N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gn = n0 + wn * 128 + wk * 32 + mfma32_row(r, lane);
                if (gn < a.N) bslab[gn] = accb[r];
            }
        }
        return;
    }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int blk = (i + wk) & 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gk = k0 + wk * 64 + j * 32 + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gn = n0 + wn * 128 + blk * 32 + mfma32_row(r, lane);
                if (gn < a.N && gk < a.K) unsafeAtomicAdd(a.dW + (size_t)gn * a.ldw + gk, a.alpha * acc[i][j][r]);
            }
        }
    }
    if (BIAS && lr == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gn = n0 + wn * 128 + wk * 32 + mfma32_row(r, lane);
            if (gn < a.N) unsafeAtomicAdd(a.dbias + gn, a.alpha * accb[r]);
        }
    }
   }  // rows > 0
    if constexpr (!RESCUE) break;
    else {
        // ---- done with what this workgroup had: look for a chunk nobody has started (gemm_nt5.hip has the protocol) ---------------------------
        const int G = a_in.nwg;
        bool got = false;
        for (;;) {
            __syncthreads();  // everybody is out of the ring (first round) / has read the mailbox (later rounds)
            if (threadIdx.x == 0) box[1] = 0x7fffffff;
            __syncthreads();
            const int self = xcd_remap(blockIdx.x, G);
            for (int t = threadIdx.x; t < G - 1; t += 512) {
                int w = self + 1 + t;
                w -= w >= G ? G : 0;
                if (__hip_atomic_load(a_in.rescue + w * OCN_RESCUE_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a_in.pieces) atomicMin(box + 1, t);
            }
            __syncthreads();
            const int t = box[1];
            if (t == 0x7fffffff) break;
            int w = self + 1 + t;
            w -= w >= G ? G : 0;
            if (threadIdx.x == 0) box[0] = atomicAdd(a_in.rescue + w * OCN_RESCUE_STRIDE, 1);
            __syncthreads();
            const int k = __builtin_amdgcn_readfirstlane(box[0]);
            if (k < a_in.pieces) {
                unit = __builtin_amdgcn_readfirstlane(w);
                piece0 = k;
                got = true;
                break;
            }
        }
        if (!got) break;
        __syncthreads();  // the mailbox has been read: the ring is the DMA's again
    }
  }  // pass
}

// dW[n,k] += alpha * sum_s ws[s][n][k] (slabs summed in split order);  dbias[n] += alpha * sum_p bias_slabs[p][n]
__global__ __launch_bounds__(256) void tn5_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW, int N, int K, int ldw,
                                                         int nsplit, float alpha, float* __restrict__ dbias, int nbias_parts) {
    const long total4 = (long)N * K / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < nsplit; ++s) acc = acc + __builtin_nontemporal_load((const f32x4*)(ws + (size_t)s * N * K) + i);
        const long e = i * 4;
        const int n = (int)(e / K), k = (int)(e % K);
        float* d = dW + (size_t)n * ldw + k;
        *(f32x4*)d = *(const f32x4*)d + acc * alpha;
    }
    if (dbias) {
        const float* bs = ws + (size_t)nsplit * N * K;
        for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long)gridDim.x * blockDim.x) {
            float acc = 0.f;
            for (int p = 0; p < nbias_parts; ++p) acc += bs[(size_t)p * N + n];
            dbias[n] += acc * alpha;
        }
    }
}

int g_tn5_num_cu = 0;

int tn5_splits(int M, int N, int K, int num_cu, int over) {
    const int ntile = ocn_cdiv(N, 256) * ocn_cdiv(K, 256);
    const int msteps = ocn_cdiv(M, 32);
    int splits = num_cu / ntile;
    if (over > 1 && splits >= 1 && splits <= 10) splits *= over;
    if (splits < 1) splits = 1;
    if (splits > msteps) splits = msteps;
    const int chunk = ocn_cdiv(msteps, splits) * 32;
    return ocn_cdiv(M, chunk);
}

}  // namespace
extern int g_ocn_tuning[16];

// Static launch, or -- ocn_set_tile_rescue(1), atomic epilogue only -- the rescue form: every M-chunk in up to 8 pieces of whole 128-row step groups
static void tn5_launch(GemmTnArgs& a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_tn5_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 64);
        (void)hipFuncSetAttribute((const void*)gemm_tn5_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 64);
        attr_set = true;
    }
    a.rescue = (a.ws || a.ablate || a.nwg < 2) ? nullptr : ocn_rescue_board(st, a.nwg);
    a.piece_rows = ocn_cdiv(ocn_cdiv(a.chunk, 8), 128) * 128;
    a.pieces = ocn_cdiv(a.chunk, a.piece_rows);
    if (a.rescue) {
        if (a.dbias) hipLaunchKernelGGL((gemm_tn5_kernel<true, true>), dim3(a.nwg), dim3(512), LDS_BYTES + 64, st, a);
        else hipLaunchKernelGGL((gemm_tn5_kernel<false, true>), dim3(a.nwg), dim3(512), LDS_BYTES + 64, st, a);
        return;
    }
    if (a.dbias) hipLaunchKernelGGL(gemm_tn5_kernel<true>, dim3(a.nwg), dim3(512), LDS_BYTES, st, a);
    else hipLaunchKernelGGL(gemm_tn5_kernel<false>, dim3(a.nwg), dim3(512), LDS_BYTES, st, a);
}

static int tn5_cus() {
    if (g_tn5_num_cu == 0) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_tn5_num_cu = n;
    }
    return g_tn5_num_cu;
}

// scratch of the reproducible launch: one fp32 slab of dW per M-split + one bias row per (split, k-tile); 0 = shape not taken by this
// kernel (the caller uses the general kernel with one split)
long ocn_tn5_workspace_bytes(int M, int N, int K) {
    if (N % 8 || K % 8 || (long)M * N * K < (1L << 31) || N < 256 || K < 256) return 0;
    const int splits = tn5_splits(M, N, K, tn5_cus(), 1);
    return ((long)splits * N * K + (long)splits * ocn_cdiv(K, 256) * N) * 4;
}

int ocn_launch_tn5(GemmTnArgs a, hipStream_t st) {
    if (a.N % 8 || a.K % 8 || a.lda % 8 || a.ldb % 8) return 1;
    if (g_tn5_num_cu == 0) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_tn5_num_cu = n;
    }
    a.ablate = g_ocn_tuning[4];
    a.tiles_n = ocn_cdiv(a.N, 256);
    a.tiles_k = ocn_cdiv(a.K, 256);
    const int ntile = a.tiles_n * a.tiles_k;
    const int msteps = ocn_cdiv(a.M, 32);
    // developer knob 15 = n: size the grid for n CUs (a stream restricted to a CU subset, tools/cu_mask_probe.py)
    const int cus = (g_ocn_tuning[15] > 0 && g_ocn_tuning[15] < g_tn5_num_cu) ? g_ocn_tuning[15] : g_tn5_num_cu;
    int splits = cus / ntile;  // one workgroup per CU ...
    // ... developer knob 11 = k: k per CU when the M-chunks stay long (<= 10 splits).  A workgroup that starts late because its CU
    // is held by another stream's kernel then costs 1/k as much (one CU held: +43 % at k = 1, +7 % at k = 2;
    // profiles/r01_persistent_gemm_occupancy_hazard.txt), but the finer split costs 4-8 % in steady state on exactly those big
    // shapes (profiles/r01_tn5_split_sweep.txt) -- more than collectives that are active for a few percent of a step can take.
    const int over = g_ocn_tuning[11] > 0 ? g_ocn_tuning[11] : 1;
    if (over > 1 && splits >= 1 && splits <= 10) splits *= over;
    if (splits < 1) splits = 1;
    if (splits > msteps) splits = msteps;
    const long ldmax = a.lda > a.ldb ? a.lda : a.ldb;
    a.chunk = ocn_cdiv(msteps, splits) * 32;
    // 32-bit buffer offsets inside an M-chunk (incl. the run-ahead past m_end): a chunk that would pass 2^31 bytes is cut further (round 6: the loss's
    // G^T product at N = 32768 -- 32768 rows of 64 KiB -- used to fall back to the general kernel here, at half the speed)
    while ((long)(a.chunk + 128) * ldmax * 2 >= 0x7fffffffL && splits < msteps && !a.ws) {
        ++splits;
        a.chunk = ocn_cdiv(msteps, splits) * 32;
    }
    splits = ocn_cdiv(a.M, a.chunk);
    a.nwg = splits * ntile;
    if ((long)(a.chunk + 128) * ldmax * 2 >= 0x7fffffffL) return 1;
    // (measured and not adopted, round 6: a plain read-add-store epilogue for launches with ONE M-split, where no other workgroup adds into a tile --
    // 159 -> 180 us on the loss's [32768 x 512] product: the dependent loads cost more than the uncontended atomics)
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_tn5_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_tn5_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    a.nsplit = splits;
    a.ntile1 = a.ntile_all = ntile;  // single problem
    a.A2 = a.B2 = nullptr; a.dW2 = a.dbias2 = nullptr; a.lda2 = a.ldb2 = a.ldw2 = a.N2 = 0;
    if (a.ws && (a.ldw % 4 || a.K % 4 || g_ocn_tuning[11] > 1 || g_ocn_tuning[15] > 0)) return 1;  // (the scratch was sized for the default split)
    tn5_launch(a, st);
    if (a.ws) {
        const long total4 = (long)a.N * a.K / 4;
        int grid = (int)((total4 + 255) / 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(tn5_reduce_kernel, dim3(grid), dim3(256), 0, st, a.ws, a.dW, a.N, a.K, a.ldw, splits, a.alpha, a.dbias, splits * a.tiles_k);
    }
    if (hipGetLastError() != hipSuccess) return OCN_ERR_LAUNCH;
    return OCN_OK;
}

int ocn_launch_tn5_pair(GemmTnArgs a, hipStream_t st) {
    if (a.N % 8 || a.N2 % 8 || a.K % 8 || a.lda % 8 || a.ldb % 8 || a.lda2 % 8 || a.ldb2 % 8) return 1;
    if ((a.dbias == nullptr) != (a.dbias2 == nullptr)) return 1;  // one kernel instantiation serves both problems
    if (g_tn5_num_cu == 0) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_tn5_num_cu = n;
    }
    a.ablate = g_ocn_tuning[4];
    a.ws = nullptr;
    a.tiles_k = ocn_cdiv(a.K, 256);
    a.tiles_n = ocn_cdiv(a.N, 256);  // (problem 1; the kernel derives a tile's row from tiles_k only)
    a.ntile1 = a.tiles_n * a.tiles_k;
    a.ntile_all = a.ntile1 + ocn_cdiv(a.N2, 256) * a.tiles_k;
    const int msteps = ocn_cdiv(a.M, 32);
    int splits = g_tn5_num_cu / a.ntile_all;
    if (splits < 1) splits = 1;
    if (splits > msteps) splits = msteps;
    a.chunk = ocn_cdiv(msteps, splits) * 32;
    splits = ocn_cdiv(a.M, a.chunk);
    a.nwg = splits * a.ntile_all;
    a.nsplit = splits;
    long ldmax = a.lda > a.ldb ? a.lda : a.ldb;
    if (a.lda2 > ldmax) ldmax = a.lda2;
    if (a.ldb2 > ldmax) ldmax = a.ldb2;
    if ((long)(a.chunk + 128) * ldmax * 2 >= 0x7fffffffL) return 1;  // 32-bit buffer offsets (incl. the run-ahead past m_end)
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_tn5_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_tn5_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    tn5_launch(a, st);
    if (hipGetLastError() != hipSuccess) return OCN_ERR_LAUNCH;
    return OCN_OK;
}
