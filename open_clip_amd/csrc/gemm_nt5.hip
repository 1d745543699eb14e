// Persistent NT GEMM for the big tower GEMMs on gfx950:  C[M,N] = A[M,K] . B[N,K]^T  (+ fused epilogue).
//
// One workgroup per CU (8 waves, 160 KiB LDS) walks its share of the 256x256 output tiles; the operand stream never
// drains between tiles.  What the measurements on the MI355X said (tools/probes/dma_probe.hip):
//   * the L2 -> LDS feed is the binding resource of a 256x256 bf16 tile (52-68 GB/s per CU achievable, 62-75 needed
//     at the MFMA peak); LDS-DMA instructions that fetch FULL 128-byte lines (8 rows x 128 B) move ~25 % more than
//     ones that fetch half lines (16 rows x 64 B)  ->  K is consumed in 64-wide tiles (128-byte LDS rows);
//   * a non-persistent workgroup pays ~8 us per tile for launch + first-load latency + a serialized epilogue
//     ->  the DMA ring runs 1.5 K-tiles ahead ACROSS tile boundaries and the epilogue's stores are fire-and-forget.
//
// K-tile = 64 k.  Operand units of 16 KiB (128 rows x 128 B): A0/A1 = the rows of the two 64-row halves of every wave's
// 128-row strip, B0/B1 = the two 32-column halves of every wave's 64-column strip.  LDS ring = 2 K-tiles x 4 units.
// A K-tile is two phases of 16 MFMAs per wave: even = A0 x (B0,B1), odd = A1 x (B0,B1) with the B fragments kept in
// registers.  One raw s_barrier + one counted vmcnt per phase:
//   even phase of K-tile g: wait vmcnt(6) -> barrier -> issue A1(g+1)              [2 DMA per wave]
//   odd  phase of K-tile g: wait vmcnt(2) -> barrier -> issue A0,B0,B1(g+2)        [6 DMA per wave]
// so every unit is issued >= 2 phases before the barrier that publishes it and into a slot whose last reader retired
// (lgkmcnt(0)) before the barrier preceding the issue.  Fragment reads are inline asm (hipcc would wait lgkmcnt(0)
// right after issuing loop-carried reads); each read group has 4 MFMAs of cover.
// The MFMAs compute C^T tiles (B fragment first), so a lane owns ONE row of C and 4 consecutive columns per
// register quad: the epilogue converts in registers, transposes through a private 4 KiB LDS staging buffer with
// ds_write_b64/b128, and stores whole 128-byte rows.
#include "gemm_args.h"

#include <type_traits>

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

// OCN_DEV_BUILD (open_clip_amd/build.py --dev -> libopenclip_hip_dev.so): the developer knobs of include/openclip_hip_debug.h that reach INTO
// the kernels -- ablation bits that drop stores / loads / arithmetic (results wrong), the per-tile timeline build, cache-policy flips -- exist only
// in that build.  In the product library ABL() folds to 0 and none of it is compiled.
#ifdef OCN_DEV_BUILD
#define ABL(ARGS, BITS) ((ARGS).ablate & (BITS))
#else
#define ABL(ARGS, BITS) 0
#endif

namespace {

#ifdef OCN_DEV_BUILD
__device__ long long g_nt5_trace[1024];  // developer timeline (DBG kernels only): 2 workgroups x 8 tiles x 8 stamps
#endif

typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
constexpr int UNIT = 16384;
constexpr int RING_BYTES = 8 * UNIT;
constexpr int STG_BYTES = 4096;
constexpr int LDS_BYTES = RING_BYTES + 8 * STG_BYTES;  // 163840 = all of a CU's LDS

#define A_SLOT(HA, P) (((HA)*2 + (P)) * 16384)
#define B_SLOT(HB, P) (65536 + ((HB)*2 + (P)) * 16384)  // LDS byte offset (DMA destination)
#define B_RD(HB, P) (((HB)*2 + (P)) * 16384)            // ds_read immediate (the 64 KiB base is in the address VGPR)

#define DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF))
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define SB() __builtin_amdgcn_sched_barrier(0)

// Epilogue staging goes through inline-asm LDS ops: for compiler-visible LDS reads hipcc inserts s_waitcnt vmcnt(0)
// (it cannot prove they do not alias an in-flight LDS-DMA), which would serialize every global store of the epilogue
// behind a full memory round trip (measured: 10 us per tile instead of ~1).
OCN_DEV void lds_w32(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
OCN_DEV void lds_r64x4(unsigned a0, u32x2_t& d0, u32x2_t& d1, u32x2_t& d2, u32x2_t& d3) {  // rows it*8 + (lane>>3) of the padded u8 staging image
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:576\n\tds_read_b64 %2, %4 offset:1152\n\tds_read_b64 %3, %4 offset:1728\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
                 : "v"(a0)
                 : "memory");
}
OCN_DEV void lds_w64(unsigned addr, bf16x4 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
OCN_DEV void lds_w128(unsigned addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <typename T>
OCN_DEV void lds_r128x4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, T& d0, T& d1, T& d2, T& d3) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                 : "memory");
}

// Global accesses of the epilogue are BUFFER instructions on descriptors that start at the tile's first row: rows
// beyond M and (by forcing the offset out of range) columns beyond N are dropped / read as zero by the bounds check, so
// the whole epilogue is one basic block and hipcc can keep counted vmcnt waits (with exec-masked branches around the
// stores it falls back to vmcnt(0) before every store).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
constexpr unsigned OOB = 0x80000000u;

OCN_DEV __amdgpu_buffer_rsrc_t tile_rsrc(const void* base, long row0, int rows_total, int ld, int esz) {
    long bytes = (long)(rows_total - row0) * ld * esz;
    if (bytes > 0x7fffffffL) bytes = 0x7fffffffL;
    if (bytes < 0) bytes = 0;
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + row0 * ld * esz), 0, (int)bytes, 0x00020000);
}

// What the main loop fetches ahead for the epilogue: PFN loads of 16 bytes per lane, issued three phases (~2 us) before the epilogue starts,
// behind every DMA the remaining phases wait for (vmcnt retires in order), so the main loop never waits for them:
//   dGELU     the saved 8-bit gelu' of the wave's WHOLE 128 x 64 sub-tile (8 KiB = 32 VGPRs): load j = rows j*16 + (lane >> 2), bytes
//             (lane & 3)*16 .. +15 of the wave's 64 -- one 64-byte segment per row and instruction.  (Round 3 fetched them per 32 x 32 block,
//             4 bytes per lane = 32-byte row segments, from inside the epilogue: the PMC pass showed the saved derivatives crossing the
//             L2 -> fabric boundary 1.85 times, and the first wait of every tile sat on a full HBM round trip.)
//   residual  the fp32 residual rows of the first two 32 x 32 blocks = slots 0 and 1 of the epilogue's operand ring
//   bf16 residual (round 6)  the bf16 residual rows of the first two 32 x 64 sub-blocks of the wave's strip, 16 bytes (8 columns) per lane in the
//             layout of the epilogue's row-wise stores; the other two are requested by the epilogue as it frees these registers
template <int EPI>
struct NtPf {
    static constexpr int N = (EPI == OCN_EPI_DGELU || EPI == OCN_EPI_BIAS_RESID_F32 || EPI == OCN_EPI_BIAS_RESID_BF16) ? 8 : 0;
};

// byte offset of this lane's 4 fp32 columns in row it*8 + (lane >> 3) of 32 x 32 block blk (fp32-staged epilogues, residual prefetch)
OCN_DEV unsigned f32blk_off(const GemmNtArgs& a, int lane_o, int row_w, int gn_w, int blk, int it, unsigned esz) {
    const int ha = blk >> 2, s = (blk >> 1) & 1, hb = blk & 1;
    const int gn = gn_w + hb * 32 + (lane_o & 7) * 4;
    const int row = row_w + ha * 64 + s * 32 + it * 8 + (lane_o >> 3);
    return ((unsigned)(row * a.ldc + gn) * esz) | (gn < a.N ? 0u : OOB);  // OOB: past the descriptor's bound (no select, no branch)
}

template <int EPI, int AUX>
OCN_DEV void epi_prefetch(const GemmNtArgs& a, int m0, int n0, int row_w, int wn, int lane, u32x4 (&pf)[8]) {  // row_w: the wave's first row inside the tile
    if constexpr (NtPf<EPI>::N > 0) {
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));  // see epilogue5: keeps the address arithmetic out of the tile loop's live set
        const int m_ld = ABL(a, 128) ? 0 : a.M;
        const int gn_w = n0 + wn * 64;
        if constexpr (EPI == OCN_EPI_DGELU) {
            const __amdgpu_buffer_rsrc_t r_aux = tile_rsrc(a.aux, m0, m_ld, a.ldc, 1);
            const int gn = gn_w + (lane_o & 3) * 16;
            const unsigned base = (unsigned)((row_w + (lane_o >> 2)) * a.ldc + gn) | (gn < a.N ? 0u : OOB);
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = __builtin_amdgcn_raw_buffer_load_b128(r_aux, base + (unsigned)(j * 16 * a.ldc), 0, (AUX & 8) ? 2 : 0);
        } else if constexpr (EPI == OCN_EPI_BIAS_RESID_BF16) {
            const __amdgpu_buffer_rsrc_t r_res = tile_rsrc(a.resid, m0, m_ld, a.ldc, 2);
            const int gn = gn_w + (lane_o & 7) * 8;
            const unsigned col_off = gn < a.N ? (unsigned)gn * 2u : OOB;
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    pf[sb * 4 + it] = __builtin_amdgcn_raw_buffer_load_b128(r_res, (unsigned)((row_w + sb * 32 + it * 8 + (lane_o >> 3)) * a.ldc) * 2u + col_off, 0, (AUX & 8) ? 2 : 0);
        } else {
            const __amdgpu_buffer_rsrc_t r_res = tile_rsrc(a.resid, m0, m_ld, a.ldc, 4);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    pf[blk * 4 + it] = __builtin_amdgcn_raw_buffer_load_b128(r_res, f32blk_off(a, lane_o, row_w, gn_w, blk, it, 4u), 0, (AUX & 8) ? 2 : 0);
        }
    }
}

OCN_DEV void lds_w64x2(unsigned addr, u32x4 v) {  // 16 bytes as two 8-byte stores (the u8 image's rows are 8-, not 16-byte aligned)
    const u32x2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
    asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:8" ::"v"(addr), "v"(lo), "v"(hi) : "memory");
}
OCN_DEV void lds_r32x8(unsigned a0, unsigned (&q)[2][4]) {  // the lane's four 4-byte quads of both 32-column halves of a row of the u8 image
    asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:8\n\tds_read_b32 %2, %8 offset:16\n\tds_read_b32 %3, %8 offset:24\n\t"
                 "ds_read_b32 %4, %8 offset:32\n\tds_read_b32 %5, %8 offset:40\n\tds_read_b32 %6, %8 offset:48\n\tds_read_b32 %7, %8 offset:56\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(q[0][0]), "=&v"(q[0][1]), "=&v"(q[0][2]), "=&v"(q[0][3]), "=&v"(q[1][0]), "=&v"(q[1][1]), "=&v"(q[1][2]), "=&v"(q[1][3])
                 : "v"(a0)
                 : "memory");
}

template <int EPI, int AUX = 0>
OCN_DEV void epilogue5(const GemmNtArgs& a, f32x16 (&acc)[2][2][2], int m0, int n0, int wm, int wn, int lane, unsigned stg, u32x4 (&pf)[8],
                       int ha_n, int row_shift, long long* dbg = nullptr, int ks = 0) {
    // ks: split-K launches (a.ksplit > 1, fp32 output only): this tile holds the partial sum of K-slice ks and goes to slab ks of the workspace a.out
    // ha_n / row_shift: 2 / 0 for a whole 256 x 256 tile; 1 / 0 or 64 for a HALF tile of the launch's last round (see the kernel): only acc[0]
    // holds results, for rows wm*128 + row_shift + 0..63
    // Lane constants are laundered through an empty asm once per tile: otherwise hipcc hoists ~40 VGPRs of epilogue
    // addresses (per-row store offsets, swizzled staging addresses) out of the tile loop and keeps them live across the
    // main loop, which is already at the 256-register budget.
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int lr = lane_o & 31, lh = lane_o >> 5;
    const int gn_w = n0 + wn * 64;
    const int rd_row = lane_o >> 3, rd_chunk = lane_o & 7;
    const unsigned rd_addr = stg + rd_row * 128 + ((rd_chunk ^ rd_row) << 4);  // + it * 1024 for rows it*8 + rd_row
    constexpr bool IS_GELU = (EPI == OCN_EPI_BIAS_GELU || EPI == OCN_EPI_BIAS_QUICKGELU);  // two-output activation epilogues
    constexpr bool IS_DGELU = (EPI == OCN_EPI_DGELU);
    constexpr bool IS_RES16 = (EPI == OCN_EPI_BIAS_RESID_BF16);  // bf16 residual stream: out = bf16(resid + bf16(acc + bias)), added behind the transpose
    constexpr bool BF16_STAGED = (EPI == OCN_EPI_BF16 || IS_GELU || IS_DGELU || IS_RES16 || EPI == OCN_EPI_CE_ONEPASS || EPI == OCN_EPI_CE_ONEPASS_FULL);
    constexpr bool OUT_F32 = (EPI == OCN_EPI_BIAS_RESID_F32 || EPI == OCN_EPI_F32);
    // developer knobs 32 / 128 (OCN_DEV_BUILD only): zero-sized descriptors -- the epilogue's stores (32) / operand loads (128) are still issued
    // but the bounds check drops them before they reach memory: what the tile loop costs with a free memory system (results wrong)
    const int m_st = ABL(a, 32) ? 0 : a.M, m_ld = ABL(a, 128) ? 0 : a.M;
    const bool aux_is_out = IS_GELU;
    // developer knob 0x80000 (OCN_DEV_BUILD only): every store of this workgroup lands in ONE 64 KiB window (per workgroup, at the head of the
    // output) that stays resident in L2 -- the stores are issued and acknowledged as usual but never have to wait for HBM: what the tile loop
    // would cost if no operand wait ever sat behind a store acknowledgement (profiles/r02_nt6_trickled_epilogue_experiment.txt, finding (a))
    const bool win = ABL(a, 0x80000) != 0;
    const unsigned omask = win ? 0xfff0u : ~0u;
    const long m_win = win ? (long)((blockIdx.x * 65536L) / ((long)a.ldc * (OUT_F32 ? 4 : 2))) : m0;
    const __amdgpu_buffer_rsrc_t r_out = tile_rsrc((const char*)a.out + (size_t)ks * a.ws_stride * 4, m_win, m_st, a.ldc, OUT_F32 ? 4 : 2);
    const __amdgpu_buffer_rsrc_t r_aux = aux_is_out ? tile_rsrc(a.aux, m_win, m_st, a.ldc, 1) : tile_rsrc(a.aux, m0, m_ld, a.ldc, 1);  // gelu' in 8 bits
    const __amdgpu_buffer_rsrc_t r_res = tile_rsrc(a.resid, m0, m_ld, a.ldc, IS_RES16 ? 2 : 4);
    const __amdgpu_buffer_rsrc_t r_bias = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? a.N * 4 : 0, 0x00020000);
    // Bias is fetched ONCE, before any store of this tile is issued (a later load would have to wait behind the stores):
    // bf16-staged epilogues add it in the accumulator layout (before rounding), fp32-staged ones after the transpose.
    // (The dGELU epilogue has none: ocn_launch_nt5 hands a dGELU GEMM with a bias to the general kernel.)
    f32x4 bv[2][4], bq[2];
    if (BF16_STAGED && !IS_DGELU) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bv[hb][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_bias, (gn_w + hb * 32 + 8 * g + 4 * lh) * 4, 0, 0));
    } else if (!BF16_STAGED) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
            bq[hb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_bias, (gn_w + hb * 32 + rd_chunk * 4) * 4, 0, 0));
    }
    // ---- fused cross-entropy epilogue (ocn_fused_logits_ce): the fp32 logits tile is consumed in the accumulator layout -- a lane
    // owns ONE row per (ha, s) and 32 of the wave's 64 columns (hb x 16 registers); its other half sits in lane ^ 32.
    // ---- one-pass form (round 6): e_ij = exp(l_ij - c_i) with a PER-ROW SHIFT c_i known before the GEMM (loss.hip::ce_prep_rows_kernel: the row's label
    // logit, clamped from below so that no entry can overflow) is written as the bf16 matrix G' and its row sums S_i = sum_j e_ij, SL_i = sum_j e_ij l_ij
    // as per-strip partials; softmax = G' / S_i is never formed: the 1 / S_i goes into the consumers' row scales (loss.py::_PairTerm.dX / dY).  The
    // logits GEMM runs ONCE instead of twice (statistics pass + gradient pass): 3 GEMMs per direction of the loss instead of 4.
    if constexpr (EPI == OCN_EPI_CE_ONEPASS || EPI == OCN_EPI_CE_ONEPASS_FULL) {
        // _FULL: M and N are multiples of 256 (every tile is whole: chosen by the HOST, a separate instantiation): no row / column masks.
        // (measured and not adopted, round 6: an unmasked copy of this loop for interior tiles behind a uniform branch INSIDE one kernel -- the second copy
        // of the epilogue costs more registers than the two compares and selects per element it saves: 187 -> 333 us on [4096 x 32768 x 512])
        constexpr bool MASKED = (EPI == OCN_EPI_CE_ONEPASS);
#pragma unroll
        for (int ha = 0; ha < 2; ++ha)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int row = m0 + wm * 128 + ha * 64 + s * 32 + lr;
                const bool rv = !MASKED || row < a.M;
                const float c2 = rv ? a.ce_shift2[row] : 0.f;  // c_i * log2(e)
                float se = 0.f, sel = 0.f;
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int col = gn_w + hb * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
                        const float v = acc[ha][s][hb][r];
                        float e = __builtin_amdgcn_exp2f(fmaf(v, 1.4426950408889634f, -c2));
                        if constexpr (MASKED) e = (rv && col < a.N) ? e : 0.f;
                        se += e;
                        sel = fmaf(e, v, sel);
                        acc[ha][s][hb][r] = e;  // stored as bf16 by the staged path below
                    }
                se += __shfl_xor(se, 32, 64);
                sel += __shfl_xor(sel, 32, 64);
                if (rv && lh == 0) {
                    float* st = a.ce_stats + ((size_t)row * a.ce_parts + (size_t)(n0 >> 8) * 4 + wn) * 2;
                    st[0] = se;
                    st[1] = sel;
                }
            }
    }
    // Every DMA issued so far must have landed: the first phases of the next tile then need no vmcnt wait and the
    // stores below drain under them.  (hipcc does not know about the asm LDS-DMAs; its own loads / stores below get
    // ordinary counted waits.)  This is also where the operands the main loop fetched ahead (pf) are waited for.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (dbg) dbg[5] = wall_clock64();
    const int row_w = wm * 128 + row_shift;  // this wave's first row inside the tile
    if (BF16_STAGED) {
        const int gn = gn_w + rd_chunk * 8;
        const unsigned col_off = gn < a.N ? (unsigned)gn * 2u : OOB;
        const unsigned col_off_aux = gn < a.N ? (unsigned)gn : OOB;  // the saved derivative: one byte per element
        // u8 staging image of a 32 x 64 sub-block: rows of 64 bytes padded to 72.  GELU: a lane writes the 4 bytes of its register quad and reads 8 bytes
        // of a row; dGELU the other way round: a lane writes the 16 bytes it loaded and reads its four quads of both 32-column halves
        const unsigned aw_addr = stg + lr * 72 + lh * 4, ar_addr = stg + rd_row * 72 + rd_chunk * 8;
        const unsigned dw_addr = stg + (lane_o >> 2) * 72 + (lane_o & 3) * 16;
#pragma unroll
        for (int ha = 0; ha < 2; ++ha) {
            if (ha >= ha_n) break;  // uniform
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // GELU: gelu(v) AND gelu'(v) come out of one evaluation of the shared erf / exp parts; the derivative is what
                // is saved for the backward (aux, 8-bit fixed point), whose epilogue is then a plain multiply (dGELU)
                bf16x4 pk[2][4];
                unsigned dq[2][4];
                if constexpr (IS_DGELU) {
                    lds_w64x2(dw_addr, pf[(ha * 2 + s) * 2]);
                    lds_w64x2(dw_addr + 16 * 72, pf[(ha * 2 + s) * 2 + 1]);
                    lds_r32x8(aw_addr, dq);
                }
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v = {acc[ha][s][hb][4 * g], acc[ha][s][hb][4 * g + 1], acc[ha][s][hb][4 * g + 2], acc[ha][s][hb][4 * g + 3]};
                        if (EPI == OCN_EPI_BF16) v = v * a.alpha + bv[hb][g]; else if (IS_GELU || IS_RES16) v = v + bv[hb][g];
                        if (IS_DGELU) v = v * dgelu_unpack4(dq[hb][g]);  // gelu'(pre-activation), saved by the forward epilogue
                        if (IS_GELU) {
                            f32x4 gv = v, dv = v;  // (developer knob 1: skip the VALU work)
                            if constexpr (EPI == OCN_EPI_BIAS_GELU) {
#ifdef OCN_DEV_BUILD
                                if (ABL(a, 0x400000)) {  // developer knob: the Abramowitz-Stegun form that shipped until round 4 (A/B)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        float g1, d1;
                                        gelu_both_as(v[e], g1, d1);
                                        gv[e] = g1;
                                        dv[e] = d1;
                                    }
                                } else
#endif
                                if (!ABL(a, 1)) gelu_both_poly4(v, gv, dv);  // polynomial-CDF form on the register quad (ocn_common.h)
                            } else if (!ABL(a, 1)) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float g1, d1;
                                    quickgelu_both(v[e], g1, d1);
                                    gv[e] = g1;
                                    dv[e] = d1;
                                }
                            }
                            dq[hb][g] = dgelu_pack4(dv[0], dv[1], dv[2], dv[3]);
                            v = gv;
                        }
                        pk[hb][g] = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                    }
                if constexpr (IS_GELU) {
                    if (dbg && ha == 1 && s == 0) dbg[6] = wall_clock64();
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) lds_w32(aw_addr + (hb * 8 + g * 2) * 4, dq[hb][g]);
                    u32x2_t d8[4];
                    lds_r64x4(ar_addr, d8[0], d8[1], d8[2], d8[3]);
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = row_w + ha * 64 + s * 32 + it * 8 + rd_row;
                        const unsigned off = (unsigned)(row * a.ldc) + col_off_aux;
                        __builtin_amdgcn_raw_buffer_store_b64(d8[it], r_aux, off & omask, 0, AUX & 2);
                    }
                }
                {
                    if (dbg && ha == 1 && s == 0 && !IS_GELU) dbg[6] = wall_clock64();
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) lds_w64(stg + lr * 128 + (((hb * 4 + g) ^ (lr & 7)) << 4) + lh * 8, pk[hb][g]);
                    bf16x8 d[4];
                    lds_r128x4(rd_addr, rd_addr + 1024, rd_addr + 2048, rd_addr + 3072, d[0], d[1], d[2], d[3]);
                    if constexpr (IS_RES16) {
                        // the residual rows arrive in the layout of the stores: sub-blocks 0 / 1 with the main loop's prefetch, 2 / 3 requested here, as soon as
                        // the registers of sub-block sb are free and BEFORE its stores are issued (vmcnt retires in issue order: the wait for sub-block sb + 2's
                        // rows then covers the stores of sub-block sb - 1 at most, never the ones just issued)
                        const int sb = ha * 2 + s;
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const bf16x8 rr = __builtin_bit_cast(bf16x8, pf[(sb & 1) * 4 + it]);
                            bf16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(d[it][e]) + bf2f(rr[e]));
                            d[it] = o;
                        }
                        if (sb + 2 < 2 * ha_n) {
#pragma unroll
                            for (int it = 0; it < 4; ++it) {
                                const int row = row_w + (ha + 1) * 64 + s * 32 + it * 8 + rd_row;
                                pf[(sb & 1) * 4 + it] = __builtin_amdgcn_raw_buffer_load_b128(r_res, (unsigned)(row * a.ldc) * 2u + col_off, 0, (AUX & 8) ? 2 : 0);
                            }
                        }
                    }
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = row_w + ha * 64 + s * 32 + it * 8 + rd_row;
                        const unsigned off = (unsigned)(row * a.ldc) * 2u + col_off;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d[it]), r_out, off & omask, 0, AUX & 2);
                    }
                }
            }
        }
    } else {
        // 32x32 fp32 blocks (128-byte rows).  The residual rows of block k + DEPTH - 1 are requested before block k's stores are issued: vmcnt
        // retires loads AND stores in issue order, so waiting for a block's operand also waits for every store issued before its request -- with a
        // ring of four that is never the store of the block just before.  Slots 0 and 1 arrive with the main loop's prefetch (pf).
        constexpr bool HAS_EX = (EPI == OCN_EPI_BIAS_RESID_F32);
        constexpr int DEPTH = HAS_EX ? 4 : 1;
        f32x4 ex[DEPTH][4];
        auto load_ex = [&](int blk, f32x4 (&e)[4]) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                e[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, f32blk_off(a, lane_o, row_w, gn_w, blk, it, 4u), 0, (AUX & 8) ? 2 : 0));
        };
        if constexpr (HAS_EX) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int it = 0; it < 4; ++it) ex[b][it] = __builtin_bit_cast(f32x4, pf[b * 4 + it]);
            load_ex(2, ex[2]);
        }
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
            if (blk >= 4 * ha_n) break;  // uniform
            const int ha = blk >> 2, s = (blk >> 1) & 1, hb = blk & 1;
            if (dbg && blk == 4) dbg[6] = wall_clock64();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[ha][s][hb][4 * g], acc[ha][s][hb][4 * g + 1], acc[ha][s][hb][4 * g + 2], acc[ha][s][hb][4 * g + 3]};
                lds_w128(stg + lr * 128 + (((g * 2 + lh) ^ (lr & 7)) << 4), v);
            }
            f32x4 d[4];
            lds_r128x4(rd_addr, rd_addr + 1024, rd_addr + 2048, rd_addr + 3072, d[0], d[1], d[2], d[3]);
            if constexpr (HAS_EX) {
                if (blk + DEPTH - 1 < 4 * ha_n) load_ex(blk + DEPTH - 1, ex[(blk + DEPTH - 1) % DEPTH]);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const f32x4 v = (EPI == OCN_EPI_F32) ? d[it] * a.alpha + bq[hb] : d[it] + bq[hb];
                const unsigned off = f32blk_off(a, lane_o, row_w, gn_w, blk, it, 4u) & omask;
                if constexpr (EPI == OCN_EPI_BIAS_RESID_F32) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v + ex[blk % DEPTH][it]), r_out, off, 0, AUX & 2);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out, off, 0, AUX & 2);
                }
            }
        }
    }
}

// DBG = developer build of the same kernel that logs a per-tile timeline into a.aux (tools/gemm_trace.py; plain bf16
// epilogue only); the production instantiation folds it away.
// AUX = cache policy of the epilogue: bit 1 (2) = non-temporal stores, bit 3 (8) = non-temporal loads of the residual / saved operand
// RESCUE (ocn_set_tile_rescue, the multi-GPU form): the static shares stay, but a share whose workgroup has NOT STARTED by the time others finish -- its
// CU is held by another stream's kernel, e.g. a collective -- is handed out entry by entry to the finishers.  The board a.rescue holds one counter per
// workgroup: the owner adds RESCUE_BIG ONCE when it starts (what it gets back = the entries finishers already took: it begins behind them, and no
// finisher can claim from it afterwards), a finisher claims entry k of share w with one atomicAdd(+1) (valid while k < the share's length).  No atomic
// sits inside the tile loop: the DMA pipeline's counted vmcnt waits retire in issue order and would stall behind one.
constexpr int RESCUE_BIG = 1 << 20;
template <int EPI, bool DBG, int AUX = 0, bool RESCUE = false>
__global__ __launch_bounds__(512, 2) void gemm_nt5_kernel(GemmNtArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int lr = lane & 31, lh = lane >> 5;
    const int kc = a.K / a.ksplit;  // K of one tile: all of it, or one slice of a split-K launch (launch5: a multiple of 128)
    const int nk = kc >> 6;   // K-tiles per output tile (even)
    const int G = gridDim.x;
    // a.tail_n > 0: the launch's last, partial round of tiles is split into HALF tiles (see below); the whole tiles are then exactly a.tail_first / G per
    // workgroup
    auto share_of = [&](int w) { return a.tail_n > 0 ? a.tail_first / G : (a.ntiles - w + G - 1) / G; };  // whole tiles of workgroup w's share
    int my_tiles = share_of((int)blockIdx.x);
    int vblk = (int)blockIdx.x, ibase = 0;  // the share this pass works on and its first entry (RESCUE: a finisher's later passes take ONE entry of another share)
    const unsigned lds_base = (unsigned)(size_t)(OCN_LDS char*)smem;
    int* const box = (int*)(smem + RING_BYTES);  // RESCUE: workgroup-wide mailbox (wave 0's staging area, free outside a tile's epilogue)
    // The owner's claim is ISSUED here and read behind the first prologue, whose operand fetches cover its latency (a device-scope atomic's round trip is
    // about what the prologue takes: waited for up front it costs 0.35 % of a training step, 0.5 ms over ~400 launches).  The prologue assumes nothing was
    // taken; when something was (the workgroup got its CU late), it is repeated behind the entries the finishers hold.
    int claim = 0;
    bool fresh = RESCUE;
    if constexpr (RESCUE) if (threadIdx.x == 0) {
        // inline asm: hipcc's own atomicAdd is consumed on the spot (its wave-reduction wrapper reads the result back at once)
        const int* cptr = a.rescue + blockIdx.x * OCN_RESCUE_STRIDE;
        asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "=v"(claim) : "v"(0), "v"(RESCUE_BIG), "s"(cptr) : "memory");
    }

    // fragment read addresses: unit row (wave strip + lane&31), 16-byte chunk ((ks*2 + lh) ^ swz) = (q<<4) ^ (ks<<5)
    unsigned va[4], vb[4];
    {
        const int q = lh ^ swz_nt(lr);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned o = (unsigned)((q << 4) ^ (ks << 5));
            va[ks] = lds_base + (unsigned)((wm * 64 + lr) * 128) + o;
            vb[ks] = lds_base + 65536u + (unsigned)((wn * 32 + lr) * 128) + o;
        }
    }

    // ---- DMA cursors ---------------------------------------------------------------------------------------------
    // Operands are fetched with "buffer_load_dwordx4 ... lds" in INLINE ASM: (a) buffer addressing keeps the whole
    // cursor in SGPRs (descriptor based at the tile's first row, K offset and row-block offset in soffset; rows beyond
    // M / N read as zero) and leaves 4 lane-constant VGPR offsets instead of 8 advancing 64-bit pointers; (b) hipcc
    // never sees an LDS-DMA, so the loads/stores of the epilogue get counted vmcnt waits (with a visible
    // global_load_lds anywhere in the function every later wait degenerates to vmcnt(0)).
    unsigned voA[2], voB[2];  // per-lane byte offsets of this wave's two pieces of an A / B unit (swizzled chunk included)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int u = (wave * 2 + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz_nt(u);
        voA[j] = (unsigned)(((u >> 6) * 128 + (u & 63)) * a.lda * 2 + c * 16);
        voB[j] = (unsigned)(((u >> 5) * 64 + (u & 31)) * a.ldb * 2 + c * 16);
    }
    const unsigned a_half = (unsigned)(64 * a.lda * 2), b_half = (unsigned)(32 * a.ldb * 2);  // A1 - A0, B1 - B0 (bytes)
    // Tile walk: the linear order is BAND-major -- column bands of a.band n-tiles, row-major inside a band -- and every
    // XCD owns a contiguous piece of it (xcd_remap), so the 32 tiles an XCD works on at any time are 32/band rows x band
    // columns and the band's B panels (band x 256 x K bf16) stay resident in that XCD's 4 MiB L2 while the A panels stream
    // through once per band.  Row-major over all of N (band = tiles_n) re-reads the whole of B once per round of tiles when
    // B does not fit in L2 next to the A panels: measured 1.74 GB of L2-miss reads per launch instead of 0.32 GB on the
    // [204800 x 3072 x 768] GEMM (profiles/r01_pmc_hbm_traffic_before_band.txt).
    // split-K (a.ksplit > 1): the walk has ksplit x as many entries; entry t = K-slice t / tiles_mn of output tile t % tiles_mn
    const int tiles_mn = a.ntiles / a.ksplit;
    const int tiles_m = tiles_mn / a.tiles_n;
    const int band_tiles = tiles_m * a.band;
    auto tile_origin = [&](int i, int& m0, int& n0, int& ks) {
        int tile = xcd_remap(vblk + (i + ibase) * G, a.ntiles);
        ks = tile / tiles_mn;
        tile -= ks * tiles_mn;
        const int cb = tile / band_tiles, r = tile - cb * band_tiles;
        const int width = min(a.band, a.tiles_n - cb * a.band);
        const int mi = r / width;
        m0 = mi * 256;
        n0 = (cb * a.band + (r - mi * width)) * 256;
    };
    auto make_desc = [&](const bf16* base, int row0, int rows, int ld, int kofs = 0) -> u32x4 {  // kofs: first column of the K-slice
        long bytes = (long)(rows - row0) * ld * 2 - (long)kofs * 2;
        bytes = bytes > 0x7fffffffL ? 0x7fffffffL : bytes;
        const unsigned long long p = (unsigned long long)(base + (size_t)row0 * ld + kofs);
        u32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
        r[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu);
        r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);
        r[3] = 0x00020000u;
        return r;
    };
    u32x4 dA1, dA0, dB;                            // descriptors of the tiles the two cursors are in
    int ab_i = 0, a1_i = 0;                        // tile ordinal of each cursor
    unsigned ab_k = 0, a1_k = 0;                   // byte offset of the K-tile each cursor issues next (kt * 128)
    const unsigned k_end = (unsigned)nk * 128u;
    auto set_a1 = [&](int i) {
        int m0, n0, ks;
        tile_origin(i, m0, n0, ks);
        dA1 = make_desc(a.A, m0, a.M, a.lda, ks * kc);
    };
    auto set_ab = [&](int i) {
        int m0, n0, ks;
        tile_origin(i, m0, n0, ks);
        dA0 = make_desc(a.A, m0, a.M, a.lda, ks * kc);
        dB = make_desc(a.B, n0, a.N, a.ldb, ks * kc);
    };
    const unsigned wave_dst = lds_base + wave * 2048;  // this wave's 2 x 1 KiB pieces inside a unit
    // one LDS-DMA piece: 64 lanes x 16 B from desc[voff + soff] to LDS m0 + lane*16 (m0 is compiler-reserved: save / restore)
#define DMA(DESC, VOFF, SOFF, SLOT, J)                                                                         \
    {                                                                                                          \
        unsigned keep_;                                                                                        \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_)                                                                            \
                     : "v"(VOFF[J]), "s"(DESC), "s"(wave_dst + (SLOT) + (J)*1024), "s"(SOFF)                    \
                     : "memory");                                                                              \
    }
    auto adv_a1 = [&]() {
        a1_k += 128;
        if (a1_k == k_end) {
            a1_k = 0;
            if (a1_i + 1 < my_tiles) set_a1(++a1_i);  // else: keep re-fetching the last tile (valid addresses, results unused)
        }
    };
    auto adv_ab = [&]() {
        ab_k += 128;
        if (ab_k == k_end) {
            ab_k = 0;
            if (ab_i + 1 < my_tiles) set_ab(++ab_i);
        }
    };
    // the A panel is used by the `band` workgroups of one tile row and then dead, B panels are re-used round after round:
    // (AUX & 16) fetches A with the non-temporal policy so that it is the A lines the L2 gives up first (developer experiment)
#define DMA_NT(DESC, VOFF, SOFF, SLOT, J)                                                                      \
    {                                                                                                          \
        unsigned keep_;                                                                                        \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen nt lds\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_)                                                                            \
                     : "v"(VOFF[J]), "s"(DESC), "s"(wave_dst + (SLOT) + (J)*1024), "s"(SOFF)                    \
                     : "memory");                                                                              \
    }
#define DMA_A1(J, P) { if constexpr ((AUX & 16) != 0) DMA_NT(dA1, voA, a1_k + a_half, A_SLOT(1, P), J) else DMA(dA1, voA, a1_k + a_half, A_SLOT(1, P), J) }
#define DMA_A0(J, P) { if constexpr ((AUX & 16) != 0) DMA_NT(dA0, voA, ab_k, A_SLOT(0, P), J) else DMA(dA0, voA, ab_k, A_SLOT(0, P), J) }
#define DMA_B0(J, P) DMA(dB, voB, ab_k, B_SLOT(0, P), J)
#define DMA_B1(J, P) DMA(dB, voB, ab_k + b_half, B_SLOT(1, P), J)

    // ---- phase stagger (a.stagger > 0: developer build only since round 5 -- see nt5_stagger: it stopped paying) ---------------
    // Persistent workgroups launched together walk tiles of equal cost in lockstep: every CU is in its main loop (HBM
    // idle) and then every CU is in its epilogue at once -- a chip-wide burst of 256 x 128..256 KiB of stores (+ residual /
    // pre-activation reads) that exceeds what the L2s can buffer, so the epilogue runs at HBM write speed while the MFMA
    // pipes idle (measured with the DBG timeline: 11 us per tile for the two-output GELU epilogue against 3 us for one
    // bf16 output; profiles/r01_nt5_tile_timeline.txt).  Workgroups therefore start in 4 phase classes (inside every XCD),
    // a quarter of a tile time apart, which spreads the epilogue traffic over the whole tile period.
    if (a.stagger > 0 && (int)blockIdx.x < a.first_wave) {  // later workgroups start whenever a CU frees up: already spread out
        int phase = ((int)blockIdx.x >> 3) & 3;
#ifdef OCN_DEV_BUILD
        // developer knob (ocn_set_tuning key 3): WHICH workgroups share a phase class.  0 (shipped): consecutive workgroups of an XCD take classes
        // 0,1,2,3,0,.. -- the `band` workgroups that stream the same A panel sit in four different classes, up to 3/4 of a tile apart in k;
        // 1: every workgroup of an XCD in one class (XCD = blockIdx % 8: the panel sharers run in step, the bursts of two XCDs coincide);
        // 2: the workgroups of one tile ROW of the band in one class (A-panel sharers in step, rows spread over the four classes)
        if (a.stagger_mode == 1) phase = (int)blockIdx.x & 3;
        else if (a.stagger_mode == 2) {
            int m0s, n0s, kss;
            tile_origin(0, m0s, n0s, kss);
            phase = (m0s >> 8) & 3;
        }
#endif
        if (phase) {
            const long long until = wall_clock64() + (long long)phase * a.stagger;
            while (wall_clock64() < until) __builtin_amdgcn_s_sleep(16);
        }
    }

    constexpr int PFN = NtPf<EPI>::N;
    u32x4 pf[8];          // epilogue operands fetched ahead by the main loop (epi_prefetch; unused when PFN == 0)
    f32x16 acc[2][2][2];  // [A half][32-row sub-block][B half], each a 32x32 C^T tile
    bf16x8 fa[2][2];      // A fragments: [buffer][sub-block]
    bf16x8 fb[2][4];      // B fragments of the whole K-tile: [B half][k-substep]
    const unsigned stg = lds_base + RING_BYTES + wave * STG_BYTES;
    long long* dbg = nullptr;
#ifdef OCN_DEV_BUILD
    if (DBG && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 133)) dbg = g_nt5_trace + (blockIdx.x ? 512 : 0);
#endif

  for (int pass = 0;; ++pass) {  // one pass without RESCUE; with it: the own share, then one rescued entry per pass
   if (!RESCUE || my_tiles > 0) {
    // ---- prologue: K-tiles 0 and 1 (minus A1(1)) ----
    ab_i = a1_i = 0;
    ab_k = a1_k = 0;
    set_ab(0);
    set_a1(0);
    DMA_A0(0, 0) DMA_A0(1, 0) DMA_B0(0, 0) DMA_B0(1, 0) DMA_B1(0, 0) DMA_B1(1, 0)
    adv_ab();
    DMA_A1(0, 0) DMA_A1(1, 0)
    adv_a1();
    DMA_A0(0, 1) DMA_A0(1, 1) DMA_B0(0, 1) DMA_B0(1, 1) DMA_B1(0, 1) DMA_B1(1, 1)
    adv_ab();

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (RESCUE) if (fresh) {  // (the vmcnt(0) above has seen the claim return: the counter retires in issue order)
        fresh = false;
        asm volatile("" : "+v"(claim)::"memory");  // no copy of the claim is made before this point
        if (threadIdx.x == 0) box[0] = claim;
        __syncthreads();
        const int lo = min(__builtin_amdgcn_readfirstlane(box[0]), my_tiles);
        if (lo > 0) {  // finishers took the head of this share before the workgroup got a CU: start again behind them
            ibase = lo;
            my_tiles -= lo;
            __syncthreads();
            --pass;
            continue;
        }
    }
    __builtin_amdgcn_s_barrier();

// OCN_PRIO_MODE (compile-time experiment, profiles/r02_setprio_experiment.txt): 1 = s_setprio 1 around every MFMA group of the main loop,
// 2 = static priority for the second-dispatched half of the workgroup (waves 4..7), 3 = both
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
#define MM(HA, FA, KS)                                                        \
    acc[HA][0][0] = mfma32(fb[0][KS], FA[0], acc[HA][0][0]);                  \
    acc[HA][0][1] = mfma32(fb[1][KS], FA[0], acc[HA][0][1]);                  \
    acc[HA][1][0] = mfma32(fb[0][KS], FA[1], acc[HA][1][0]);                  \
    acc[HA][1][1] = mfma32(fb[1][KS], FA[1], acc[HA][1][1]);
#define RD_A(FA, KS, HA, P) DSR(FA[0], va[KS], A_SLOT(HA, P)); DSR(FA[1], va[KS], A_SLOT(HA, P) + 4096);
#define RD_B(KS, P) DSR(fb[0][KS], vb[KS], B_RD(0, P)); DSR(fb[1][KS], vb[KS], B_RD(1, P));

    // one K-tile held in ring parity P; SKIP: the epilogue (or prologue) before it already drained every DMA
    // (Measured, no effect: issuing the DMA pieces of the two waves that share a SIMD in different k-substeps.)
    // PFM (epilogues with a prefetch, PFN > 0): 1 = the tile's second-to-last K-tile: the PFN epilogue-operand loads are issued behind its two
    // A1 pieces; from there on every counted wait carries + PFN (the loads sit between the DMAs in the in-order vmcnt queue and are never
    // waited for); 2 = the tile's last K-tile: its odd phase needs nothing that is still in flight (its A1 unit was published by the even
    // phase) -- the wait that used to publish the NEXT tile's first K-tile would now wait for the prefetch, so that publication moves to a
    // barrier behind the epilogue's own vmcnt(0).
#define KTILE(P, SKIP, LAST, PFM)                                                                 \
    {                                                                                             \
        /* ---- even phase: A0 x (B0, B1) ---- */                                                 \
        if (!(SKIP)) {                                                                            \
            if ((PFM) == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + PFN) : "memory");         \
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                 \
        }                                                                                         \
        __builtin_amdgcn_s_barrier();                                                             \
        SB();                                                                                     \
        RD_A(fa[1], 1, 0, P) RD_B(1, P)                                                           \
        SB(); PRIO_ON()                                                                                     \
        acc[0][0][0] = mfma32(fb[0][0], fa[0][0], acc[0][0][0]);                                  \
        acc[0][0][1] = mfma32(fb[1][0], fa[0][0], acc[0][0][1]);                                  \
        DMA_A1(0, (P) ^ 1)                                                                        \
        acc[0][1][0] = mfma32(fb[0][0], fa[0][1], acc[0][1][0]);                                  \
        DMA_A1(1, (P) ^ 1)                                                                        \
        acc[0][1][1] = mfma32(fb[1][0], fa[0][1], acc[0][1][1]);                                  \
        PRIO_OFF() SB(); LGKM0(); SB();                                                                      \
        RD_A(fa[0], 2, 0, P) RD_B(2, P)                                                           \
        if ((PFM) == 1) epi_prefetch<EPI, AUX>(a, pm0, pn0, wm * 128, wn, lane, pf);              \
        SB(); PRIO_ON()                                                                                     \
        MM(0, fa[1], 1)                                                                           \
        PRIO_OFF() SB(); LGKM0(); SB();                                                                      \
        RD_A(fa[1], 3, 0, P) RD_B(3, P)                                                           \
        SB(); PRIO_ON()                                                                                     \
        MM(0, fa[0], 2)                                                                           \
        adv_a1();                                                                                 \
        PRIO_OFF() SB(); LGKM0(); SB();                                                                      \
        RD_A(fa[0], 0, 1, P)                                                                      \
        SB(); PRIO_ON()                                                                                     \
        MM(0, fa[1], 3)                                                                           \
        PRIO_OFF() SB(); LGKM0(); SB();                                                                      \
        /* ---- odd phase: A1 x (B0, B1), B fragments from registers ---- */                      \
        if ((PFM) == 1) { if (!(SKIP)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + PFN) : "memory"); } \
        else if ((PFM) == 0 || PFN == 0) { if (!(SKIP)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); } \
        __builtin_amdgcn_s_barrier();                                                             \
        SB();                                                                                     \
        RD_A(fa[1], 1, 1, P)                                                                      \
        SB(); PRIO_ON()                                                                                     \
        acc[1][0][0] = mfma32(fb[0][0], fa[0][0], acc[1][0][0]);                                  \
        DMA_A0(0, P)                                                                              \
        acc[1][0][1] = mfma32(fb[1][0], fa[0][0], acc[1][0][1]);                                  \
        DMA_A0(1, P)                                                                              \
        acc[1][1][0] = mfma32(fb[0][0], fa[0][1], acc[1][1][0]);                                  \
        DMA_B0(0, P)                                                                              \
        acc[1][1][1] = mfma32(fb[1][0], fa[0][1], acc[1][1][1]);                                  \
        DMA_B0(1, P)                                                                              \
        PRIO_OFF() SB(); LGKM0(); SB();                                                                      \
        RD_A(fa[0], 2, 1, P)                                                                      \
        SB(); PRIO_ON()                                                                                     \
        acc[1][0][0] = mfma32(fb[0][1], fa[1][0], acc[1][0][0]);                                  \
        DMA_B1(0, P)                                                                              \
        acc[1][0][1] = mfma32(fb[1][1], fa[1][0], acc[1][0][1]);                                  \
        DMA_B1(1, P)                                                                              \
        acc[1][1][0] = mfma32(fb[0][1], fa[1][1], acc[1][1][0]);                                  \
        acc[1][1][1] = mfma32(fb[1][1], fa[1][1], acc[1][1][1]);                                  \
        PRIO_OFF() SB(); LGKM0(); SB();                                                                      \
        RD_A(fa[1], 3, 1, P)                                                                      \
        SB(); PRIO_ON()                                                                                     \
        MM(1, fa[0], 2)                                                                           \
        adv_ab();                                                                                 \
        PRIO_OFF() SB(); LGKM0(); SB();                                                                      \
        /* first fragments of the next K-tile (ring parity P^1; published by this phase's barrier) */ \
        if (!(LAST)) { RD_A(fa[0], 0, 0, (P) ^ 1) RD_B(0, (P) ^ 1) }                              \
        SB(); PRIO_ON()                                                                                     \
        MM(1, fa[1], 3)                                                                           \
        PRIO_OFF() SB(); LGKM0(); SB();                                                                      \
    }

#define STAMP(IDX) if (DBG && dbg && i < 8) dbg[i * 8 + (IDX)] = wall_clock64();
#if (OCN_PRIO_MODE & 2)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    for (int i = 0; i < my_tiles; ++i) {
        STAMP(0)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int z = 0; z < 2; ++z)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[x][y][z][r] = 0.f;
        // first fragments of the tile (its K-tile 0 was published by a barrier before the previous epilogue / prologue)
        SB();
        RD_A(fa[0], 0, 0, 0) RD_B(0, 0)
        LGKM0();
        SB();
        int pm0, pn0, pks;
        tile_origin(i, pm0, pn0, pks);
        for (int kt = 0; kt + 2 < nk; kt += 2) {
            const bool skip = (kt == 0);
            KTILE(0, skip, false, 0)
            KTILE(1, false, false, 0)
        }
        {  // the tile's last two K-tiles (peeled: the epilogue-operand prefetch and its wait counts are compile-time)
            const bool skip = (nk == 2);
            KTILE(0, skip, false, 1)
            KTILE(1, false, true, 2)
        }
        STAMP(3)
        epilogue5<EPI, AUX>(a, acc, pm0, pn0, wm, wn, lane, stg, pf, 2, 0, (DBG && dbg && i < 8) ? dbg + i * 8 : nullptr, pks);
        if (ABL(a, 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // developer knob: let the tile's stores drain before the next main loop
        if constexpr (PFN > 0) __builtin_amdgcn_s_barrier();  // publishes the next tile's first K-tile (see KTILE, PFM = 2)
        STAMP(4)
    }
   }  // my_tiles > 0
    // ---- the launch's last round, in HALF tiles ---------------------------------------------------------------------------
    // ntiles is rarely a multiple of the grid: the N = 768 GEMMs of ViT-B-32 at batch 4096 have 2400 tiles for 256 CUs -- nine full rounds and
    // a tenth in which 96 workgroups compute a whole tile each while 160 CUs idle (6 % of the launch; 8.5 % for the text tower's N = 512
    // shapes).  When at most half of the workgroups would get a tail tile, each tail tile is split into its two 128 x 256 halves -- the rows
    // {wm*128 + h*64 + 0..63}, i.e. the A0 (h = 0) or A1 (h = 1) operand unit of every wave's strip -- and 2R workgroups compute one half each:
    // half the MFMAs of a tile on 3/4 of its operand bytes, no exchange between workgroups, nothing to reduce.  A half tile is ONE phase per
    // K-tile (16 MFMAs per wave: A x (B0, B1) with the B fragments read as they are used) and one barrier: the slot of K-tile g is free once
    // every wave has read its last fragments, which is where the barrier sits, and is refilled right behind it -- B two K-tiles ahead (it comes
    // from L2), A four K-tiles ahead through the two A units a half tile does not otherwise need (it comes from HBM).
    if constexpr (EPI != OCN_EPI_CE_ONEPASS && EPI != OCN_EPI_CE_ONEPASS_FULL) if (a.tail_n > 0 && pass == 0) {
        int t = -1, h = 0;
        if ((int)blockIdx.x < a.tail_n) t = (int)blockIdx.x;
        else if ((int)blockIdx.x >= a.tail_partner && (int)blockIdx.x < a.tail_partner + a.tail_n) { t = (int)blockIdx.x - a.tail_partner; h = 1; }
        if (t >= 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave has left the ring of the last whole tile
            int m0, n0;
            {
                const int tile = xcd_remap(t + a.tail_first, a.ntiles);
                const int cb = tile / band_tiles, r = tile - cb * band_tiles;
                const int width = min(a.band, a.tiles_n - cb * a.band);
                const int mi = r / width;
                m0 = mi * 256;
                n0 = (cb * a.band + (r - mi * width)) * 256;
            }
            dA0 = make_desc(a.A, m0, a.M, a.lda);
            dB = make_desc(a.B, n0, a.N, a.ldb);
            const unsigned hoff = h ? a_half : 0u;  // this half's rows: the A1 unit's offset
            unsigned kb = 0, ka = 0;                // byte offsets of the K-tiles the B / A cursors issue next (clamped at the end: junk, never read)
            const unsigned k_last = k_end - 128u;
#define HDMA_A(SLOT) { const unsigned so_ = (ka < k_end ? ka : k_last) + hoff; DMA(dA0, voA, so_, SLOT, 0) DMA(dA0, voA, so_, SLOT, 1) ka += 128; }
#define HDMA_B(P) { const unsigned so_ = kb < k_end ? kb : k_last; DMA(dB, voB, so_, B_SLOT(0, P), 0) DMA(dB, voB, so_, B_SLOT(0, P), 1) \
                    DMA(dB, voB, so_ + b_half, B_SLOT(1, P), 0) DMA(dB, voB, so_ + b_half, B_SLOT(1, P), 1) kb += 128; }
            // the six pieces of a K-tile's refill, one at a time between MFMAs (B pieces first: the wait counts below rely on that order)
#define HPIECE(N, P, SLOT)                                                                                               \
    {                                                                                                                    \
        if ((N) < 4) { const unsigned so_ = (kb < k_end ? kb : k_last) + (((N) & 2) ? b_half : 0u); DMA(dB, voB, so_, B_SLOT(((N) >> 1) & 1, P), (N) & 1) } \
        else { const unsigned so_ = (ka < k_end ? ka : k_last) + hoff; DMA(dA0, voA, so_, SLOT, (N) & 1) }              \
    }
            HDMA_A(A_SLOT(0, 0)) HDMA_B(0) HDMA_A(A_SLOT(0, 1)) HDMA_B(1) HDMA_A(A_SLOT(1, 0)) HDMA_A(A_SLOT(1, 1))
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int z = 0; z < 2; ++z)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][y][z][r] = 0.f;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            SB();
            RD_A(fa[0], 0, 0, 0) RD_B(0, 0)
            LGKM0();
            SB();
            // K-tile g = 4 n + Q: A unit in ring slot Q (= A_SLOT(Q >> 1, Q & 1)), B units in parity Q & 1
#define HKT(Q)                                                                                          \
    {                                                                                                   \
        RD_A(fa[1], 1, (Q) >> 1, (Q) & 1) RD_B(1, (Q) & 1)                                              \
        SB();                                                                                           \
        MM(0, fa[0], 0)                                                                                 \
        SB(); LGKM0(); SB();                                                                            \
        RD_A(fa[0], 2, (Q) >> 1, (Q) & 1) RD_B(2, (Q) & 1)                                              \
        SB();                                                                                           \
        MM(0, fa[1], 1)                                                                                 \
        SB(); LGKM0(); SB();                                                                            \
        RD_A(fa[1], 3, (Q) >> 1, (Q) & 1) RD_B(3, (Q) & 1)                                              \
        SB();                                                                                           \
        MM(0, fa[0], 2)                                                                                 \
        SB(); LGKM0(); SB();                                                                            \
        /* this wave has read the last fragment of K-tile g; its own pieces of K-tile g + 1 have landed when at most the two A pieces issued */ \
        /* one K-tile ago are still in flight (the B pieces are issued in front of them) */            \
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                                \
        __builtin_amdgcn_s_barrier();                                                                   \
        SB();                                                                                           \
        RD_A(fa[0], 0, (((Q) + 1) & 3) >> 1, ((Q) + 1) & 1) RD_B(0, ((Q) + 1) & 1)                      \
        SB();                                                                                           \
        acc[0][0][0] = mfma32(fb[0][3], fa[1][0], acc[0][0][0]);                                        \
        HPIECE(0, (Q) & 1, 0) HPIECE(1, (Q) & 1, 0)                                                     \
        acc[0][0][1] = mfma32(fb[1][3], fa[1][0], acc[0][0][1]);                                        \
        HPIECE(2, (Q) & 1, 0) HPIECE(3, (Q) & 1, 0)                                                     \
        acc[0][1][0] = mfma32(fb[0][3], fa[1][1], acc[0][1][0]);                                        \
        HPIECE(4, 0, A_SLOT((Q) >> 1, (Q) & 1)) HPIECE(5, 0, A_SLOT((Q) >> 1, (Q) & 1))                  \
        acc[0][1][1] = mfma32(fb[1][3], fa[1][1], acc[0][1][1]);                                        \
        kb += 128; ka += 128;                                                                           \
        SB(); LGKM0(); SB();                                                                            \
    }
            for (int kt = 0; kt < nk; kt += 4) { HKT(0) HKT(1) HKT(2) HKT(3) }
#undef HKT
#undef HDMA_A
#undef HDMA_B
#undef HPIECE
            epi_prefetch<EPI, AUX>(a, m0, n0, wm * 128 + h * 64, wn, lane, pf);
            epilogue5<EPI, AUX>(a, acc, m0, n0, wm, wn, lane, stg, pf, 1, h * 64);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing prefetches must land before the LDS is released
    if constexpr (!RESCUE) break;
    else {
        // ---- this workgroup is done with what it had: look for a share nobody has started -------------------------------------------------
        bool got = false;
        for (;;) {
            __syncthreads();  // everybody is out of the ring and the staging areas (first round) / has read the mailbox (later rounds)
            if (threadIdx.x == 0) box[1] = 0x7fffffff;
            __syncthreads();
            for (int t = threadIdx.x; t < G - 1; t += 512) {  // candidates in rotated order: the finishers spread over the victims
                int w = (int)blockIdx.x + 1 + t;
                w -= w >= G ? G : 0;
                if (__hip_atomic_load(a.rescue + w * OCN_RESCUE_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < share_of(w)) atomicMin(box + 1, t);
            }
            __syncthreads();
            const int t = box[1];
            if (t == 0x7fffffff) break;  // every share has been started or handed out
            int w = (int)blockIdx.x + 1 + t;
            w -= w >= G ? G : 0;
            if (threadIdx.x == 0) box[0] = atomicAdd(a.rescue + w * OCN_RESCUE_STRIDE, 1);
            __syncthreads();
            const int k = __builtin_amdgcn_readfirstlane(box[0]);
            if (k < share_of(w)) {  // entry k of share w is this workgroup's (else: the owner started or another finisher was faster -- look again)
                vblk = __builtin_amdgcn_readfirstlane(w);
                ibase = k;
                my_tiles = 1;
                got = true;
                break;
            }
        }
        if (!got) break;
        __syncthreads();  // the mailbox has been read: wave 0's staging area is the epilogue's again
    }
  }  // pass
#undef KTILE
#undef MM
#undef RD_A
#undef RD_B
#undef DMA
#undef DMA_NT
#undef DMA_A1
#undef DMA_A0
#undef DMA_B0
#undef DMA_B1
}

int g_num_cu = 0;
}  // namespace
extern int g_ocn_tuning[16];
namespace {

// Band width of the tile walk (see tile_origin).  L2-miss read model per launch, in bytes:
//   row-major over all of N:  A once + B once per round of tiles per XCD when B does not stay in L2  = A + (ntiles / 32) * B
//   nb bands:                 A once per band + B once per XCD                                        = nb * A + 8 * B
// A band must leave room for the streaming A panels: band * panel <= 2.5 MiB of the XCD's 4 MiB.  `forced` = developer knob.
int nt5_band(int M, int N, int K, int forced) {
    const int tiles_n = ocn_cdiv(N, 256);
    if (forced > 0) return forced < tiles_n ? forced : tiles_n;
    const double panel = 256.0 * K * 2, Bt = (double)N * K * 2, At = (double)M * K * 2;
    const double l2_band = 2.5 * 1048576;
    if (Bt <= l2_band) return tiles_n;
    const double ntiles = (double)ocn_cdiv(M, 256) * tiles_n;
    const double row_major = At + (ntiles / 32.0 < 8.0 ? 8.0 : ntiles / 32.0) * Bt;
    for (int nb = 2; nb <= tiles_n; ++nb) {
        const int band = ocn_cdiv(tiles_n, nb);
        if (band * panel > l2_band) continue;
        return (nb * At + 8.0 * Bt < row_major) ? band : tiles_n;  // the first feasible nb is the cheapest banded walk
    }
    return tiles_n;
}

// Start stagger between four classes of workgroups (100 MHz ticks per class; see the kernel).  It was worth +5 % on the c_fc + GELU GEMM when the
// epilogues waited for their own operands and stored 4 bytes per element of the second output (round 1); with the operands fetched ahead by the
// main loop and the 8-bit second output it COSTS 1.4 % on the four GELU / dGELU shapes and 0.4 ms on the step (round 5, profiles/
// r05_nt5_start_stagger.txt: every run of three alternations), and no assignment of workgroups to classes (per XCD, per tile row) does better
// than none.  Off in the product; `forced` (developer build, us per class) brings it back for measurements: 0 = off, 63 = off, 62 = the old
// automatic rule (a quarter of the expected tile time for the heavy epilogues when a workgroup walks >= 12 tiles).
template <int EPI>
int nt5_stagger(int ntiles, int K, int forced) {
    if (forced == 0 || forced == 63) return 0;
    if (forced != 62) return forced * 100;
    constexpr bool heavy = (EPI == OCN_EPI_BIAS_GELU || EPI == OCN_EPI_BIAS_QUICKGELU || EPI == OCN_EPI_DGELU || EPI == OCN_EPI_BIAS_RESID_F32 || EPI == OCN_EPI_BIAS_RESID_BF16 || EPI == OCN_EPI_F32);
    if (!heavy || ntiles < 12 * g_num_cu) return 0;
    const int tile_us = (K / 64) * 2 + 6;
    return tile_us * 100 / 4;
}

template <int EPI, int AUXV>
int launch5_aux(GemmNtArgs a, int grid, hipStream_t st) {
    static bool set_ = false;
    if (!set_) {
        (void)hipFuncSetAttribute((const void*)gemm_nt5_kernel<EPI, false, AUXV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
#ifndef OCN_DEV_BUILD
        (void)hipFuncSetAttribute((const void*)gemm_nt5_kernel<EPI, false, AUXV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
#endif
        set_ = true;
    }
#ifndef OCN_DEV_BUILD  // (the developer build's many cache-policy instantiations exist in the static form only)
    a.rescue = grid > 1 ? ocn_rescue_board(st, grid) : nullptr;
    if (a.rescue) {
        hipLaunchKernelGGL((gemm_nt5_kernel<EPI, false, AUXV, true>), dim3(grid), dim3(512), LDS_BYTES, st, a);
        OCN_CHECK_LAUNCH("ocn_gemm_nt");
        return OCN_OK;
    }
#endif
    a.rescue = nullptr;
    hipLaunchKernelGGL((gemm_nt5_kernel<EPI, false, AUXV>), dim3(grid), dim3(512), LDS_BYTES, st, a);
    OCN_CHECK_LAUNCH("ocn_gemm_nt");
    return OCN_OK;
}

template <int EPI>
int launch5(GemmNtArgs a, hipStream_t st) {
    if (g_num_cu == 0) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_num_cu = n;
    }
#ifndef OCN_DEV_BUILD
    a.ablate = 0;
#endif
    a.tiles_n = ocn_cdiv(a.N, 256);
    if (a.ksplit < 1) a.ksplit = 1;
    a.ntiles = ocn_cdiv(a.M, 256) * a.tiles_n * a.ksplit;  // split-K: every K-slice of every output tile is an entry of the walk
    a.band = nt5_band(a.M, a.N, a.K, (a.ablate >> 8) & 31);
    a.stagger = nt5_stagger<EPI>(a.ntiles, a.K, (a.ablate >> 13) & 63);
    a.first_wave = g_num_cu;
#ifdef OCN_DEV_BUILD
    a.stagger_mode = g_ocn_tuning[3];
#else
    a.stagger_mode = 0;
#endif
    // One workgroup per CU.  Developer knob 10 = k launches k per CU with 1/k of the tiles each (they queue behind each other on a
    // CU): a workgroup that cannot start with the rest -- a CU held by another stream's kernel, e.g. a collective -- then delays the
    // launch by 1/k of its length instead of by half of it (one CU held: +50 % at k = 1, +11..18 % at k = 3;
    // profiles/r01_persistent_gemm_occupancy_hazard.txt).  In isolation k = 3 costs nothing
    // (profiles/r01_nt5_workgroups_per_cu_sweep.txt), in the training step it costs 1.0 % (235.0 -> 237.4 ms, same box), which is
    // more than collectives that are active for ~1.5 % of a step can take back -- so k = 1 stays the default.
    const int per_cu = g_ocn_tuning[10] > 0 ? g_ocn_tuning[10] : 1;
    const int grid = a.ntiles < g_num_cu * per_cu ? a.ntiles : g_num_cu * per_cu;
    // The last, partial round of tiles as half tiles (see the kernel): when R = ntiles mod grid tiles are left over for at most half of the
    // workgroups, 2R workgroups compute a 128 x 256 half each -- the two halves of a tile on the same XCD (partner offset a multiple of 8).
    // Developer knob 12 = 1 (OCN_DEV_BUILD): whole tail tiles as before (A/B).
    a.tail_first = a.tail_n = a.tail_partner = 0;
    {
        const int R = a.ntiles % grid, P = (R + 7) / 8 * 8;
        constexpr bool ce = (EPI == OCN_EPI_CE_ONEPASS || EPI == OCN_EPI_CE_ONEPASS_FULL);  // its row statistics are laid out per whole tile
        if (!ce && a.ksplit == 1 && a.ntiles > grid && R > 0 && R + P <= grid && (a.K / 64) % 4 == 0 && !ABL(a, 0x200000)) {
            a.tail_first = a.ntiles - R;
            a.tail_n = R;
            a.tail_partner = P;
        }
    }
    // Cache policy of the epilogue (template parameter AUX: bit 1 (2) = non-temporal stores, bit 3 (8) = non-temporal loads of the
    // residual / saved derivative; profiles/r01_nt5_cache_policy_sweep.txt, profiles/r04_nt5_epilogue_prefetch.txt):
    //   stores: non-temporal for the two-output GELU epilogue (256 KiB per tile that nothing re-reads before they are long evicted:
    //           streamed past the L2 they stop displacing the operand panels, +5..8 %) and for wide bf16 outputs (N >= 1024: the QKV
    //           projections +4..6 %, the dGELU output);
    //   loads:  non-temporal for the fp32 residual and the saved gelu' -- both read exactly once.
    constexpr bool is_gelu = (EPI == OCN_EPI_BIAS_GELU || EPI == OCN_EPI_BIAS_QUICKGELU);
    const bool st_nt = is_gelu || ((EPI == OCN_EPI_BF16 || EPI == OCN_EPI_DGELU) && a.N >= 1024);
    const bool ld_nt = (EPI == OCN_EPI_BIAS_RESID_F32 || EPI == OCN_EPI_BIAS_RESID_BF16 || EPI == OCN_EPI_DGELU);
#ifdef OCN_DEV_BUILD
    if (a.ablate & 64) {  // per-tile timeline into g_nt5_trace (tools/gemm_trace.py)
        static bool dbg_attr_set = false;
        if (!dbg_attr_set) {
            (void)hipFuncSetAttribute((const void*)gemm_nt5_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            dbg_attr_set = true;
        }
        hipLaunchKernelGGL((gemm_nt5_kernel<EPI, true>), dim3(grid), dim3(512), LDS_BYTES, st, a);
        OCN_CHECK_LAUNCH("ocn_gemm_nt");
        return OCN_OK;
    }
    // developer knob bits 2 / 8 flip the store / load policy, bit 16 fetches the A operand non-temporally
    const int aux = ((st_nt != ((a.ablate & 2) != 0)) ? 2 : 0) | ((ld_nt != ((a.ablate & 8) != 0)) ? 8 : 0) | ((a.ablate & 16) ? 16 : 0);
    switch (aux) {
        case 0: return launch5_aux<EPI, 0>(a, grid, st);
        case 2: return launch5_aux<EPI, 2>(a, grid, st);
        case 8: return launch5_aux<EPI, 8>(a, grid, st);
        case 10: return launch5_aux<EPI, 10>(a, grid, st);
        case 16: return launch5_aux<EPI, 16>(a, grid, st);
        case 18: return launch5_aux<EPI, 18>(a, grid, st);
        case 24: return launch5_aux<EPI, 24>(a, grid, st);
        default: return launch5_aux<EPI, 26>(a, grid, st);
    }
#else
    // the product library holds exactly the instantiations the rules above select
    if constexpr (is_gelu) return launch5_aux<EPI, 2>(a, grid, st);
    else if constexpr (EPI == OCN_EPI_BIAS_RESID_F32 || EPI == OCN_EPI_BIAS_RESID_BF16) return launch5_aux<EPI, 8>(a, grid, st);
    else if constexpr (EPI == OCN_EPI_DGELU) return st_nt ? launch5_aux<EPI, 10>(a, grid, st) : launch5_aux<EPI, 8>(a, grid, st);
    else if constexpr (EPI == OCN_EPI_BF16) return st_nt ? launch5_aux<EPI, 2>(a, grid, st) : launch5_aux<EPI, 0>(a, grid, st);
    else return launch5_aux<EPI, 0>(a, grid, st);
#endif
}

}  // namespace

extern "C" int ocn_debug_nt5_trace(long long* host_out /*[1024]*/) {
#ifdef OCN_DEV_BUILD
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nt5_trace), sizeof(long long) * 1024) == hipSuccess ? OCN_OK : OCN_ERR_LAUNCH;
#else
    (void)host_out;
    ocn_set_error("ocn_debug_nt5_trace: the per-tile timeline exists only in the developer build (open_clip_amd/build.py --dev)");
    return OCN_ERR_INVALID;
#endif
}

int ocn_launch_nt5(int epilogue, const GemmNtArgs& a, hipStream_t st) {
    // needs whole 128-k pairs of K-tiles, 16-byte aligned bf16 rows on the output side and vector-width columns
    if (a.K % 128 != 0 || a.N % 8 != 0 || a.ldc % 8 != 0) return 1;
    if (a.ksplit > 1 && (epilogue != OCN_EPI_F32 || a.K % (128 * a.ksplit) != 0 || a.bias)) return 1;  // split-K: fp32 partial sums into the caller's slabs
    if (epilogue == OCN_EPI_DGELU && a.bias) return 1;  // the persistent kernel's dGELU epilogue carries no bias (no caller has one)
    if ((long)a.ldc * 4 * 256 >= 0x7fffffffL) return 1;  // 32-bit buffer offsets inside a tile
    switch (epilogue) {
        case OCN_EPI_BF16: return launch5<OCN_EPI_BF16>(a, st);
        case OCN_EPI_BIAS_GELU: return launch5<OCN_EPI_BIAS_GELU>(a, st);
        case OCN_EPI_BIAS_QUICKGELU: return launch5<OCN_EPI_BIAS_QUICKGELU>(a, st);
        case OCN_EPI_BIAS_RESID_F32: return launch5<OCN_EPI_BIAS_RESID_F32>(a, st);
        case OCN_EPI_BIAS_RESID_BF16: return launch5<OCN_EPI_BIAS_RESID_BF16>(a, st);
        case OCN_EPI_DGELU: return launch5<OCN_EPI_DGELU>(a, st);
        case OCN_EPI_F32: return launch5<OCN_EPI_F32>(a, st);
        case OCN_EPI_CE_ONEPASS: return launch5<OCN_EPI_CE_ONEPASS>(a, st);
        case OCN_EPI_CE_ONEPASS_FULL: return launch5<OCN_EPI_CE_ONEPASS_FULL>(a, st);
    }
    return 1;
}
