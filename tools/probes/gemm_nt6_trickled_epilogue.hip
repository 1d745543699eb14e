// Persistent NT GEMM with the epilogue UNDER the next tile's main loop (gfx950):  C[M,N] = A[M,K] . B[N,K]^T (+ fused epilogue).
//
// Why a second persistent kernel next to gemm_nt5.hip (256x256 tile, everything in the main loop's favour): measured on the
// MI355X (profiles/r02_nt5_epilogue_ablation.txt), nt5's tile loop with its epilogue's memory traffic and VALU work removed
// runs the [204800x3072x768] GEMM in 0.81 ms; with the GELU epilogue it takes 1.18 ms, with the dGELU one 1.10, and the
// out-proj (fp32 residual) GEMM 0.36 instead of 0.22 ms: while a workgroup converts and stores a tile, its MFMA pipes idle,
// and the 128-256 KiB burst of stores then sits in front of the next tile's operand loads in the CU's in-order memory pipe.
// A 256x256 tile leaves no register or LDS space to park a finished tile (128 accumulator registers per lane + the ring).
//
// Here a workgroup (8 waves as 4 x 2) owns a 256x128 tile, a wave 64x64 of it = 64 accumulator registers, and the
// accumulators are DOUBLE-BUFFERED: while tile i accumulates into one set, the finished fp32 values of tile i-1 stay parked
// in the other and are converted / transposed through a 2 KiB per-wave LDS staging buffer / stored a slice per K-tile
// ("trickled") in the issue slots between the MFMAs of tile i.  The tile loop has no epilogue section at all.
// Price: 1.5x the L2 -> LDS operand traffic of the 256x256 tile (48 KiB per 64-wide K-tile for half the outputs).
//
// K-tile = 64 k = 128-byte LDS rows (full-line LDS-DMA pieces of 8 rows x 128 B).  Ring = 3 slots x (A 32 KiB + B 16 KiB).
// One phase per K-tile, one raw s_barrier + one counted vmcnt per phase:
//   phase g:  s_waitcnt vmcnt(..) [this wave's pieces of K-tile g have landed]  ->  s_barrier [everybody's have; everybody
//             has retired its fragment reads of K-tile g-1]  ->  trickle memory ops, then the 6 LDS-DMAs of K-tile g+2 into the
//             slot K-tile g-1 just left  ->  fragment reads / MFMAs of K-tile g, skewed by one k-substep across the barrier
//             (the last 4 MFMAs of K-tile g-1 cover the first fragment reads of K-tile g).
// vmcnt is in order, and every memory instruction of the loop is inline asm (hipcc never sees one, so it never inserts a
// wait of its own): with the trickle ops issued BEFORE the phase's DMAs, the phase-start wait "all but the previous phase's
// ops" = vmcnt(T(prev) + 6) also guarantees that every trickle load / store issued two phases ago has completed.
// EXPERIMENT, NOT PRODUCT CODE (round 2): measured slower than gemm_nt5 on every shape -- see profiles/r02_nt6_trickled_epilogue_experiment.txt.
// To try it again: copy next to gemm_nt5.hip, declare ocn_launch_nt6 in gemm_args.h, dispatch variant 6 in gemm.hip::launch_nt, tools/nt6_check.py.
#include "../../open_clip_amd/csrc/gemm_args.h"

namespace {

constexpr int SLOT6 = 49152;             // one K-tile: A 256 rows x 128 B, then B 128 rows x 128 B
constexpr int RING6 = 3 * SLOT6;         // 147456
constexpr int STG6 = 2048;               // per-wave staging: one 32x32 bf16 block / half a 32x32 fp32 block (64-byte rows)
constexpr int LDS6 = RING6 + 8 * STG6;   // 163840 = all of a CU's LDS

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define DSR6(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF))
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define SB() __builtin_amdgcn_sched_barrier(0)
#define VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory")

OCN_DEV void lds6_w64(unsigned addr, bf16x4 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
OCN_DEV void lds6_w128(unsigned addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
OCN_DEV void lds6_w128u(unsigned addr, u32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <typename T>
OCN_DEV void lds6_r128x2(unsigned a0, T& d0, T& d1) {  // rows [0,16) and [16,32) of the staging image; waits for the data
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(d0), "=&v"(d1) : "v"(a0) : "memory");
}
OCN_DEV void lds6_r64x4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, bf16x4& d0, bf16x4& d1, bf16x4& d2, bf16x4& d3) {
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                 : "memory");
}

// Global accesses of the epilogue: buffer instructions on a descriptor based at the tile's first row (rows beyond M are dropped /
// read as zero by the bounds check).  Inline asm, so that hipcc's vmcnt bookkeeping never meets them (see the header).
OCN_DEV void st128(u32x4 v, u32x4 desc, unsigned off, bool nt) {
    if (nt) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt\n\ts_nop 1" ::"v"(v), "v"(off), "s"(desc) : "memory");
    else asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(off), "s"(desc) : "memory");
}
OCN_DEV void ld128(u32x4& v, u32x4 desc, unsigned off) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(desc) : "memory");
}

struct Tile6 {      // what the epilogue of one tile needs (all wave-uniform)
    u32x4 d_out;    // out rows [m0, M)
    u32x4 d_aux;    // aux rows [m0, M)   (GELU: second output; dGELU: saved gelu')
    u32x4 d_res;    // residual rows [m0, M)
    int n0;
};

OCN_DEV u32x4 make_desc6(const void* base, long row0, int rows, int ld, int esz) {
    long bytes = (long)(rows - row0) * ld * esz;
    bytes = bytes > 0x7fffffffL ? 0x7fffffffL : (bytes < 0 ? 0 : bytes);
    const unsigned long long p = (unsigned long long)((const char*)base + row0 * ld * esz);
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes);
    r[3] = 0x00020000u;
    return r;
}

// ---- epilogue rounds ------------------------------------------------------------------------------------------------------------
// A lane of the C^T accumulators owns ONE row (lr) of a 32x32 block and, per register quad g, the 4 columns 8g + 4*lh + {0..3}.
// Staging image: 32 rows x 64 bytes; 16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 3).
// Lane constants of a round (the row / chunk this lane reads back and stores): rr = lane >> 2 (+16), rc = lane & 3.
struct Lane6 {
    unsigned wr;   // staging write base: stg + lr * 64 (chunk swizzle applied per access)
    unsigned rd;   // staging read address of row rr, chunk rc (rows rr + 16 at +1024)
    int lr, lh, rr, rc, sw_w;
};

OCN_DEV Lane6 lane6(unsigned stg, int lane) {
    Lane6 l;
    l.lr = lane & 31; l.lh = lane >> 5; l.rr = lane >> 2; l.rc = lane & 3;
    l.sw_w = (l.lr >> 1) & 3;
    l.wr = stg + l.lr * 64;
    l.rd = stg + l.rr * 64 + ((l.rc ^ ((l.rr >> 1) & 3)) << 4);  // (rr + 16) >> 1 & 3 == rr >> 1 & 3
    return l;
}

// write one block's 4 packed quads (bf16x4 = 8 bytes: chunk g, half lh) / read back 2 x 16 rows x 64 B
OCN_DEV void stage_w_bf16(const Lane6& l, const bf16x4 (&pk)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) lds6_w64(l.wr + (((g ^ l.sw_w) << 4) | (l.lh << 3)), pk[g]);
}
// byte offset (from the tile descriptor's base) of this lane's 16-byte piece of rows rr (+16) of block (ha, hb); esz = element size
OCN_DEV unsigned out_off(const Lane6& l, int row_w, int col_g, int ha, int ldc, int esz, int cols_per_chunk) {
    return (unsigned)(((row_w + ha * 32 + l.rr) * ldc + col_g + l.rc * cols_per_chunk) * esz);
}

template <int EPI>
OCN_DEV void epi_block_sync(const GemmNtArgs& a, const Tile6& t, f32x16& acc, int ha, int hb, int row_w, int col_w, const Lane6& l, bool st_nt) {
    // one 32x32 block, start to finish (the synchronous form: final tile of a workgroup, and the reference for the trickled rounds)
    const int col_g = t.n0 + col_w + hb * 32;
    const unsigned row16 = (unsigned)(16 * a.ldc);
    if constexpr (EPI == OCN_EPI_BF16 || EPI == OCN_EPI_BIAS_GELU || EPI == OCN_EPI_DGELU) {
        bf16x4 pk[4], pk2[4];
        bf16x4 dg[4];
        if constexpr (EPI == OCN_EPI_DGELU) {  // saved gelu' of this block: rows x 64 B -> staging -> accumulator layout
            const unsigned o = out_off(l, row_w, col_g, ha, a.ldc, 2, 8);
            u32x4 x0, x1;
            ld128(x0, t.d_aux, o);
            ld128(x1, t.d_aux, o + row16 * 2u);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0), "+v"(x1)::"memory");
            lds6_w128u(l.rd, x0);
            lds6_w128u(l.rd + 1024, x1);
            const unsigned b0 = l.wr + (l.lh << 3);
            lds6_r64x4(b0 + ((0 ^ l.sw_w) << 4), b0 + ((1 ^ l.sw_w) << 4), b0 + ((2 ^ l.sw_w) << 4), b0 + ((3 ^ l.sw_w) << 4), dg[0], dg[1], dg[2], dg[3]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            if constexpr (EPI == OCN_EPI_BIAS_GELU) {
                f32x4 gv = v, dv = v;  // (developer knob 1: skip the VALU work)
                if (!(a.ablate & 1)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float g1, d1;
                        gelu_both(v[e], g1, d1);
                        gv[e] = g1;
                        dv[e] = d1;
                    }
                }
                pk[g] = (bf16x4){f2bf(gv[0]), f2bf(gv[1]), f2bf(gv[2]), f2bf(gv[3])};
                pk2[g] = (bf16x4){f2bf(dv[0]), f2bf(dv[1]), f2bf(dv[2]), f2bf(dv[3])};
            } else if constexpr (EPI == OCN_EPI_DGELU) {
                pk[g] = (bf16x4){f2bf(v[0] * bf2f(dg[g][0])), f2bf(v[1] * bf2f(dg[g][1])), f2bf(v[2] * bf2f(dg[g][2])), f2bf(v[3] * bf2f(dg[g][3]))};
            } else {
                pk[g] = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            }
        }
        const unsigned o = out_off(l, row_w, col_g, ha, a.ldc, 2, 8);
        u32x4 d0, d1;
        stage_w_bf16(l, pk);
        lds6_r128x2(l.rd, d0, d1);
        st128(d0, t.d_out, o, st_nt);
        st128(d1, t.d_out, o + row16 * 2u, st_nt);
        if constexpr (EPI == OCN_EPI_BIAS_GELU) {
            stage_w_bf16(l, pk2);
            lds6_r128x2(l.rd, d0, d1);
            st128(d0, t.d_aux, o, st_nt);
            st128(d1, t.d_aux, o + row16 * 2u, st_nt);
        }
    } else {
        // fp32 outputs: two half blocks (16 columns = 64-byte rows); the residual is fetched in the store layout
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned o = out_off(l, row_w, col_g + 16 * h, ha, a.ldc, 4, 4);
            u32x4 r0, r1;
            if constexpr (EPI == OCN_EPI_BIAS_RESID_F32) {
                ld128(r0, t.d_res, o);
                ld128(r1, t.d_res, o + row16 * 4u);
            }
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int g = 2 * h + gg;
                const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                lds6_w128(l.wr + (((2 * gg + l.lh) ^ l.sw_w) << 4), v);
            }
            f32x4 d0, d1;
            lds6_r128x2(l.rd, d0, d1);
            if constexpr (EPI == OCN_EPI_BIAS_RESID_F32) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1)::"memory");
                d0 += __builtin_bit_cast(f32x4, r0);
                d1 += __builtin_bit_cast(f32x4, r1);
            }
            st128(__builtin_bit_cast(u32x4, d0), t.d_out, o, st_nt);
            st128(__builtin_bit_cast(u32x4, d1), t.d_out, o + row16 * 4u, st_nt);
        }
    }
}

// ---- trickled epilogue --------------------------------------------------------------------------------------------------------------
// The parked tile (previous tile's finished accumulators, bias already inside) leaves in the first 7 phases of the next tile.  A phase
// offers: `pre` (right after the barrier), `vm` (memory instructions, issued before the phase's 6 DMAs -- tf6<EPI, J>() of them, which
// is what the NEXT phase's vmcnt allows to stay in flight), four `slot`s (one per group of 4 MFMAs; slot 0 shares its group with the
// DMAs and, in phase 0, with the parked tile's own last MFMAs, so nothing touches the parked registers there), and around the
// lgkmcnt(0) that ends each slot `bnd` (LDS reads issued just before it: it retires them) and `aft` (just after it).
// Blocks b = 0..3 = (ha, hb) = (b >> 1, b & 1).  Schedules (phase numbers J):
//   bf16          convert block b (cvt, 4 x ds_write_b64, read back 2 x 16 rows) in J = b+1, its 2 stores open J = b+2
//   bias + GELU   convert block b in J = b+1 (one quad = 4 elements per slot: ~1 element per MFMA); gelu(x) rows are read back at the end
//                 of J = b+1, the gelu' quads wait in registers and pass through the staging image in slot 0 of J = b+2;
//                 stores: gelu(b) opens J = b+2, gelu'(b) opens J = b+3
//   dGELU         saved gelu' of block b: 2 loads (store layout) open J = b; in J = b+2 they go through the staging image into the
//                 accumulator layout, are multiplied in, converted, staged again and read back; 2 stores open J = b+3
//   fp32 (+resid) stores straight from the accumulator layout (16 B per lane, no LDS): block b opens J = b+1; with a residual its
//                 4 loads (same layout) open J = b and the 4 stores of (acc + resid) open J = b+2
struct Trk6 {
    u32x4 d0, d1;    // staged rows of the block just converted (stored when the next phase opens)
    u32x4 e0, e1;    // GELU: staged rows of the gelu' block
    bf16x4 pd[4];    // GELU: gelu' quads of the block being converted
    bf16x4 pk0;      // GELU: quad 0 of the block being converted (the staging image is still being read back when it is ready)
    u32x4 x[2][2];   // dGELU: saved gelu' of the two blocks in flight (store layout)
    bf16x4 dg[4];    // dGELU: the current block's gelu' in accumulator layout
    u32x4 t[2][4];   // residual: the 4 quads of the two blocks in flight (accumulator layout)
};
struct Ctx6 {
    Tile6 tp;        // the parked tile (zero-sized descriptors while there is none)
    int ldc, row_w, col_w, ablate;
    unsigned omask;  // developer knob 4: store / load offsets wrapped into a 64 KiB window of the first tile (L2-resident: what the epilogue
                     // costs when its memory operations complete quickly); ~0u otherwise
    Lane6 L;
};

template <int EPI, int J>
constexpr int tf6() {
    if (EPI == OCN_EPI_BF16) return (J >= 2 && J <= 5) ? 2 : 0;
    if (EPI == OCN_EPI_BIAS_GELU) return (J == 2 || J == 6) ? 2 : ((J >= 3 && J <= 5) ? 4 : 0);
    if (EPI == OCN_EPI_DGELU) return (J >= 0 && J <= 2) ? 2 : (J == 3 ? 4 : ((J >= 4 && J <= 6) ? 2 : 0));
    if (EPI == OCN_EPI_BIAS_RESID_F32) return (J == 0 || J == 1 || J == 4 || J == 5) ? 4 : ((J == 2 || J == 3) ? 8 : 0);
    if (EPI == OCN_EPI_F32) return (J >= 1 && J <= 4) ? 4 : 0;
    return 0;
}

OCN_DEV unsigned blk_off_rows(const Ctx6& c, int b, int esz, int cols_per_chunk) {  // store-layout offset of block b (rows rr; +16 rows: + 16*ldc*esz)
    return out_off(c.L, c.row_w, c.tp.n0 + c.col_w + (b & 1) * 32, b >> 1, c.ldc, esz, cols_per_chunk) & c.omask;
}
OCN_DEV unsigned blk_off_acc(const Ctx6& c, int b, int g) {  // accumulator-layout offset (fp32) of quad g of block b: row lr, columns 8g + 4*lh
    return (unsigned)(((c.row_w + (b >> 1) * 32 + c.L.lr) * c.ldc + c.tp.n0 + c.col_w + (b & 1) * 32 + 8 * g + 4 * c.L.lh) * 4) & c.omask;
}

template <int EPI, int J>
OCN_DEV void trk_pre(const Ctx6& c, f32x16 (&park)[2][2], Trk6& s) {
    if constexpr (EPI == OCN_EPI_DGELU && J >= 2 && J <= 5) {  // saved gelu' of block J-2 (landed: issued two phases ago) -> staging
        constexpr int b = J - 2;
        lds6_w128u(c.L.rd, s.x[b & 1][0]);
        lds6_w128u(c.L.rd + 1024, s.x[b & 1][1]);
    }
}

template <int EPI, int J>
OCN_DEV void trk_vm(const Ctx6& c, f32x16 (&park)[2][2], Trk6& s) {
    const unsigned r16 = (unsigned)(16 * c.ldc);
    if constexpr (EPI == OCN_EPI_BF16) {
        if constexpr (J >= 2 && J <= 5) {
            const unsigned o = blk_off_rows(c, J - 2, 2, 8);
            st128(s.d0, c.tp.d_out, o, false);
            st128(s.d1, c.tp.d_out, (o + r16 * 2u) & c.omask, false);
        }
    } else if constexpr (EPI == OCN_EPI_BIAS_GELU) {
        if constexpr (J >= 2 && J <= 5) {
            const unsigned o = blk_off_rows(c, J - 2, 2, 8);
            st128(s.d0, c.tp.d_out, o, false);
            st128(s.d1, c.tp.d_out, (o + r16 * 2u) & c.omask, false);
        }
        if constexpr (J >= 3 && J <= 6) {
            const unsigned o = blk_off_rows(c, J - 3, 2, 8);
            st128(s.e0, c.tp.d_aux, o, false);
            st128(s.e1, c.tp.d_aux, (o + r16 * 2u) & c.omask, false);
        }
    } else if constexpr (EPI == OCN_EPI_DGELU) {
        if constexpr (J >= 3 && J <= 6) {
            const unsigned o = blk_off_rows(c, J - 3, 2, 8);
            st128(s.d0, c.tp.d_out, o, false);
            st128(s.d1, c.tp.d_out, (o + r16 * 2u) & c.omask, false);
        }
        if constexpr (J >= 0 && J <= 3) {
            const unsigned o = blk_off_rows(c, J, 2, 8);
            ld128(s.x[J & 1][0], c.tp.d_aux, o);
            ld128(s.x[J & 1][1], c.tp.d_aux, (o + r16 * 2u) & c.omask);
        }
    } else if constexpr (EPI == OCN_EPI_F32) {
        if constexpr (J >= 1 && J <= 4) {
            constexpr int b = J - 1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {park[b >> 1][b & 1][4 * g], park[b >> 1][b & 1][4 * g + 1], park[b >> 1][b & 1][4 * g + 2], park[b >> 1][b & 1][4 * g + 3]};
                st128(__builtin_bit_cast(u32x4, v), c.tp.d_out, blk_off_acc(c, b, g), false);
            }
        }
    } else if constexpr (EPI == OCN_EPI_BIAS_RESID_F32) {
        if constexpr (J >= 2 && J <= 5) {
            constexpr int b = J - 2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {park[b >> 1][b & 1][4 * g], park[b >> 1][b & 1][4 * g + 1], park[b >> 1][b & 1][4 * g + 2], park[b >> 1][b & 1][4 * g + 3]};
                st128(__builtin_bit_cast(u32x4, v + __builtin_bit_cast(f32x4, s.t[b & 1][g])), c.tp.d_out, blk_off_acc(c, b, g), false);
            }
        }
        if constexpr (J >= 0 && J <= 3) {
#pragma unroll
            for (int g = 0; g < 4; ++g) ld128(s.t[J & 1][g], c.tp.d_res, blk_off_acc(c, J, g));
        }
    }
}

template <int EPI, int J, int S>
OCN_DEV void trk_slot(const Ctx6& c, f32x16 (&park)[2][2], Trk6& s) {
    if constexpr (EPI == OCN_EPI_BF16) {
        if constexpr (J >= 1 && J <= 4) {  // quad S of block J-1
            constexpr int b = J - 1;
            const f32x16& q = park[b >> 1][b & 1];
            const bf16x4 pk = {f2bf(q[4 * S]), f2bf(q[4 * S + 1]), f2bf(q[4 * S + 2]), f2bf(q[4 * S + 3])};
            lds6_w64(c.L.wr + (((S ^ c.L.sw_w) << 4) | (c.L.lh << 3)), pk);
        }
    } else if constexpr (EPI == OCN_EPI_BIAS_GELU) {
        if constexpr (S == 0 && J >= 2 && J <= 5) stage_w_bf16(c.L, s.pd);  // gelu' quads of block J-2 into the (read-back) staging image
        if constexpr (J >= 1 && J <= 4) {
            constexpr int b = J - 1;
            const f32x16& q = park[b >> 1][b & 1];
            f32x4 gv, dv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float g1 = q[4 * S + e], d1 = q[4 * S + e];
                if (!(c.ablate & 1)) gelu_both(q[4 * S + e], g1, d1);
                gv[e] = g1;
                dv[e] = d1;
            }
            const bf16x4 pk = {f2bf(gv[0]), f2bf(gv[1]), f2bf(gv[2]), f2bf(gv[3])};
            s.pd[S] = (bf16x4){f2bf(dv[0]), f2bf(dv[1]), f2bf(dv[2]), f2bf(dv[3])};
            if constexpr (S == 0) s.pk0 = pk;  // the staging image is read back first (trk_bnd / trk_aft of slot 0)
            else lds6_w64(c.L.wr + (((S ^ c.L.sw_w) << 4) | (c.L.lh << 3)), pk);
        }
    } else if constexpr (EPI == OCN_EPI_DGELU) {
        if constexpr (S == 1 && J >= 2 && J <= 5) {  // block J-2: acc * gelu' -> bf16 -> staging
            constexpr int b = J - 2;
            const f32x16& q = park[b >> 1][b & 1];
            bf16x4 pk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pk[g] = (bf16x4){f2bf(q[4 * g] * bf2f(s.dg[g][0])), f2bf(q[4 * g + 1] * bf2f(s.dg[g][1])), f2bf(q[4 * g + 2] * bf2f(s.dg[g][2])),
                                 f2bf(q[4 * g + 3] * bf2f(s.dg[g][3]))};
            stage_w_bf16(c.L, pk);
        }
    }
}

// LDS reads issued right before the lgkmcnt(0) that ends slot S (no wait of their own)
template <int EPI, int J, int S>
OCN_DEV void trk_bnd(const Ctx6& c, Trk6& s) {
    constexpr bool rd_d = (EPI == OCN_EPI_BF16 && S == 3 && J >= 1 && J <= 4) || (EPI == OCN_EPI_BIAS_GELU && S == 3 && J >= 1 && J <= 4) ||
                          (EPI == OCN_EPI_DGELU && S == 2 && J >= 2 && J <= 5);
    if constexpr (rd_d) asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(s.d0), "=&v"(s.d1) : "v"(c.L.rd) : "memory");
    if constexpr (EPI == OCN_EPI_BIAS_GELU && S == 0 && J >= 2 && J <= 5)
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(s.e0), "=&v"(s.e1) : "v"(c.L.rd) : "memory");
    if constexpr (EPI == OCN_EPI_DGELU && S == 0 && J >= 2 && J <= 5) {
        const unsigned b0 = c.L.wr + (c.L.lh << 3);
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7"
                     : "=&v"(s.dg[0]), "=&v"(s.dg[1]), "=&v"(s.dg[2]), "=&v"(s.dg[3])
                     : "v"(b0 + ((0 ^ c.L.sw_w) << 4)), "v"(b0 + ((1 ^ c.L.sw_w) << 4)), "v"(b0 + ((2 ^ c.L.sw_w) << 4)), "v"(b0 + ((3 ^ c.L.sw_w) << 4))
                     : "memory");
    }
}
// right after that lgkmcnt(0): the values read in trk_bnd are there (passed through an asm so that nothing is scheduled above the wait)
template <int EPI, int J, int S>
OCN_DEV void trk_aft(const Ctx6& c, Trk6& s) {
    constexpr bool rd_d = (EPI == OCN_EPI_BF16 && S == 3 && J >= 1 && J <= 4) || (EPI == OCN_EPI_BIAS_GELU && S == 3 && J >= 1 && J <= 4) ||
                          (EPI == OCN_EPI_DGELU && S == 2 && J >= 2 && J <= 5);
    if constexpr (rd_d) asm volatile("" : "+v"(s.d0), "+v"(s.d1));
    if constexpr (EPI == OCN_EPI_BIAS_GELU && S == 0) {
        if constexpr (J >= 2 && J <= 5) asm volatile("" : "+v"(s.e0), "+v"(s.e1));
        if constexpr (J >= 1 && J <= 4) lds6_w64(c.L.wr + (((0 ^ c.L.sw_w) << 4) | (c.L.lh << 3)), s.pk0);
    }
    if constexpr (EPI == OCN_EPI_DGELU && S == 0 && J >= 2 && J <= 5) asm volatile("" : "+v"(s.dg[0]), "+v"(s.dg[1]), "+v"(s.dg[2]), "+v"(s.dg[3]));
}
// the phase-start wait: everything but the previous phase's memory instructions (its tf6 trickle ops + 6 DMAs) has completed; the
// registers that loads of two phases ago were aimed at pass through it
template <int EPI, int J>
OCN_DEV void trk_wait(Trk6& s) {
    constexpr int N = 6 + (J >= 1 && J <= 8 ? tf6<EPI, (J >= 1 ? J - 1 : 0)>() : 0);
    if constexpr (EPI == OCN_EPI_DGELU && J >= 2 && J <= 5) {
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(s.x[J & 1][0]), "+v"(s.x[J & 1][1]) : "i"(N) : "memory");
    } else if constexpr (EPI == OCN_EPI_BIAS_RESID_F32 && J >= 2 && J <= 5) {
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(s.t[J & 1][0]), "+v"(s.t[J & 1][1]), "+v"(s.t[J & 1][2]), "+v"(s.t[J & 1][3]) : "i"(N) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
    }
}

// accumulator start value of a tile = its bias (the epilogue then never adds it): column 8g + 4*lh + e of block hb.  SCALAR buffer
// loads in inline asm: a vector load here would make hipcc wait vmcnt(0) -- i.e. for every LDS-DMA in flight -- and hipcc does not
// turn loads through a by-value kernel-argument pointer into s_load on its own.  A NULL bias gets a zero-sized descriptor: the
// bounds check returns 0.
typedef __attribute__((ext_vector_type(8))) float f32x8;
OCN_DEV void init_acc(u32x4 d_bias, f32x16 (&acc)[2][2], int n0, int col_w, int lh) {
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        f32x8 b[4];
        const unsigned o0 = (unsigned)(n0 + col_w + hb * 32) * 4u, o1 = o0 + 32u, o2 = o0 + 64u, o3 = o0 + 96u;
        asm volatile("s_buffer_load_dwordx8 %0, %4, %5\n\ts_buffer_load_dwordx8 %1, %4, %6\n\ts_buffer_load_dwordx8 %2, %4, %7\n\t"
                     "s_buffer_load_dwordx8 %3, %4, %8\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(b[0]), "=&s"(b[1]), "=&s"(b[2]), "=&s"(b[3])
                     : "s"(d_bias), "s"(o0), "s"(o1), "s"(o2), "s"(o3)
                     : "memory");
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = lh ? b[g][4 + e] : b[g][e];
                acc[0][hb][4 * g + e] = v;
                acc[1][hb][4 * g + e] = v;
            }
    }
}

// MODE 1 = the product; 0 = the same main loop with a synchronous epilogue per tile; 2 = the phase structure of mode 1 without the
// trickle work (only the last tile of a workgroup is written: timing experiments)
template <int EPI, int MODE>
__global__ __launch_bounds__(512, 2) void gemm_nt6_kernel(GemmNtArgs a) {
    constexpr bool TRK = (MODE == 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int nk = a.K >> 6;
    const int G = gridDim.x;
    const int my_tiles = (a.ntiles - (int)blockIdx.x + G - 1) / G;
    const unsigned lds_base = (unsigned)(size_t)(OCN_LDS char*)smem;
    const int row_w = wm * 64, col_w = wn * 64;

    // fragment read addresses (slot 0): row (strip + lane&31) * 128 + chunk ((ks*2 + lh) ^ swz) * 16
    unsigned va0[4], vb0[4];
    {
        const int q = lh ^ swz_nt(lr);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned o = (unsigned)((q << 4) ^ (ks << 5));
            va0[ks] = lds_base + (unsigned)((row_w + lr) * 128) + o;
            vb0[ks] = lds_base + 32768u + (unsigned)((col_w + lr) * 128) + o;
        }
    }
    // DMA source offsets of this wave's pieces: A rows wave*32 + j*8 + (lane>>3) (j = 0..3), B rows wave*16 + j*8 + (lane>>3) (j = 0..1);
    // pieces j and j+2 differ by 16 rows (same swizzle) -> 2 lane offsets + a scalar row-block offset
    unsigned voA[2], voB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ra = wave * 32 + j * 8 + (lane >> 3), rb = wave * 16 + j * 8 + (lane >> 3);
        voA[j] = (unsigned)(ra * a.lda * 2 + (((lane & 7) ^ swz_nt(ra)) << 4));
        voB[j] = (unsigned)(rb * a.ldb * 2 + (((lane & 7) ^ swz_nt(rb)) << 4));
    }
    const unsigned a16 = (unsigned)(16 * a.lda * 2);

    const int tiles_m = a.ntiles / a.tiles_n;
    const int band_tiles = tiles_m * a.band;
    auto tile_origin = [&](int i, int& m0, int& n0) {
        const int tile = xcd_remap((int)blockIdx.x + i * G, a.ntiles);
        const int cb = tile / band_tiles, r = tile - cb * band_tiles;
        const int width = min(a.band, a.tiles_n - cb * a.band);
        const int mi = r / width;
        m0 = mi * 256;
        n0 = (cb * a.band + (r - mi * width)) * 128;
    };
    u32x4 dA, dB;        // descriptors of the tile the DMA cursor is in
    int c_i = 0;         // tile ordinal of the cursor
    unsigned c_k = 0;    // byte offset of the K-tile it issues next
    const unsigned k_end = (unsigned)nk * 128u;
    auto set_cursor = [&](int i) {
        int m0, n0;
        tile_origin(i, m0, n0);
        dA = make_desc6(a.A, m0, a.M, a.lda, 2);
        dB = make_desc6(a.B, n0, a.N, a.ldb, 2);
    };
    auto adv = [&]() {
        c_k += 128;
        if (c_k == k_end) {
            c_k = 0;
            if (c_i + 1 < my_tiles) set_cursor(++c_i);  // else: keep re-fetching the last tile (valid addresses, results unused)
        }
    };
    const unsigned dstA = lds_base + wave * 4096, dstB = lds_base + 32768u + wave * 2048;
#define DMA6(DESC, VOFF, SOFF, DST)                                                                                    \
    {                                                                                                                  \
        unsigned keep_;                                                                                                \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_)                                                                                    \
                     : "v"(VOFF), "s"(DESC), "s"(DST), "s"(SOFF)                                                       \
                     : "memory");                                                                                      \
    }
    // the 6 pieces of one K-tile into ring slot offset SO (bytes)
#define DMA_A(J, SO) DMA6(dA, voA[(J) & 1], c_k + ((J) >> 1) * a16, dstA + (SO) + (J) * 1024)
#define DMA_B(J, SO) DMA6(dB, voB[J], c_k, dstB + (SO) + (J) * 1024)

    f32x16 acc[2][2][2];  // [buffer][ha][hb]
    bf16x8 fa[2][2];      // A fragments [k-substep parity][ha]
    bf16x8 fb[2][2];      // B fragments [k-substep parity][hb]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
#pragma unroll
            for (int z = 0; z < 2; ++z)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][y][z][r] = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) { fa[x][y][r] = (bf16)0.f; fb[x][y][r] = (bf16)0.f; }
        }

    // ---- prologue: K-tiles 0 and 1 ----
    set_cursor(0);
    DMA_A(0, 0) DMA_A(1, 0) DMA_A(2, 0) DMA_A(3, 0) DMA_B(0, 0) DMA_B(1, 0)
    adv();
    DMA_A(0, SLOT6) DMA_A(1, SLOT6) DMA_A(2, SLOT6) DMA_A(3, SLOT6) DMA_B(0, SLOT6) DMA_B(1, SLOT6)
    adv();
    unsigned so_rd = 0, so_wr = 2 * SLOT6;  // ring slot (byte offset) of the K-tile being consumed / of the next DMA

    const u32x4 d_bias = make_desc6(a.bias, 0, a.bias ? 1 : 0, a.N, 4);  // N floats, or empty (reads as 0) without a bias
    const unsigned stg = lds_base + RING6 + wave * STG6;
    Ctx6 ctx;
    ctx.ldc = a.ldc; ctx.row_w = row_w; ctx.col_w = col_w; ctx.ablate = a.ablate;
    ctx.omask = (a.ablate & 4) ? 0xfff0u : ~0u;
    ctx.L = lane6(stg, lane);
    Trk6 trk;

    auto tile_desc = [&](int i) -> Tile6 {  // i < 0: no such tile -> zero-sized descriptors (stores dropped, loads read 0)
        int m0 = 0, n0 = 0;
        if (i >= 0) tile_origin(i, m0, n0);
        if (a.ablate & 4) m0 = 0;
        // developer knobs 32 / 128 (as in gemm_nt5.hip): zero-sized descriptors drop the epilogue's stores / operand loads
        const int m_st = (i < 0 || (a.ablate & 32)) ? 0 : a.M, m_ld = (i < 0 || (a.ablate & 128)) ? 0 : a.M;
        Tile6 t;
        t.n0 = n0;
        t.d_out = make_desc6(a.out, m0, m_st, a.ldc, (EPI == OCN_EPI_BIAS_RESID_F32 || EPI == OCN_EPI_F32) ? 4 : 2);
        t.d_aux = make_desc6(a.aux, m0, EPI == OCN_EPI_BIAS_GELU ? m_st : m_ld, a.ldc, 2);
        t.d_res = make_desc6(a.resid, m0, m_ld, a.ldc, 4);
        return t;
    };

#define MM6(ACC, P)                                              \
    ACC[0][0] = mfma32(fb[P][0], fa[P][0], ACC[0][0]);           \
    ACC[0][1] = mfma32(fb[P][1], fa[P][0], ACC[0][1]);           \
    ACC[1][0] = mfma32(fb[P][0], fa[P][1], ACC[1][0]);           \
    ACC[1][1] = mfma32(fb[P][1], fa[P][1], ACC[1][1]);
#define RD6(P, KS)                                               \
    {                                                            \
        const unsigned xa_ = va0[KS] + so_rd, xb_ = vb0[KS] + so_rd; \
        DSR6(fa[P][0], xa_, 0); DSR6(fa[P][1], xa_, 4096);       \
        DSR6(fb[P][0], xb_, 0); DSR6(fb[P][1], xb_, 4096);       \
    }
#define BND6(PARK, J, S)                                         \
    SB();                                                        \
    if constexpr (TRK) trk_bnd<EPI, J, S>(ctx, trk);             \
    LGKM0();                                                     \
    if constexpr (TRK) trk_aft<EPI, J, S>(ctx, trk);             \
    SB();
    // one phase = one K-tile.  ACCP: accumulators of the K-tile before it (its last 4 MFMAs run here), ACCC: this K-tile's, PARK: the
    // finished previous tile (trickled out in phases J = 0..6; J = 8: any later phase)
#define PHASE6(ACCP, ACCC, PARK, J)                                                                    \
    {                                                                                                  \
        if constexpr (TRK) trk_wait<EPI, J>(trk); else VMCNT(6);                                       \
        __builtin_amdgcn_s_barrier();                                                                  \
        SB();                                                                                          \
        if constexpr (TRK) { trk_pre<EPI, J>(ctx, PARK, trk); trk_vm<EPI, J>(ctx, PARK, trk); }         \
        SB();                                                                                          \
        RD6(0, 0)                                                                                      \
        SB();                                                                                          \
        ACCP[0][0] = mfma32(fb[1][0], fa[1][0], ACCP[0][0]);                                           \
        DMA_A(0, so_wr) DMA_A(1, so_wr)                                                                \
        ACCP[0][1] = mfma32(fb[1][1], fa[1][0], ACCP[0][1]);                                           \
        DMA_A(2, so_wr) DMA_A(3, so_wr)                                                                \
        ACCP[1][0] = mfma32(fb[1][0], fa[1][1], ACCP[1][0]);                                           \
        DMA_B(0, so_wr) DMA_B(1, so_wr)                                                                \
        ACCP[1][1] = mfma32(fb[1][1], fa[1][1], ACCP[1][1]);                                           \
        adv();                                                                                         \
        if constexpr (TRK && (J) >= 1) trk_slot<EPI, J, 0>(ctx, PARK, trk);                            \
        BND6(PARK, J, 0)                                                                               \
        RD6(1, 1)                                                                                      \
        SB();                                                                                          \
        MM6(ACCC, 0)                                                                                   \
        if constexpr (TRK) trk_slot<EPI, J, 1>(ctx, PARK, trk);                                        \
        BND6(PARK, J, 1)                                                                               \
        RD6(0, 2)                                                                                      \
        SB();                                                                                          \
        MM6(ACCC, 1)                                                                                   \
        if constexpr (TRK) trk_slot<EPI, J, 2>(ctx, PARK, trk);                                        \
        BND6(PARK, J, 2)                                                                               \
        RD6(1, 3)                                                                                      \
        SB();                                                                                          \
        MM6(ACCC, 0)                                                                                   \
        if constexpr (TRK) trk_slot<EPI, J, 3>(ctx, PARK, trk);                                        \
        so_rd = so_rd == 2 * SLOT6 ? 0u : so_rd + SLOT6;                                               \
        so_wr = so_wr == 2 * SLOT6 ? 0u : so_wr + SLOT6;                                               \
        BND6(PARK, J, 3)                                                                               \
    }

    // synchronous epilogue of a finished tile: every block start to finish, then everything drained (last tile of a workgroup; TRK = false)
#define EPI_SYNC6(ACC, I)                                                                              \
    {                                                                                                  \
        const Tile6 t_ = tile_desc(I);                                                                 \
        _Pragma("unroll") for (int ha_ = 0; ha_ < 2; ++ha_) _Pragma("unroll") for (int hb_ = 0; hb_ < 2; ++hb_)                \
            epi_block_sync<EPI>(a, t_, ACC[ha_][hb_], ha_, hb_, row_w, col_w, ctx.L, false);           \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
    }
    // one tile: phase 0 finishes the previous tile's accumulators (ACCP), which then stay parked and are written out under phases 1..6
#define TILE6(ACCP, ACCC, I)                                                                           \
    {                                                                                                  \
        int m0_, n0_;                                                                                  \
        tile_origin(I, m0_, n0_);                                                                      \
        init_acc(d_bias, ACCC, n0_, col_w, lh);                                                        \
        if constexpr (TRK) ctx.tp = tile_desc((I)-1);                                                  \
        PHASE6(ACCP, ACCC, ACCP, 0)                                                                    \
        if constexpr (MODE == 0) { if ((I) > 0) EPI_SYNC6(ACCP, (I)-1) }                                    \
        PHASE6(ACCC, ACCC, ACCP, 1)                                                                    \
        PHASE6(ACCC, ACCC, ACCP, 2)                                                                    \
        PHASE6(ACCC, ACCC, ACCP, 3)                                                                    \
        PHASE6(ACCC, ACCC, ACCP, 4)                                                                    \
        PHASE6(ACCC, ACCC, ACCP, 5)                                                                    \
        PHASE6(ACCC, ACCC, ACCP, 6)                                                                    \
        PHASE6(ACCC, ACCC, ACCP, 7)                                                                    \
        for (int kt = 8; kt < nk; ++kt) PHASE6(ACCC, ACCC, ACCP, 8)                                    \
    }

    for (int i = 0; i < my_tiles; i += 2) {
        TILE6(acc[1], acc[0], i)
        if (i + 1 < my_tiles) TILE6(acc[0], acc[1], i + 1)
    }
    // the last K-tile's final 4 MFMAs, then the last tile's epilogue (nothing left to hide it under)
    VMCNT(0);
    if (my_tiles & 1) {
        MM6(acc[0], 1)
        EPI_SYNC6(acc[0], my_tiles - 1)
    } else {
        MM6(acc[1], 1)
        EPI_SYNC6(acc[1], my_tiles - 1)
    }
#undef TILE6
#undef EPI_SYNC6
#undef PHASE6
#undef BND6
#undef RD6
#undef MM6
#undef DMA_A
#undef DMA_B
#undef DMA6
}

}  // namespace
extern int g_ocn_tuning[16];
namespace {

int g6_num_cu = 0;

// band width (in 128-column tiles) of the band-major tile walk; same L2 model as nt5_band (gemm_nt5.hip)
int nt6_band(int M, int N, int K, int forced) {
    const int tiles_n = ocn_cdiv(N, 128);
    if (forced > 0) return forced < tiles_n ? forced : tiles_n;
    const double panel = 128.0 * K * 2, Bt = (double)N * K * 2, At = (double)M * K * 2;
    const double l2_band = 2.5 * 1048576;
    if (Bt <= l2_band) return tiles_n;
    const double ntiles = (double)ocn_cdiv(M, 256) * tiles_n;
    const double row_major = At + (ntiles / 32.0 < 8.0 ? 8.0 : ntiles / 32.0) * Bt;
    for (int nb = 2; nb <= tiles_n; ++nb) {
        const int band = ocn_cdiv(tiles_n, nb);
        if (band * panel > l2_band) continue;
        return (nb * At + 8.0 * Bt < row_major) ? band : tiles_n;
    }
    return tiles_n;
}

template <int EPI, int MODE>
int launch6_aux(GemmNtArgs a, int grid, hipStream_t st) {
    static bool set_ = false;
    if (!set_) {
        (void)hipFuncSetAttribute((const void*)gemm_nt6_kernel<EPI, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS6);
        set_ = true;
    }
    hipLaunchKernelGGL((gemm_nt6_kernel<EPI, MODE>), dim3(grid), dim3(512), LDS6, st, a);
    OCN_CHECK_LAUNCH("ocn_gemm_nt");
    return OCN_OK;
}

template <int EPI>
int launch6(GemmNtArgs a, hipStream_t st) {
    if (g6_num_cu == 0) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g6_num_cu = n;
    }
    a.tiles_n = a.N / 128;
    a.ntiles = ocn_cdiv(a.M, 256) * a.tiles_n;
    a.band = nt6_band(a.M, a.N, a.K, (a.ablate >> 8) & 31);
    a.stagger = 0;
    a.first_wave = g6_num_cu;
    const int grid = a.ntiles < g6_num_cu ? a.ntiles : g6_num_cu;
    // developer knob 13 = 1: the synchronous-epilogue build of the same main loop (A/B of what the trickle buys)
    if (g_ocn_tuning[13] == 1) return launch6_aux<EPI, 0>(a, grid, st);
    if (g_ocn_tuning[13] == 2) return launch6_aux<EPI, 2>(a, grid, st);
    return launch6_aux<EPI, 1>(a, grid, st);
}

}  // namespace

int ocn_launch_nt6(int epilogue, const GemmNtArgs& a, hipStream_t st) {
    // whole 128-column tiles (bias / tile descriptors are not clipped in N), 64-wide K-tiles, alpha folded nowhere
    // (the parked tile leaves during the first 7 K-tiles of its successor: K >= 512)
    if (a.K % 64 != 0 || a.K < 512 || a.N % 128 != 0 || a.ldc % 8 != 0 || a.alpha != 1.0f) return 1;
    if ((long)a.ldc * 4 * 256 >= 0x7fffffffL || (long)a.lda * 2 * 256 >= 0x7fffffffL) return 1;
    switch (epilogue) {
        case OCN_EPI_BF16: return launch6<OCN_EPI_BF16>(a, st);
        case OCN_EPI_BIAS_GELU: return launch6<OCN_EPI_BIAS_GELU>(a, st);
        case OCN_EPI_BIAS_RESID_F32: return launch6<OCN_EPI_BIAS_RESID_F32>(a, st);
        case OCN_EPI_DGELU: return launch6<OCN_EPI_DGELU>(a, st);
        case OCN_EPI_F32: return launch6<OCN_EPI_F32>(a, st);
    }
    return 1;
}
