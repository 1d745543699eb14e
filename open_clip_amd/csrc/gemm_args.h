// Argument block shared by the NT GEMM kernels (gemm.hip, gemm_nt5.hip).
#pragma once
#include "ocn_common.h"

struct GemmNtArgs {
    const bf16* A;
    const bf16* B;
    void* out;
    const float* bias;
    const void* resid;  // fp32 (OCN_EPI_BIAS_RESID_F32) or bf16 (OCN_EPI_BIAS_RESID_BF16) [M, ldc]
    unsigned char* aux;  // saved gelu' in 8 bits (ocn_common.h: dgelu_pack4 / dgelu_unpack4), [M, ldc] bytes
    int lda, ldb, ldc, M, N, K;
    float alpha;
    int tiles_n, ntiles;
    int stagger; // start offset between the 4 phase classes of workgroups, in 100 MHz ticks (0 = none; gemm_nt5.hip)
    int first_wave;  // workgroups [0, first_wave) start together (one per CU) and are the ones that get staggered
    int stagger_mode;  // which workgroups share a phase class (gemm_nt5.hip; 0 = shipped; other values: developer build only, ocn_set_tuning key 3)
    int band;    // tile walk order: column bands of `band` n-tiles, row-major inside a band (gemm_nt5.hip)
    // the launch's last, partial round of tiles computed as HALF tiles (gemm_nt5.hip; set by launch5, 0 = off): tiles [tail_first, tail_first +
    // tail_n) of the walk are split; workgroup j < tail_n takes the upper half of tail tile j, workgroup tail_partner + j the lower half
    int tail_first, tail_n, tail_partner;
    int ablate;  // developer ablation mask (tools/gemm_bench.py)
    // split-K (gemm_nt5.hip; ocn_gemm_nt_splitk): K in `ksplit` slices, slice s of every output tile is a tile of its own and writes its fp32 partial sum
    // to out + s * ws_stride elements (out = the caller's workspace of ksplit slabs); 0 / 1 = off
    int ksplit;
    long ws_stride;
    int* rescue;  // tile-rescue board of this launch (core.hip::ocn_rescue_board): [w] = counter of workgroup w's share, zero at launch; null = off
    // fused logits + cross-entropy epilogue (OCN_EPI_CE_ONEPASS below; ocn_fused_logits_ce in loss.hip)
    float* ce_stats;        // [M][ce_parts][2]: per row and 64-column strip (sum e, sum e * logit), e = exp(logit - shift)
    float* ce_label_logit;  // [M] the logit of the row's label column (host side of the launch only)
    const float* ce_lse;    // (unused since round 6)
    const float* ce_shift2; // [M] the row's shift c_i * log2(e)
    float* ce_dscale;       // (unused since round 6)
    int ce_parts, ce_label_offset;
    float ce_grad_scale;
};
// internal epilogues of the persistent NT kernel (not part of enum ocn_epilogue): the logits tile never leaves the registers
enum { OCN_EPI_CE_ONEPASS = 9, OCN_EPI_CE_ONEPASS_FULL = 10 };  // _FULL: M, N multiples of 256, no masks  // (5, 6: the two-pass statistics / gradient epilogues of rounds 2-5, removed)

// chunk swizzle for 128-byte LDS rows: bijection on 3 bits built from row bits 1..3, chosen so that
// (a) the four 16-lane groups of a ds_read_b128 fragment read hit 16 distinct 16-byte slots and
// (b) 4 consecutive rows land in different 64-byte quarters (needed by tr16 reads of the same image).
OCN_DEV int swz_nt(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// persistent 256x256 NT kernel (gemm_nt5.hip); returns 1 if the shape is not supported by it (caller falls back)
int ocn_launch_nt5(int epilogue, const GemmNtArgs& a, hipStream_t st);

struct GemmTnArgs {
    const bf16* A;
    const bf16* B;
    float* dW;
    float* dbias;
    int lda, ldb, ldw, M, N, K;
    float alpha;
    int tiles_n, tiles_k, chunk, nwg;
    int ablate;  // developer knob (ocn_set_tuning key 4): 1 = skip the atomic epilogue (timing only)
    float* ws;   // optional workspace [splits][N][K] (+ [splits][N] for dbias behind it): partial tiles instead of atomics (gemm_tn5.hip)
    int nsplit;
    // optional SECOND problem of the same launch (gemm_tn5.hip, ocn_gemm_tn_accum2): dW2[N2,K] += alpha * A2[M,N2]^T . B2[M,K] over the
    // same M rows and the same K; its tiles follow the first problem's in the tile index space (ntile1 = tiles of problem 1; 0 = none)
    const bf16* A2;
    const bf16* B2;
    float* dW2;
    float* dbias2;
    int lda2, ldb2, ldw2, N2, ntile1, ntile_all;
    int* rescue;  // tile-rescue board (as GemmNtArgs::rescue); a share = the `pieces` sub-chunks of a workgroup's M-chunk
    int pieces, piece_rows;
};

// tile rescue (core.hip): a zeroed board for ONE launch on `st` when ocn_set_tile_rescue(1) is in force and the launch has at most OCN_RESCUE_SLOTS workgroups, else null
// counters per board (= most workgroups of a launch), boards per stream, ints from one counter to the next (measured: one counter per 128-byte line,
// stride 32, changes nothing -- the owners' claims are hidden behind their prologues either way -- and makes the finishers' scan 32 x as wide)
constexpr int OCN_RESCUE_SLOTS = 1024, OCN_RESCUE_RING = 512, OCN_RESCUE_STRIDE = 1;
int* ocn_rescue_board(hipStream_t st, int workgroups);

// hand-scheduled 256x256 TN (wgrad) kernel (gemm_tn5.hip); returns 1 if the shape is not supported by it (caller falls back)
int ocn_launch_tn5(GemmTnArgs a, hipStream_t st);
// two wgrads over the same rows in ONE launch (a.A2 ... a.N2 set by the caller); returns 1 if the pair does not fit the kernel
int ocn_launch_tn5_pair(GemmTnArgs a, hipStream_t st);
// bytes of workspace with which ocn_launch_tn5 replaces its atomic epilogue by partial tiles + a reduce pass (0: atomics are fine)
long ocn_tn5_workspace_bytes(int M, int N, int K);
