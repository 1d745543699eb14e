/*
 * openclip_hip.h -- C ABI of libopenclip_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * CLIP training hot path of mlfoundations/open_clip.
 *
 * The reference has no native code: its device work is a set of ATen / c10d call sites
 * (SURVEY.md 2.3, K1..K14).  Every entry point below names the reference call site it replaces
 * (file:line under /root/reference/src/open_clip).  A reference-side binding is plain ctypes
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers + sizes; the caller owns all memory (device pointers unless stated otherwise);
 *     the library never allocates.
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it (async).
 *   - re-entrant, no global device state (the backward pass runs on the autograd thread).  The only process-global state in the
 *     library are the developer knobs of include/openclip_hip_debug.h (kernel selection / ablation switches for experiments:
 *     never needed, never touched by the product path, defaults = the shipped behaviour).
 *   - return 0 on success, negative ocn_status on error; ocn_last_error() holds the message
 *     (thread-local).  Python wrappers turn that into RuntimeError, like the reference's asserts.
 *   - "bf16" buffers are raw 16-bit bfloat16; residual stream, statistics, losses and weight
 *     gradients are fp32.
 */
#ifndef OPENCLIP_HIP_H
#define OPENCLIP_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ocn_stream_t; /* hipStream_t */

enum ocn_status { OCN_OK = 0, OCN_ERR_INVALID = -1, OCN_ERR_LAUNCH = -2, OCN_ERR_UNSUPPORTED = -3 };

/* epilogues of ocn_gemm_nt */
enum ocn_epilogue {
    OCN_EPI_BF16 = 0,           /* out_bf16 = alpha*acc + bias                                            */
    OCN_EPI_BIAS_GELU = 1,      /* out_bf16 = gelu(acc+bias); aux_u8 = gelu'(acc+bias) in 8-bit fixed point (for EPI 3) */
    OCN_EPI_BIAS_RESID_F32 = 2, /* out_f32 = resid_f32 + acc + bias                                       */
    OCN_EPI_DGELU = 3,          /* out_bf16 = acc * decode(aux_u8)   (aux = the gelu' saved by EPI 1)      */
    OCN_EPI_F32 = 4,            /* out_f32 = alpha*acc + bias                                             */
    OCN_EPI_BIAS_QUICKGELU = 7, /* EPI 1 with QuickGELU, x * sigmoid(1.702 x) (layers.py:29-32; `quick_gelu` configs); (9: internal) */
    OCN_EPI_BIAS_RESID_BF16 = 8 /* out_bf16 = bf16(resid_bf16 + bf16(acc + bias)): the residual add on a bf16 stream exactly as the reference's
                                   autocast evaluates it (F.linear's bf16 result, then `q_x + ...` in bf16: transformer.py:328-329) */
};

const char* ocn_last_error(void);
/* ABI version of this header: ocn_version() of the library that is loaded must EQUAL it (open_clip_amd/_lib.py::load and
 * __graft_entry__.build() check; a stale .so behind OCN_LIB_PATH would otherwise receive shifted arguments without any error).
 *   102 (round 5)  ocn_sumsq_multi takes a per-chunk workspace (reproducible sum); ocn_siglip_rows takes the bias as a device value.  BREAKING since 101 and now carried by the number:
 *                  ocn_layernorm_bwd / ocn_embed_assemble_bwd / ocn_token_embed_bwd_sorted[_varlen] / ocn_softmax_ce_rows / ocn_siglip_rows took
 *                  extra arguments in round 4, and the logit-gradient matrix G written by ocn_softmax_ce_rows / ocn_fused_logits_ce /
 *                  ocn_siglip_rows holds softmax (sigmoid) * grad_scale WITHOUT the -onehot term (the caller applies it as an exact rank-1
 *                  update: open_clip_amd/loss.py::_PairTerm.dX / dY). */
/*   103 (round 6)  bf16 residual stream of the image tower: ocn_layernorm_fwd / ocn_layernorm_bwd / ocn_gather_rows take dtype flags, ocn_gemm_nt
 *                  takes `resid` as void* (fp32 or bf16 by epilogue) and knows OCN_EPI_BIAS_RESID_BF16; new: ocn_comm_count, ocn_comm_sendrecv; ocn_fused_logits_ce is ONE
 *                  pass now: G holds exp(logit - shift), the row scale comes back in `rowscale` (new argument); new: ocn_gemm_nt_splitk[_plan],
 *                  ocn_scale_rows_bf16, ocn_sub_scaled_rows.
 *   104 (round 6)  new: ocn_set_tile_rescue / ocn_get_tile_rescue (no signature changed). */
#define OCN_ABI_VERSION 104
int ocn_version(void);

/* ---- GEMMs (MFMA v_mfma_f32_32x32x16_bf16, fp32 accumulate) ------------------------------------
 * ocn_gemm_nt: C[M,N] = A[M,K] . B[N,K]^T with a fused epilogue.  Replaces F.linear
 *   (transformer.py:169 in_proj, :246 out_proj, :295-299 c_fc + nn.GELU + c_proj), the residual adds of
 *   transformer.py:328-329, `pooled @ proj` (:923), `x @ text_projection` (model.py:409), the logit
 *   matmul (loss.py:103-110) and every dgrad of the backward (a17).  K % 32 == 0; A, B bf16 row-major.
 *   bias [N] fp32 or NULL; resid fp32 [M,ldc] (EPI 2) or bf16 [M,ldc] (EPI 8); aux uint8 [M,ldc] (EPI 1: written, EPI 3: read): the GELU derivative, the only
 *   thing the backward needs of the pre-activation, as q = round((gelu' + 0.13) * 200) in [0, 252] -- gelu' lies in [-0.129, 1.129], so
 *   the decoded value q / 200 - 0.13 is within 0.0025 of it (unbiased; rms 0.0014, what rounding a value in [0.5, 1) to bf16 costs) at
 *   half the bytes of a bf16 copy. */
int ocn_gemm_nt(int epilogue, const void* A, int lda, const void* B, int ldb, void* out, int ldc, int M, int N, int K,
                const float* bias, const void* resid, void* aux, float alpha, ocn_stream_t stream);

/* ocn_gemm_tn_accum: dW[N,K] += alpha * A[M,N]^T . B[M,K]  (fp32 atomics into dW, which the caller zeroes or
 *   pre-loads); if dbias != NULL also dbias[N] += alpha * colsum(A).  The wgrad + bias-grad of every Linear
 *   (autograd of transformer.py:169,246,295-299) and the G^T products of the loss backward.
 *   N % 8 == 0, K % 8 == 0; M arbitrary. */
int ocn_gemm_tn_accum(const void* A, int lda, const void* B, int ldb, float* dW, int ldw, int M, int N, int K,
                      float* dbias, float alpha, ocn_stream_t stream);

/* Two weight gradients over the SAME M rows and the same K in ONE launch (both bias gradients or neither):
 *   dW1[N1,K] += alpha * A1[M,N1]^T . B1[M,K],  dW2[N2,K] += alpha * A2[M,N2]^T . B2[M,K].
 * The out-proj and QKV weight gradients of a residual block (autograd of transformer.py:169,246) are such a pair: alone, the small one
 * needs 28-64 M-splits to fill the chip and spends 26-37 % of its time in contended atomics; paired they share 7 splits.  Shapes the
 * paired kernel does not take are executed as two ocn_gemm_tn_accum calls (same results up to fp32 summation order). */
int ocn_gemm_tn_accum2(const void* A1, int lda1, const void* B1, int ldb1, float* dW1, int ldw1, float* dbias1, int N1,
                       const void* A2, int lda2, const void* B2, int ldb2, float* dW2, int ldw2, float* dbias2, int N2,
                       int M, int K, float alpha, ocn_stream_t stream);

/* Split-K form of ocn_gemm_nt for products with FEW output tiles and a LONG K (the loss's `G @ T` of loss.py:103-110's backward: [4096, 512] from
 * K = 32768 has 32 tiles for 256 CUs): K is cut into `ksplit` slices, every slice of every 256 x 256 tile is a tile of the persistent kernel's walk and
 * leaves its fp32 partial sum in slab s of `workspace` (ksplit * M * ldc floats); a second kernel forms
 *   out[m, n] = scale * (rowscale[m] * sum_s ws[s][m][n] - sub_alpha * sub_rows[m][n])
 * (rowscale fp32 [M], sub_rows bf16 [M, ld_sub], scale_dev a 1-element DEVICE value: each may be NULL) -- the row scale of the one-pass cross-entropy,
 * the exact -onehot part of the logit gradient and logit_scale, which the caller would otherwise apply in four more passes over out.
 * ocn_gemm_nt_splitk_plan: the ksplit that fills the chip (1 = no split: use ocn_gemm_nt).  K % (128 * ksplit) == 0. */
int ocn_gemm_nt_splitk_plan(int M, int N, int K);
int ocn_gemm_nt_splitk(const void* A, int lda, const void* B, int ldb, float* out, int ldc, int M, int N, int K, int ksplit, float* workspace,
                       const float* rowscale, const void* sub_rows_bf16, int ld_sub, float sub_alpha, const float* scale_dev, ocn_stream_t stream);

/* Run-to-run reproducible form of ocn_gemm_tn_accum (same arguments, same result up to fp32 summation ORDER, which here is fixed): no
 * two workgroups add into one address.  Shapes the hand-scheduled kernel takes (ocn_gemm_tn_det_workspace_bytes(M, N, K) > 0) need that
 * many bytes of 16-byte aligned scratch: every M-split stores its partial dW tile (and bias row) to its own slab and a second kernel sums
 * the slabs in split order; other shapes run the general kernel with one M-split and need no scratch.  The reference's counterpart is
 * torch.use_deterministic_algorithms for its own GEMMs; cost here: one extra pass over splits x N x K floats per launch. */
int64_t ocn_gemm_tn_det_workspace_bytes(int M, int N, int K);
int ocn_gemm_tn_accum_det(const void* A, int lda, const void* B, int ldb, float* dW, int ldw, int M, int N, int K, float* dbias,
                          float alpha, void* workspace, int64_t workspace_bytes, ocn_stream_t stream);

/* ---- casts -------------------------------------------------------------------------------------
 * amp_bf16 policy (precision.py:6-16): fp32 master weights, bf16 GEMM operands. */
int ocn_cast_f32_bf16(const float* src, void* dst, int64_t n, ocn_stream_t stream);
/* dst = bf16(src * *scale_dev): the scalar is read on the device, so a caller that holds it in a tensor (logit_scale.exp(),
 * loss.py:103-110) never has to synchronise the host to pass it */
int ocn_cast_f32_bf16_scaled(const float* src, void* dst, int64_t n, const float* scale_dev, ocn_stream_t stream);
/* dst[C,R] (bf16) = transpose(src[R,C] fp32): weight copies laid out for the NT dgrad / `@ proj` GEMMs */
int ocn_cast_transpose_f32_bf16(const float* src, void* dst, int R, int C, ocn_stream_t stream);

/* ---- LayerNorm (layers.py:20-26, eps 1e-5; fp32 statistics) ------------------------------------
 * fwd: y = (x-mean)*rstd*w + b over the last dim C (C % 4 == 0, C <= 2048); writes y as bf16 and/or fp32
 *      (either pointer may be NULL) and mean/rstd [M] for the backward.
 * bwd: dx = LN'(dy) (+ dres if non-NULL: the residual branch's gradient), written as fp32 and/or bf16;
 *      dw[C] += sum_rows dy*xhat, db[C] += sum_rows dy (fp32 atomics; caller zeroes).
 *      dy is bf16 (dy_is_f32 = 0) or fp32 (1).
 *      dcol (may be NULL): dcol[C] += sum_rows dx in fp32, before dx is rounded to bf16.  dx is the gradient of the output of the linear
 *      in front of this LayerNorm's input (out_proj, or the previous block's c_proj: transformer.py:246, :299), so this is that layer's bias
 *      gradient summed from fp32 values instead of from the bf16 operand of its weight-gradient GEMM.
 * ocn_colsum_f32: out[C] += sum_rows x[R, C] (fp32; the same for tensors no LayerNorm backward produces).
 * bf16 residual stream (round 6; the IMAGE tower as the reference's autocast runs it: transformer.py:794 conv1 under autocast -> bf16,
 *      layers.py:23-26 `x.to(orig_type)`): x is fp32 (x_is_bf16 = 0) or bf16 (1); in the backward a bf16 x takes a bf16 dy, no dcol, and the
 *      residual gradient dres as bf16 (dres_is_bf16 = 1: the gradient of a bf16 tensor is bf16 under autograd) or fp32.  Statistics and
 *      arithmetic are fp32 in every form. */
int ocn_layernorm_fwd(const void* x, int x_is_bf16, const float* w, const float* b, void* y_bf16, float* y_f32, float* mean,
                      float* rstd, int M, int C, float eps, ocn_stream_t stream);
int ocn_layernorm_bwd(const void* dy, int dy_is_f32, const void* x, int x_is_bf16, const float* w, const float* mean,
                      const float* rstd, const void* dres, int dres_is_bf16, float* dx_f32, void* dx_bf16, float* dw, float* db, float* dcol,
                      float* det_workspace, int M, int C, ocn_stream_t stream);
/* det_workspace (may be NULL = fp32 atomics): ocn_layernorm_bwd_det_workspace_floats(M, C) floats; the workgroups' partial dw / db / dcol rows go to
 * their own slabs and a second kernel adds them in workgroup order: bit-reproducible from run to run (torch.use_deterministic_algorithms) */
int64_t ocn_layernorm_bwd_det_workspace_floats(int M, int C);
int ocn_colsum_f32(const float* x, float* out, int R, int C, int deterministic, ocn_stream_t stream);
/* ---- attention core (transformer.py:199-244: head split + F.scaled_dot_product_attention) ------
 * qkv bf16 [B*L, 3*H*64] (q | k | v column blocks, heads contiguous inside each: the layout F.linear with
 * in_proj_weight produces, transformer.py:169); out bf16 [B*L, H*64]; lse fp32 [B*H*L] (natural log).
 * head_dim is 64; L <= 320.  causal != 0 applies the text tower's upper-triangular -inf mask
 * (transformer.py:1716-1722) as a predicate. */
int ocn_attn_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int causal, float scale,
                 ocn_stream_t stream);
int ocn_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int L, int H,
                 int causal, float scale, ocn_stream_t stream);
/* The same with an explicit head_dim (64, 80, 88, 96, 104, 112 or 128; qkv [B*L, 3*H*head_dim]): head_dim 64 up to 128 tokens (forward)
 * / 320 tokens (backward) runs the head-resident kernels above, everything else (ViT-H-14: head_dim 80, 257 tokens; ViT-L-14's 257 tokens;
 * ViT-g / bigG / e: head_dim 88 / 104 / 112) the STREAMED kernels of csrc/attention_generic.hip: 4-wave workgroups that own four 32-row
 * blocks and stream the operand all of them need through a two-slot LDS ring in 64-row chunks (18-35 KB of LDS per workgroup, any
 * sequence length).  The backward's workspace `delta_ws` is fp32 [B*H*L] (sum_d dO*O, exchanged between its launches; may be NULL on the
 * head-resident path). */
int ocn_attn_fwd_hd(const void* qkv, void* out, float* lse, int B, int L, int H, int head_dim, int causal, float scale,
                    ocn_stream_t stream);
int ocn_attn_bwd_hd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta_ws, int B, int L,
                    int H, int head_dim, int causal, float scale, ocn_stream_t stream);
/* Packed ("varlen") batches, head_dim 64.  The text tower pools x[b, argmax(text[b])] (transformer.py:941-944) under the causal
 * mask (:1716-1722), so the tokens behind the pooled one cannot influence the feature or any gradient: the native text tower keeps
 * only the first eot[b]+1 tokens of each sequence.  Sequence b owns rows seq_off[b] .. seq_off[b+1] of qkv / out / dout / dqkv
 * (seq_off: B+1 ascending int32 on the device, 1 <= length <= Lmax <= 320); lse keeps the dense fp32 [B,H,Lmax] layout.
 * Bucketing (optional; both pointers or neither): `order` = the B sequence ids grouped by ceil(length / 32) ascending (DEVICE, from
 * ocn_seq_bucket_plan), `bucket_counts` = how many sequences each group holds (HOST array of ceil(Lmax / 32) ints summing to B).  One
 * launch per non-empty group, its workgroups sized for that group's length instead of Lmax (these kernels are latency-bound: their
 * rate is the number of resident workgroups); results do not depend on the bucketing. */
int ocn_attn_fwd_varlen(const void* qkv, void* out, float* lse, const int32_t* seq_off, const int32_t* order, const int32_t* bucket_counts,
                        int B, int Lmax, int H, int causal, float scale, ocn_stream_t stream);
int ocn_attn_bwd_varlen(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, const int32_t* seq_off,
                        const int32_t* order, const int32_t* bucket_counts, int B, int Lmax, int H, int causal, float scale,
                        ocn_stream_t stream);

/* Single-query attention of a tower's LAST block, head_dim 64 (csrc/attention_pooled.hip).  Both poolers read one row per sequence of the
 * last block's output (transformer.py:829-831 `x[:, 0]`, :941-944 `x[arange, text.argmax(-1)]`), and rows only mix inside the attention: of
 * that block's attention exactly one query per (sequence, head) is needed.  q / out / dout / dq are [B, H*64] (the pooled rows), kv / dkv
 * [M, 2*H*64] (K | V column blocks of EVERY row: F.linear with in_proj_weight[C:], transformer.py:169), lse fp32 [B*H].  Sequence b owns rows
 * seq_off[b] .. seq_off[b+1] of kv (packed text rows) or, with seq_off = NULL, rows b*L .. (b+1)*L; rows[b] = absolute row of its pooled
 * token, which under `causal` sees the keys up to and including its own row.  The backward writes every key row of dkv (zeros behind the
 * pooled row). */
int ocn_attn_pooled_fwd(const void* q, const void* kv, void* out, float* lse, const int32_t* seq_off, const int32_t* rows, int B, int L,
                        int H, int causal, float scale, ocn_stream_t stream);
int ocn_attn_pooled_bwd(const void* q, const void* kv, const void* out, const void* dout, const float* lse, void* dq, void* dkv,
                        const int32_t* seq_off, const int32_t* rows, int B, int L, int H, int causal, float scale, ocn_stream_t stream);

/* ---- image tower embedding (transformer.py:793-808) -------------------------------------------
 * patchify: image [B,3,H,W] (fp32, or bf16 when image_is_bf16) -> patches bf16 [B*gh*gw, Kpad], column order
 *   (c, i, j) = conv1.weight.reshape(width, 3*P*P) (transformer.py:632-638, :794-796); zero-padded to Kpad.
 * embed_assemble_fwd: emb[b,0,:] = cls + pos[0]; emb[b,1+g,:] = patch_out[b*G+g,:] + pos[1+g]  (:799-801)
 * embed_assemble_bwd: dpatch bf16 [B*G, C] = demb[b,1+g,:]; dpos[T,C] += sum_b demb; dcls[C] += sum_b demb[b,0] */
int ocn_patchify(const void* image, int image_is_bf16, void* patches, int B, int H, int W, int P, int Kpad,
                 ocn_stream_t stream);
/* uint8 input path (8f-4): pixels as decoded ([B,H,W,3] when hwc, else [B,3,H,W]); the kernel applies ToTensor + Normalize
 * ((x/255 - mean[c]) / std[c]; mean3 / std3 = HOST pointers to 3 floats, src/open_clip/constants.py:1-2) while building the
 * same bf16 patch matrix as ocn_patchify. */
int ocn_patchify_u8(const void* image_u8, int hwc, const float* mean3, const float* std3, void* patches, int B, int H, int W,
                    int P, int Kpad, ocn_stream_t stream);
int ocn_embed_assemble_fwd(const float* patch_out, const float* cls, const float* pos, float* emb, int B, int G, int C,
                           ocn_stream_t stream);
int ocn_embed_assemble_bwd(const float* demb, void* dpatch_bf16, float* dpos, float* dcls, int B, int G, int C,
                           int deterministic, ocn_stream_t stream);

/* ---- text tower embedding (model.py:399-401) ---------------------------------------------------
 * fwd: x[b,l,:] = table[text[b,l],:] + pos[l,:];  bwd: dtable[text[b,l],:] += dx[b,l,:] (fp32 atomics),
 * dpos[l,:] += sum_b dx[b,l,:].  text is int64. */
int ocn_token_embed_fwd(const int64_t* text, const float* table, const float* pos, float* x, int B, int L, int C,
                        int vocab, ocn_stream_t stream);
int ocn_token_embed_bwd(const int64_t* text, const float* dx, float* dtable, float* dpos, int B, int L, int C, int vocab,
                        ocn_stream_t stream);

/* The same backward from SORTED ids (no per-occurrence atomics): sorted_tokens = the B*L token ids in ascending order, order[i] = the flat
 * row (b*L + l) of dx that sorted_tokens[i] came from (any stable or unstable sort; torch.sort on the device).  dtable must arrive ZEROED
 * (complete runs are stored, not added); dpos is accumulated into.  dx is fp32 or (dx_is_bf16) bf16 [B*L, C].
 * deterministic != 0 (here, in ocn_embed_assemble_bwd and in the packed form below): the reproducible form -- every run of equal ids is summed
 * by ONE workgroup in sorted order (needs a STABLE sort), dpos / dcls by a single writer per element in batch order: no fp32 atomics, bit-identical
 * from run to run (torch.use_deterministic_algorithms); long runs (SOT / EOT, the zero padding of a dense batch) serialise. */
int ocn_token_embed_bwd_sorted(const int64_t* sorted_tokens, const int64_t* order, const void* dx, int dx_is_bf16, float* dtable, float* dpos,
                               int B, int L, int C, int vocab, int deterministic, ocn_stream_t stream);

/* ---- packed text batches (see ocn_attn_fwd_varlen) -----------------------------------------------
 * seq_pack_plan: eot[b] = argmax(text[b,:]); seq_off = exclusive scan of (eot+1) (B+1 entries, seq_off[B] = packed row count M);
 *   last_row[b] = seq_off[b+1]-1 (the pooled row).  seq_pack_rows: tokens[seq_off[b]+l] = text[b,l], posidx[..] = l for l <= eot[b].
 * token_embed_fwd_rows: x[r,:] = table[tokens[r]] + pos[posidx[r]] (fp32 [M,C]).
 * token_embed_bwd_sorted_varlen: ocn_token_embed_bwd_sorted over the M packed rows (sorted_tokens / order index packed rows). */
int ocn_seq_pack_plan(const int64_t* text, int32_t* eot, int32_t* seq_off, int32_t* last_row, int B, int L, ocn_stream_t stream);
/* *bad_count = number of ids outside [0, vocab) among text[0..n) (one workgroup, no atomics; written, not accumulated).  The embedding
 * kernels clamp such ids; nn.Embedding (src/open_clip/model.py:399) raises on them -- the host raises from this count. */
int ocn_token_range_check(const int64_t* text, long n, int vocab, int32_t* bad_count, ocn_stream_t stream);
/* order[0..B) = sequence ids grouped by ceil(length / 32) ascending, counts[k] = number of sequences with ceil(length / 32) == k + 1
 * (ceil(Lmax / 32) <= 16 entries, device): the buckets of ocn_attn_{fwd,bwd}_varlen.  The order inside a bucket is unspecified. */
int ocn_seq_bucket_plan(const int32_t* seq_off, int32_t* order, int32_t* counts, int B, int Lmax, ocn_stream_t stream);
int ocn_seq_pack_rows(const int64_t* text, const int32_t* seq_off, int64_t* tokens, int32_t* posidx, int B, int L, ocn_stream_t stream);
int ocn_token_embed_fwd_rows(const int64_t* tokens, const int32_t* posidx, const float* table, const float* pos, float* x, long M, int C,
                             int vocab, ocn_stream_t stream);
int ocn_token_embed_bwd_sorted_varlen(const int64_t* sorted_tokens, const int64_t* order, const void* dx, int dx_is_bf16, float* dtable,
                                      float* dpos, const int32_t* seq_off, int B, int L, long M, int C, int vocab, int deterministic, ocn_stream_t stream);

/* ---- pooling (transformer.py:786-787 'tok'; :941-944 'argmax') ---------------------------------
 * argmax_rows: idx[b] = first index of max(text[b,:]) (torch.argmax semantics)
 * gather_rows: out[b,:] = x[(b*L + idx[b]),:] (idx NULL -> token 0);  scatter_rows: dx (pre-zeroed)[b*L+idx[b],:] = d[b,:]
 *   (L = 0: idx holds absolute row numbers -- the packed text tower's last_row) */
int ocn_argmax_rows(const int64_t* text, int32_t* idx, int B, int L, ocn_stream_t stream);
int ocn_gather_rows(const void* x, int x_is_bf16, const int32_t* idx, float* out, int B, int L, int C, ocn_stream_t stream); /* x fp32 or bf16; out fp32 */
int ocn_gather_rows_bf16(const void* x, const int32_t* idx, void* out, int B, int L, int C, ocn_stream_t stream); /* bf16 x / out, C % 8 == 0 */
int ocn_scatter_rows(const float* d, const int32_t* idx, float* dx, void* dx_bf16, int B, int L, int C,
                     ocn_stream_t stream);
/* dx[row_b] += d[b] (fp32), dx_bf16[row_b] = bf16 of the sum (may be NULL): the pooled rows' share of the last block's input gradient added
 * into the all-row LayerNorm backward's result instead of travelling through a zero [M, C] residual-gradient matrix.  dx = NULL (bf16
 * gradient stream): dx_bf16[row_b] = bf16(dx_bf16[row_b] + d[b]). */
int ocn_scatter_add_rows(const float* d, const int32_t* idx, float* dx, void* dx_bf16, int B, int L, int C, ocn_stream_t stream);

/* ---- F.normalize (model.py:391,411; eps 1e-12) -------------------------------------------------
 * fwd: y = x / max(||x||, eps) as fp32 and bf16, inv_norm[B] saved; bwd: dx = (dy - y*(y.dy)) * inv_norm */
int ocn_l2norm_fwd(const float* x, float* y, void* y_bf16, float* inv_norm, int B, int E, float eps, ocn_stream_t stream);
int ocn_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int B, int E, ocn_stream_t stream);

/* ---- contrastive losses on materialised logits -------------------------------------------------
 * softmax CE rows (F.cross_entropy(logits, arange+offset), loss.py:78-89,136-139): for each of R rows of
 *   logits fp32 [R,N]: lse, loss_sum += (lse - logits[r, r+label_offset]) * loss_scale; writes
 *   G bf16 [R,N] = softmax * grad_scale -- the -onehot * grad_scale part of the logit gradient is NOT stored: the caller applies it exactly in
 *   fp32 (dX_r -= grad_scale * Y[r + label_offset], dY[r + label_offset] -= grad_scale * X_r): rounded to bf16 the label entry
 *   (p - 1) * grad_scale loses its p, a bias of the same sign in every row that every sum over the batch adds up coherently;
 *   dscale_sum += sum((softmax - onehot) * grad_scale * logits) * inv_logit_scale.
 * siglip (loss.py:344-367): z = labels*logits (labels -1 off-diagonal, +1 on (r, r+label_offset) unless
 *   negative_only); loss_sum += -logsigmoid(z)*loss_scale; with g = -labels*sigmoid(-z) = sigmoid(logits) - [positive]: G =
 *   sigmoid(logits)*grad_scale (the positives' -grad_scale is the caller's, exact, as for the cross-entropy);
 *   dscale_sum += sum(g*grad_scale*(logits-bias))*inv_logit_scale; dbias_sum += sum(g*grad_scale). */
int ocn_softmax_ce_rows(const float* logits, int ld, void* G, int ldg, int R, int N, int label_offset, float loss_scale,
                        float grad_scale, float inv_logit_scale, float* loss_sum, float* dscale_sum, float* det_rows,
                        ocn_stream_t stream);
/* det_rows (may be NULL = fp32 atomics into the three sums): fp32 [R, 3]; row r's contributions to loss_sum / dscale_sum / dbias_sum are
 * written there instead, for the caller to add up in a fixed order (ocn_colsum_f32(..., deterministic = 1)): the reproducible form */
/* The same cross-entropy WITHOUT materialised logits (loss.py:103-110 + :136-139 for a [R, N] block of logits_per_image / _per_text;
 * the row-sharded global loss of 8 GPUs has R = 4096, N = 32768): X bf16 [R, E] (already times logit_scale), Y bf16 [N, E].  ONE pass of the MFMA
 * GEMM (round 6; two until round 5): its epilogue consumes the fp32 logits tile in registers and writes G' = exp(logit - c_r) as bf16 [R, ldg] with a
 * per-row shift c_r fixed before the GEMM (the row's label logit, clamped from below so that nothing can overflow), then
 *   rowscale[r] = grad_scale / sum_j G'_rj  (fp32 [R]):  the logit gradient is  G_rj = G'_rj * rowscale[r] - [j = r + label_offset] * grad_scale -- the row
 *   scale is applied by the caller where it is free (on the [R, E] result of G' @ Y, on the [R, E] operand of G'^T @ X), the onehot part exactly as above;
 *   loss_sum += sum_r (lse_r - logit[r, r + label_offset]) * loss_scale;  dscale_sum += sum((softmax - onehot) * grad_scale * logits) (divide by
 *   logit_scale for d/d logit_scale).
 * Exact for any scale: rows whose shifted sums under- / overflow (impossible for unit vectors and logit_scale <= 78, practically for <= 150) are
 * redone from the operands with their true maximum.  E % 128 == 0, N % 8 == 0; `workspace` = ocn_fused_logits_ce_workspace_floats(R, N) floats. */
int64_t ocn_fused_logits_ce_workspace_floats(int R, int N);
int ocn_fused_logits_ce(const void* X, int ldx, const void* Y, int ldy, int R, int N, int E, int label_offset, float loss_scale,
                        float grad_scale, void* G, int ldg, float* workspace, float* rowscale, float* loss_sum, float* dscale_sum, ocn_stream_t stream);
/* the caller's side of that row scale: out_bf16[r, :] = bf16(scale[r] * x_bf16[r, :]) (the [R, E] operand of G'^T @ X), and
 * out_f32[r, :] -= (alpha / scale[r]) * x_bf16[r, :] (the label rows' -onehot part of that product, from the same rounded rows).  E % 8 == 0. */
int ocn_scale_rows_bf16(const void* x_bf16, int ldx, const float* scale, void* out_bf16, int ldo, int R, int E, ocn_stream_t stream);
int ocn_sub_scaled_rows(float* out, int ldo, const void* x_bf16, int ldx, const float* scale, float alpha, int R, int E, ocn_stream_t stream);
/* bias_dev (may be NULL): the logit bias as a 1-element DEVICE value; overrides `bias` (logit_bias is a parameter: the step never reads it on the
 * host).  It is subtracted per element inside the dscale sum -- not as bias * dbias_sum afterwards, which cancels catastrophically. */
int ocn_siglip_rows(const float* logits, int ld, void* G, int ldg, int R, int N, int label_offset, int negative_only,
                    float bias, const float* bias_dev, float loss_scale, float grad_scale, float inv_logit_scale, float* loss_sum,
                    float* dscale_sum, float* dbias_sum, float* det_rows, ocn_stream_t stream);

/* ---- optimizer (train.py:181-182, image_text_task.py:91-101) --------------
 * sumsq: out[0] += sum(x^2) (grad-norm for clip_grad_norm_);
 * adamw_step: torch.optim.AdamW semantics (decoupled weight decay, bias correction), optional grad scale
 *   (clip coefficient read from device pointer clip_coef or 1.0 when NULL); also refreshes the bf16 shadow
 *   copy when w_bf16 != NULL. */
int ocn_sumsq_accum(const float* x, int64_t n, float* out, ocn_stream_t stream);
int ocn_adamw_step(float* w, const float* g, float* m, float* v, void* w_bf16, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, const float* clip_coef, ocn_stream_t stream);

/* Whole-model optimizer step in one launch (8f-1): `entries` = device array of 88-byte records
 *   { float* w; const float* g; float* m; float* v; bf16* w16n; bf16* w16t; int64 numel; int32 rows, cols, mode, pad;
 *     float lr, wd, bc1 (= 1 - beta1^step), bc2_sqrt (= sqrt(1 - beta2^step)); }
 * `chunks` = device array of int32 pairs {entry, index}: 8192-element ranges (mode 0 = 16-byte vectors, 2 = scalar) or 64x64
 * tiles of a [rows, cols] weight (mode 1; rows, cols % 64 == 0).  AdamW as ocn_adamw_step; additionally rewrites the bf16
 * operand copies w16n [rows, cols] and (mode 1) w16t [cols, rows] when non-NULL.  gnorm_sq != NULL: gradients are scaled by
 * min(1, max_norm / (sqrt(*gnorm_sq) + 1e-6)) = torch.nn.utils.clip_grad_norm_ (train.py:181); ocn_sumsq_multi accumulates
 * the squared norm of every entry's gradient into out[0]. */
int ocn_adamw_multi(const void* entries, const void* chunks, int n_chunks, float beta1, float beta2, float eps,
                    const float* gnorm_sq, float max_norm, ocn_stream_t stream);
/* chunk_ws == NULL: every chunk adds its partial sum to out[0] with an fp32 atomic (order varies from run to run).  chunk_ws = n_chunks floats:
 * the reproducible form -- every chunk stores its partial, one workgroup adds them in chunk order (clip_grad_norm_ under deterministic=True). */
int ocn_sumsq_multi(const void* entries, const void* chunks, int n_chunks, float* out, float* chunk_ws, ocn_stream_t stream);

/* ---- collectives (loss.py:23-54 gather_features and its backward) ------------------
 * Direct RCCL calls on the caller's stream (RCCL is bound at run time from the librccl.so the process has loaded; no link-time
 * dependency).  One process per GPU.  ocn_comm_unique_id: rank 0 makes the 128-byte id, the caller distributes it (any
 * side channel); ocn_comm_init: every rank, collectively.  dtype: 0 = fp32, 1 = bf16 (ocn_comm_broadcast also 2 = raw bytes: any other tensor, exact).  Counts are ELEMENTS per rank.
 *   allgather:           recv [world * count] = concat_r send_r [count]             (the packed [B, 2E] feature exchange)
 *   reduce_scatter_sum:  recv [count] = (sum_r send_r [world * count]) [rank slice]   (backward of the gather, loss.py:23-26)
 *   allreduce_sum:       buf [count] = sum_r buf_r, in place                         (row-sharded loss scalars, gradient buckets) */
int ocn_comm_unique_id(void* id_out_128);
int ocn_comm_init(const void* id_128, int rank, int world, void** comm_out);
int ocn_comm_destroy(void* comm);
int ocn_comm_allgather(void* comm, const void* send, void* recv, int64_t count_per_rank, int dtype, ocn_stream_t stream);
int ocn_comm_reduce_scatter_sum(void* comm, const void* send, void* recv, int64_t count_per_rank, int dtype, ocn_stream_t stream);
int ocn_comm_allreduce_sum(void* comm, void* buf, int64_t count, int dtype, ocn_stream_t stream);
/* in-place MEAN over ranks (ncclAvg): the gradient all-reduce that replaces DistributedDataParallel's reducer (base_task.py:219-232) */
int ocn_comm_allreduce_avg(void* comm, void* buf, int64_t count, int dtype, ocn_stream_t stream);
/* buf [count] on every rank = rank `root`'s, in place (ncclBroadcast): the parameter broadcast at the start of training (DDP's, base_task.py:227) */
int ocn_comm_broadcast(void* comm, void* buf, int64_t count, int dtype, int root, ocn_stream_t stream);
/* what the communicator reports about itself: *count_out = ncclCommCount (ranks RCCL connected), *rank_out = ncclCommUserRank (may be NULL) */
int ocn_comm_count(void* comm, int* count_out, int* rank_out);
/* one neighbour exchange (loss.py:226-243 `neighbour_exchange`: one isend + one irecv batched): send [count] to to_rank, recv [count] from from_rank, as
 * one grouped RCCL operation on the caller's stream; dtype as above (0 fp32, 1 bf16, 2 raw bytes) */
int ocn_comm_sendrecv(void* comm, const void* send, int to_rank, void* recv, int from_rank, int64_t count, int dtype, ocn_stream_t stream);

/* ---- the persistent GEMMs next to other streams' kernels (multi-GPU) ----------------------------
 * ocn_gemm_nt / ocn_gemm_tn_accum launch one workgroup per CU with a fixed share of the tiles.  A workgroup whose CU is held by another stream's kernel
 * (RCCL's, during the gradient all-reduce) starts only when a CU frees up and then walks its whole share alone: the launch takes up to twice as long.
 * ocn_set_tile_rescue(1) (process-wide, takes effect with the next launch) selects the kernels' rescue form: the shares stay static and no atomic enters
 * the tile loop, but workgroups that finish hand out -- entry by entry, through one counter per workgroup on a per-stream board -- the shares of workgroups
 * that have not started.  Results: ocn_gemm_nt bit-identical to the static form (every tile is computed once, by whichever workgroup), ocn_gemm_tn_accum
 * sums the same products with fp32 atomics in a different order, as between any two of its runs; the reproducible wgrad (workspace form) and launches
 * recorded into a graph (stream capture) stay static.
 * Cost without contention: one atomic per workgroup at either end of a launch.  Off by default; open_clip_amd turns it on when world_size > 1. */
int ocn_set_tile_rescue(int on);
int ocn_get_tile_rescue(void);

/* ---- self-test probes (used by tests/ to pin the hardware fragment layouts this library assumes) */
int ocn_probe_mfma32(const void* a_bf16 /*[32,16]*/, const void* b_bf16 /*[32,16] (n,k)*/, float* c /*[32,32]*/,
                     ocn_stream_t stream);
int ocn_probe_tr16(const void* in_bf16 /*[16 rows][32 cols]*/, void* out_bf16 /*[64 lanes][4]*/, ocn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
