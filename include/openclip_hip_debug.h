/*
 * openclip_hip_debug.h -- DEVELOPER entry points of libopenclip_hip.so: kernel-selection and ablation knobs used by the experiment
 * tools (tools/gemm_bench.py, tools/sweep.py, tools/occupancy_hazard_probe.py, tools/gemm_trace.py) and by tests that force a
 * fallback kernel.  NOT part of the drop-in boundary (include/openclip_hip.h): the knobs are process-global, several of them
 * deliberately produce WRONG results (they skip arithmetic or memory traffic to time what is left), and nothing in open_clip_amd/
 * calls them.  Defaults (all zero) are the shipped behaviour.
 */
#ifndef OPENCLIP_HIP_DEBUG_H
#define OPENCLIP_HIP_DEBUG_H
#include "openclip_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* kernel choice of ocn_gemm_nt / ocn_gemm_tn_accum (process-global):
 *   bits 0..3  NT kernel: 0 auto, 4 the general 256x256 ring kernel, 5 the persistent 256x256 kernel (falls back to 4)
 *   bits 4..7  TN kernel: 0 auto, 1 the general 128x128 kernel, 3 the hand-scheduled 256x256 kernel (falls back to 1)
 *   bits 8..   developer knobs of the persistent NT kernel: (v >> 8) & 1 skip GELU arithmetic (results wrong), & 2 / & 8 flip the
 *              epilogue's store / load cache policy, & 4 drain stores per tile, & 16 non-temporal A-operand loads, & 32 / & 128 drop the
 *              epilogue's stores / operand loads (results wrong), & 64 timeline build, & 0x200000 whole tail tiles instead of half tiles,
 *              & 0x400000 GELU arithmetic by the Abramowitz-Stegun erfc form that shipped until round 4 (gelu_both_as; the product uses the polynomial
 *                         normal CDF gelu_both_poly4; same tolerances, tests/test_dev_build_gpu.py);
 *              (v >> 16) & 31 tile-walk band width; (v >> 21) & 63 start stagger in us per class (0 / 63 = off = shipped since round 5; 62 = the automatic rule that shipped until round 4).
 *              The knobs of this group that live INSIDE the kernel exist in the developer build only (libopenclip_hip_dev.so:
 *              python -m open_clip_amd.build --dev, -DOCN_DEV_BUILD; select it with OCN_LIB_PATH): the product library compiles none of
 *              them and ignores these bits. */
int ocn_set_gemm_variant(int nt_variant);
/* developer knobs (process-global; experiments and A/B measurements of tools/sweep.py, never needed by a user):
 *   key 1  attention-backward ablation mask (1 skip the input staging, 2 skip the arithmetic, 4 skip the stores: results wrong)
 *   key 3  persistent NT GEMM (developer build): which workgroups share a start-stagger class: 0 consecutive workgroups alternate (shipped), 1 per XCD, 2 per tile row of the band
 *   key 2  attention backward: 4 = two-pass dK / dV build, 5 = one-pass build (default: two-pass up to 4 waves)   key 4  wgrad GEMM: 1 = skip the atomic epilogue
 *   key 5  attention backward: extra KiB of LDS per workgroup (occupancy probe)  key 6  1 = generic instead of causal bwd kernel
 *   key 7  1 = always the streamed (explicit head_dim) attention kernels, 2 = always the head-resident ones (head_dim 64, L <= 320)
 *   key 8  1 = LayerNorm backward, default cache policy
 *   key 9  2 = attention forward, non-temporal policy for its LDS-DMA loads
 *   key 10 workgroups per CU of the persistent NT GEMM's grid (0 = default 1)   key 11 wgrad GEMM: M-splits per CU when few (0/1 = one)
 *   key 12 1 = LayerNorm forward, default cache policy for x (default: non-temporal)
 *   key 13 1 = ocn_gemm_tn_accum2 never pairs (runs its two problems as two launches: A/B of the paired wgrad)
 *   key 14 n = workgroups of the LayerNorm backward's grid (default: one 16-wave workgroup per CU) */
int ocn_set_tuning(int key, int value);
/* developer probe: n workgroups that each hold (most of) a CU's LDS for `micros` microseconds on `stream` -- a stand-in for
 * collective kernels occupying CUs while a persistent GEMM starts (tools/occupancy_hazard_probe.py) */
int ocn_debug_occupy(int n_workgroups, int micros, int* sink, ocn_stream_t stream);

/* developer probe: creates a HIP stream restricted to the compute units whose bits are set in mask[0..nwords) (hipExtStreamCreateWithCUMask) */
int ocn_debug_stream_with_cu_mask(const uint32_t* mask, int nwords, void** stream_out);
/* per-tile timeline of the persistent NT kernel's developer build (knob bit 64): copies 1024 int64 stamps to host_out */
int ocn_debug_nt5_trace(long long* host_out);

#ifdef __cplusplus
}
#endif
#endif
