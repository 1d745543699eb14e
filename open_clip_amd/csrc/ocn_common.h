// Shared device/host helpers for the gfx950 (MI355X, CDNA4) CLIP kernels.
// wave = 64 lanes everywhere; MFMA = v_mfma_f32_32x32x16_bf16 (fp32 accumulate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openclip_hip.h"
#include "../../include/openclip_hip_debug.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define OCN_LDS __attribute__((address_space(3)))
#define OCN_GLB __attribute__((address_space(1)))
#define OCN_DEV __device__ __forceinline__

// ---- error plumbing (C ABI returns 0 / negative code; message via ocn_last_error) -------------
void ocn_set_error(const char* fmt, ...);
#define OCN_CHECK_ARG(cond, ...)        \
    do {                                \
        if (!(cond)) {                  \
            ocn_set_error(__VA_ARGS__); \
            return OCN_ERR_INVALID;     \
        }                               \
    } while (0)
#define OCN_CHECK_LAUNCH(name)                                                      \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            ocn_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
            return OCN_ERR_LAUNCH;                                                  \
        }                                                                           \
    } while (0)

// ---- small device helpers ----------------------------------------------------------------------
OCN_DEV float bf2f(bf16 v) { return (float)v; }
OCN_DEV bf16 f2bf(float v) { return (bf16)v; }  // round-to-nearest-even (v_cvt_pk_bf16_f32 on gfx950)

// 2^x by the hardware instruction alone (v_exp_f32, 1 ulp): the softmax kernels work in the exp2 domain and only ever see x <= 0 or a
// discarded lane -- exp2f() adds a denormal-range rescue (5 more VALU operations per element) that nothing here needs
OCN_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

OCN_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
OCN_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 16-byte async global -> LDS copy (LDS-DMA).  LDS destination = wave-uniform base + lane*16.
OCN_DEV void glds16(const void* gptr, OCN_LDS void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const OCN_GLB void*)gptr, lds_wave_base, 16, 0, 0);
}

// LDS transposing read: within each 16-lane group, lane i passes the address of 4 contiguous bf16
// = element [i>>2][(i&3)*4 .. +3] of a 4 x 16 block and receives column i: {blk[0][i] .. blk[3][i]}.
OCN_DEV s16x4 lds_read_tr16(const OCN_LDS void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((OCN_LDS s16x4*)p);
}

OCN_DEV f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// row of accumulator register `reg` (0..15) of a 32x32 MFMA result for this lane; column = lane & 31
OCN_DEV int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// erf-GELU (nn.GELU(), reference transformer.py:295-299) and its derivative, gelu(x) AND gelu'(x) from one evaluation of the shared parts (the
// forward GELU epilogue saves the derivative for the backward).  The GEMM epilogue that applies it is VALU-issue bound with every MFMA pipe idle
// (0.17-0.21 ms of the 1.1 ms c_fc GEMM at batch 4096: profiles/r02_nt6_trickled_epilogue_experiment.txt), so the issue-slot count is what matters.
//   Phi(x) - 1/2 = x Q(x^2) on |x| <= 4.25, x clamped beyond (Phi(4.25) = 1 - 1.07e-5): the normal CDF as an ODD POLYNOMIAL, nine coefficients from
//   tools/gelu_poly_fit.py (weighted minimax; |Phi error| <= 1.24e-5 in fp32 -- 1/160 of the bf16 resolution of the stored result; the coefficients
//   in this header are checked against erf in float64 by tests/test_gelu_poly.py);  gelu = x Phi;  gelu' = Phi + x exp(-x^2/2) / sqrt(2 pi).
// No v_rcp_f32, no copysign; on a register quad every plain operation is a <4 x float> one that gfx950 issues as two v_pk_*_f32: 18.9 static VALU
// slots per element of the GELU kernel against 23.2 for the Abramowitz-Stegun 7.1.25 erfc form that shipped until round 4 (now `gelu_both_as`,
// developer build only).  Adopted in round 5 after the kernel suite ran on it: -2.0 / -2.7 % on the two c_fc + GELU GEMMs
// (profiles/r04_dev_polynomial_gelu_ab.txt).
OCN_DEV void gelu_both_poly4(f32x4 x, f32x4& g, f32x4& dg) {
    f32x4 xc;
#pragma unroll
    for (int i = 0; i < 4; ++i) xc[i] = __builtin_amdgcn_fmed3f(x[i], -4.25f, 4.25f);
    const f32x4 u = xc * xc;
    f32x4 q = u * 5.565356162e-11f + -5.328117947e-09f;
    q = q * u + 2.255534781e-07f;
    q = q * u + -5.626594884e-06f;
    q = q * u + 9.342017438e-05f;
    q = q * u + -1.108568278e-03f;
    q = q * u + 9.815989994e-03f;
    q = q * u + -6.634451449e-02f;
    q = q * u + 3.989023566e-01f;
    const f32x4 cdf = xc * q + 0.5f;
    const f32x4 ku = (x * x) * -0.72134752044448170f;  // of the UNCLAMPED x: phi -> 0 beyond the clamp
    f32x4 e;
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(ku[i]);  // exp(-x^2/2)
    g = x * cdf;
    dg = (xc * 0.39894228040143268f) * e + cdf;
}
// (Tail of the quad form: beyond the clamp the CDF stays at Phi(+-4.25), so gelu(x) = 1.07e-5 * x instead of 0 for x < -4.25 -- an absolute error of
// 1.07e-5 |x|, below the bf16 resolution of any neighbouring value for |x| < 100; one more select per element would remove it and is not spent there.)
// One element, the same polynomial in scalar arithmetic (the general fallback GEMM's epilogue: small / ragged shapes) -- with exact tails: the CDF is
// 0 / 1 beyond the clamp.
OCN_DEV void gelu_both(float x, float& g, float& dg) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.25f, 4.25f);
    const float u = xc * xc;
    float q = fmaf(u, 5.565356162e-11f, -5.328117947e-09f);
    q = fmaf(q, u, 2.255534781e-07f);
    q = fmaf(q, u, -5.626594884e-06f);
    q = fmaf(q, u, 9.342017438e-05f);
    q = fmaf(q, u, -1.108568278e-03f);
    q = fmaf(q, u, 9.815989994e-03f);
    q = fmaf(q, u, -6.634451449e-02f);
    q = fmaf(q, u, 3.989023566e-01f);
    float cdf = fmaf(xc, q, 0.5f);
    cdf = x < -4.25f ? 0.f : (x > 4.25f ? 1.f : cdf);
    const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170f);  // exp(-x^2/2)
    g = x * cdf;
    dg = fmaf(xc * 0.39894228040143268f, e, cdf);
}
#ifdef OCN_DEV_BUILD
// the form that shipped until round 4 (A/B knob of the developer build): erfc by Abramowitz-Stegun 7.1.25, |abs err| <= 2.5e-5;
//   h(x) = erfc(|x|/sqrt 2) / 2 = t (a1 + t (a2 + t a3)) / 2 * exp(-x^2/2),  t = 1 / (1 + p |x| / sqrt 2);  Phi(x) = 1/2 + copysign(1/2 - h, x)
OCN_DEV void gelu_both_as(float x, float& g, float& dg) {
    const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.47047f * 0.70710678118654752f, 1.0f));
    const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * (x * x));  // exp(-x^2/2)
    float p = fmaf(0.5f * 0.7478556f, t, 0.5f * -0.0958798f);
    p = fmaf(p, t, 0.5f * 0.3480242f);
    const float h = p * t * e;
    const float cdf = 0.5f + __builtin_copysignf(0.5f - h, x);
    g = x * cdf;
    dg = fmaf(x * 0.39894228040143268f, e, cdf);
}
#endif
// The derivative saved for the backward (aux of OCN_EPI_BIAS_GELU / OCN_EPI_DGELU) is stored in 8 bits: gelu'(x) lies in
// [-0.1290, 1.1290], q = round((gelu' + 0.13) * 200) in [0, 252], gelu' ~ q / 200 - 0.13 with |error| <= 0.0025 (uniform, unbiased;
// rms 0.0014 -- what rounding a value in [0.5, 1) to bf16 costs).  Half the bytes of a bf16 copy in the forward epilogue's second
// output and in the backward epilogue's operand load.  The rounding is done by the fp32 adder: x * 200 + (2^23 + 26) has its integer
// part in the low mantissa bits (round-to-nearest-even), so the low byte of the result's bit pattern IS q.
constexpr float OCN_DGELU_SCALE = 200.0f, OCN_DGELU_OFFSET = 0.13f;
// (the code is clamped to [0, 255] in the float domain -- one v_med3_f32: gelu' / QuickGELU' stay inside [0, 252] by construction, a pre-activation that
// is NaN / Inf or any future activation must not wrap around in the low byte)
OCN_DEV unsigned dgelu_q_bits(float d) {
    const float r = fmaf(d, OCN_DGELU_SCALE, 8388608.0f + OCN_DGELU_OFFSET * OCN_DGELU_SCALE);
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_fmed3f(r, 8388608.0f, 8388608.0f + 255.0f));
}
OCN_DEV unsigned dgelu_pack4(float d0, float d1, float d2, float d3) {
    const unsigned q0 = dgelu_q_bits(d0), q1 = dgelu_q_bits(d1), q2 = dgelu_q_bits(d2), q3 = dgelu_q_bits(d3);
    // v_perm_b32: byte i of the result = byte sel[i] of {S0 (4..7), S1 (0..3)}
    const unsigned lo = __builtin_amdgcn_perm(q1, q0, 0x0c0c0400u), hi = __builtin_amdgcn_perm(q3, q2, 0x0c0c0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}
OCN_DEV float dgelu_unq(unsigned q) { return fmaf((float)q, 1.0f / OCN_DGELU_SCALE, -OCN_DGELU_OFFSET); }
OCN_DEV f32x4 dgelu_unpack4(unsigned q) {
    return (f32x4){dgelu_unq(q & 255u), dgelu_unq((q >> 8) & 255u), dgelu_unq((q >> 16) & 255u), dgelu_unq(q >> 24)};
}

// QuickGELU (reference layers.py:29-32: x * sigmoid(1.702 x); the OpenAI / LAION-400M checkpoints were trained with it) and its derivative
// s + 1.702 x s (1 - s), which lies in [-0.099, 1.099]: inside the 8-bit code range of the saved derivative above
OCN_DEV void quickgelu_both(float x, float& g, float& dg) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
    g = x * s;
    dg = fmaf(1.702f * g, 1.0f - s, s);
}
// activation of an epilogue: gelu_both for OCN_EPI_BIAS_GELU, quickgelu_both for OCN_EPI_BIAS_QUICKGELU
template <bool QUICK>
OCN_DEV void act_both(float x, float& g, float& dg) {
    if (QUICK) quickgelu_both(x, g, dg); else gelu_both(x, g, dg);
}

// bijective XCD-aware remap of a linear workgroup id: hardware places block b on XCD b % 8, so give
// each XCD a contiguous chunk of the logical tile space (neighbouring tiles share operand panels in L2).
OCN_DEV int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    if (nwg < nx) return bid;
    const int xcd = bid % nx, idx = bid / nx;
    const int q = nwg / nx, r = nwg % nx;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

static inline int ocn_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
