// Shared host/device definitions for the gfx950 LGTEUN kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include "../../include/lgteun_hip.h"

// ----------------------------------------------------------------------------------------------
// parameter slots, in Pansharpening.state_dict() order (reference models/unlg_former.py:22-48,
// models/common/LGT.py:252-303; key list in SURVEY.md section 8b)
// ----------------------------------------------------------------------------------------------
enum SharedSlot { S_D1W, S_D1B, S_D3W, S_D3B, S_DT1W, S_DT1B, S_DT3W, S_DT3B, S_RW, S_RB, S_RTW, S_RTB, S_NSHARED };

// one LGB block = 21 tensors: LGMixer (11) then feed_forward (10)
enum BlockSlot {
    B_POS, B_QKVW, B_QKVB, B_AMPW, B_AMPB, B_PHAW, B_PHAB, B_PROJW, B_PROJB, B_LN1G, B_LN1B,
    B_W1, B_B1, B_W2, B_B2, B_DWW, B_DWB, B_W3, B_B3, B_LN2G, B_LN2B, B_NSLOT
};
// LGT-level layout: patch_embed(6) enc0(21) enc1(21) down(2) bott(21) up(2) fuse(2) dec0(21) dec1(21) tail(2)
enum LgtSlot {
    L_PE_DWW = 0, L_PE_DWB, L_PE_W, L_PE_B, L_PE_LNG, L_PE_LNB,
    L_ENC0 = 6, L_ENC1 = 27, L_DOWNW = 48, L_DOWNB, L_BOTT = 50, L_UPW = 71, L_UPB, L_FUSEW, L_FUSEB,
    L_DEC0 = 75, L_DEC1 = 96, L_TAILW = 117, L_TAILB, L_NSLOT = 119
};
static inline int block_base(int blk) {
    const int b[5] = {L_ENC0, L_ENC1, L_BOTT, L_DEC0, L_DEC1};
    return b[blk];
}

struct lg_plan {
    lg_config cfg;
    int n_offsets;
    int ffn_tile;  // A/B switch (lg_config.variant & LG_VAR_FFN_IMPL_MASK; Python side: LG_FFN_IMPL = strip | tile | xp): the f32-MFMA fused FFN kernels instead of the split-bf16 ones
    int save_mode; // A/B switch (lg_config.variant LG_VAR_FFN_SAVE3 | _SAVE5; Python side: LG_FFN_SAVE = 5 | 3 | 2, default 2): what the live stage's e = 16 FFN half-blocks
    // keep for the backward.  2 (default): the pre-activations h2, h3 -- h1 is re-computed from x by k_ffn1_bwd_xs (k_ffn_bwd_x.hip), which
    // also forms dW1 / dW2 on the bf16 matrix pipe; 3: h1, h2, h3 (round 2's default: k_ffn1_bwd<16> + k_wgrad_t re-evaluate gelu / gelu');
    // 5: gelu(h1), gelu'(h1), h2, gelu(h3), gelu'(h3) (GELU-free backward; the only form of the other widths and of precision = 'bf16').
    // In modes 2 / 3 the tensors sit in the a1 / h2 / a3 slots (workspace.h) and the g1 / g3 slots stay unused.
    int dwbwd_tile; // A/B switch (lg_config.variant LG_VAR_FFN_DWBWD_TILE; Python side: LG_FFN_DWBWD=tile): round 2's tile kernel k_ffn_dw_bwd<16> + k_wgrad_t for dW3
                    // instead of the strip-walking k_ffn_dw_bwd_xs
    // precision = 'bf16' applies where a plain-bf16 kernel exists: e = 16 and e = 32.  The e = 64 half-blocks (level 1 of the 8-band net) have
    // only the round-1 f32-MFMA pair in that form (436 + 372 us against 123 + 95 us for the split-bf16 k_ffn_x64 pair), so they run the
    // default kernels with fp32 storage in both modes -- 'bf16' is never slower than the default (c3 / c5)
    bool hidden_bf16(int e) const { return cfg.precision == 1 && e != 64; }
    // precision = 'bf16' (plain bf16 MFMA, bf16 storage of the saved tensors) knows modes 2 and 5 only (3 falls back to 5)
    int attn_bwd_old; // A/B switch (lg_config.variant LG_VAR_ATTN_BWD_R3; Python side: LG_ATTN_BWD=r3): 1 = round 3's k_attn_bwd_core + k_attn_bwd_epi + k_wgrad_t at e = 16 instead of k_attn_bwd_f
    int ffn_bwd_bf16x3; // A/B switch (lg_config.variant LG_VAR_FFN_BWD_BF16X3; Python side: LG_FFN_BWD_SPLIT=bf16x3): the pixelwise half of the FFN backward on three bf16 pieces / six products (rounds 3 - 4) instead of f16 pairs
    int attn_bwd_core_m; // A/B switch (lg_config.variant LG_VAR_ATTN_BWD_CORE_M; Python side: LG_ATTN_BWD_CORE=m): the matrix-pipe k_attn_bwd_core_m at e = 32 instead of the vector-pipe k_attn_bwd_core
    int fft_full;      // A/B switch (lg_config.variant LG_VAR_FFT_FULL; Python side: LG_FFT=full): complex-row in-LDS FFT mixer kernels instead of the real-input ones
    int ffn_bf16x3;    // A/B switch (lg_config.variant LG_VAR_FFN_BF16X3; Python side: LG_FFN_SPLIT=bf16x3): the fused FFN forward's GEMMs as three bf16 pieces / six
                       // products (round 2) instead of two f16 pieces / three products with proven power-of-two operand scales (round 5, k_ffn_prep.hip)
    bool ffn_f16x2(int e) const { return cfg.precision == 0 && ffn_tile == 0 && !ffn_bf16x3 && (e == 16 || e == 32 || e == 64); }
    int attn_restats;  // A/B switch (lg_config.variant LG_VAR_ATTN_BWD_RESTATS; Python side: LG_ATTN_BWD_STATS=recompute): k_attn_bwd_f re-derives the softmax row statistics instead of reading the forward's (round 6)
    int ffn_h3_re;     // A/B switch (lg_config.variant LG_VAR_FFN_H3_RECOMPUTE; Python side: LG_FFN_H3=recompute): the live stage's e = 16 FFN saves h2 only and its backward re-computes h3 (k_ffn_dw_bwd_h, round 6) instead of saving h2 and h3 (k_ffn_dw_bwd_xs, the default)
    bool ffn_h3_recompute(int e) const { return e == 16 && ffn_bwd_x(e) && cfg.precision == 0 && !dwbwd_tile && ffn_h3_re && !ffn_xs && ffn_f16x2(e); }
    int ffn_xs;        // A/B switch (lg_config.variant LG_VAR_FFN_XS; Python side: LG_FFN_FWD=xs): rounds 2 - 5's channel-split k_ffn_xs at e = 16 instead of the register-chain k_ffn_xr
    int attn_bf16x3;   // A/B switch (lg_config.variant LG_VAR_ATTN_BF16X3; Python side: LG_ATTN_SPLIT=bf16x3): to_qkv and Q K^T of k_attn_m on three bf16 pieces / six products (round 5) instead of f16 pairs with static scales (round 6)
    // the live stage's local mixers leave their row log-sum-exp and attention output for k_attn_bwd_f / k_attn_bwd_core (fp32-equivalent mode, matrix-pipe forward)
    bool attn_saves_stats(int e) const { return cfg.precision == 0 && !attn_fwd_valu && !attn_restats && !(e == 32 && attn_bwd_core_m); }
    bool attn_f16x2() const { return cfg.precision == 0 && !attn_bf16x3 && !attn_fwd_valu && ffn_tile == 0 && !ffn_bf16x3; }   // (the scales ride in the FFN prep launch: the f16-pair FFN arithmetic must be on)
    int attn_fwd_valu; // A/B switch (lg_config.variant LG_VAR_ATTN_FWD_VALU; Python side: LG_ATTN_FWD=valu): round 2's vector-pipe k_attn instead of the matrix-pipe k_attn_m
    int dstep_tiles; // A/B switch (lg_config.variant LG_VAR_DSTEP_TILES; Python side: LG_DSTEP=tiles): the tile kernels of the data step also where the one-launch
                     // plane-in-LDS form (k_dstep.hip) exists
    bool dstep_fused(int h, int w) const;   // k_dstep.hip: square planes of 128 or 64
    int bwd32_old; // A/B switch (lg_config.variant LG_VAR_FFN_BWD32_PAIR turns it on; Python side: LG_FFN_BWD32=pair): 1 = k_ffn1_bwd_x32 + two k_wgrad_t launches at e = 32 (round 2 .. 4's default);
                   // 0 = k_ffn1_bwd_xs<32>, the e = 16 kernel's template at 8 waves / one workgroup per CU -- correct, but slower there
    bool ffn1_bwd_x32(int e) const { return e == 32 && ffn_tile == 0 && !bwd32_old; }
    // e = 32 (round 4): the spatial half through the strip-walking k_ffn_dw_bwd_xs<32> (dW3 / db3 included): the forward saves the PRE-activation h3
    // in the a3 slot and nothing in g3; the pixelwise half stays k_ffn1_bwd_x32 + the 128 x 128 weight-gradient launch on the saved gelu(h1) / gelu'(h1)
    // (a strip is 16 columns wide and every output pixel of a step must be inside the plane: level-1 planes whose width is 8 mod 16 keep round 2's kernels)
    bool ffn_dw_x32(int e, int h, int w) const { return e == 32 && ffn_tile == 0 && !dwbwd_tile && (h & 7) == 0 && (w & 15) == 0; }
    bool ffn_bwd_x(int e) const { return e == 16 && ffn_tile == 0 && save_mode == 2; }   // h1 not saved; backward through k_ffn_dw_bwd_xs + k_ffn1_bwd_xs
    bool ffn_saves_preact(int e) const { return e == 16 && ffn_tile == 0 && (save_mode == 2 || (save_mode == 3 && cfg.precision == 0)); }
    int64_t* off;  // host copy of offsets
    int64_t shared(int s) const { return off[s]; }
    int64_t eta(int i) const { return off[S_NSHARED + i]; }
    int64_t lgt(int stage, int slot) const { return off[S_NSHARED + cfg.K + stage * L_NSLOT + slot]; }
    int64_t blk(int stage, int b, int slot) const { return lgt(stage, block_base(b) + slot); }
};

void lg_set_error(const char* fmt, ...);

// one-time per-DEVICE setup of a kernel (hipFuncSetAttribute for > 64 KiB of dynamic LDS).  Safe from several host threads: the
// guarded call is idempotent, so a race only repeats it.
struct DeviceOnce {
    std::atomic<uint64_t> mask{0};
    static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
    bool need() const { return !((mask.load(std::memory_order_acquire) >> dev()) & 1); }
    void done() { mask.fetch_or(1ull << dev(), std::memory_order_release); }
};

// per-(stage, block) dropout seed, shared by forward and backward
static inline uint64_t mix_seed(uint64_t seed, int stage, int blk) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(stage * 8 + blk + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}


#define LG_CHECK_LAUNCH()                                                              \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            lg_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return (int)e__;                                                           \
        }                                                                              \
    } while (0)

#define LG_EPS 1e-5f

// live per-kernel timing (api.hip): RAII scope used at the top of a launcher
void lg_prof_begin(int kid, hipStream_t s);
void lg_prof_end(int kid, hipStream_t s);
struct ProfScope {
    int kid;
    hipStream_t s;
    ProfScope(int k, hipStream_t st) : kid(k), s(st) { lg_prof_begin(kid, s); }
    ~ProfScope() { lg_prof_end(kid, s); }
};

#ifdef __HIPCC__
// ----------------------------------------------------------------------------------------------
// device helpers
// ----------------------------------------------------------------------------------------------
// erf(|z|) by Abramowitz & Stegun 7.1.26 (max abs error 1.5e-7, i.e. fp32 rounding level), branch-free:
// 1 rcp + 1 exp + 6 fma instead of the piecewise libm erff.  Returns 1 - erf(|z|) as `tail` too (keeps the far
// negative GELU tail from cancelling against 1).
__device__ __forceinline__ float erfc_abs_f(float az) {
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    return poly * __expf(-az * az);
}
// GELU(x) = x * Phi(x), Phi(x) = 0.5*erfc(-x/sqrt2)   (nn.GELU() exact/erf form, LGT.py:97,99)
__device__ __forceinline__ float gelu_f(float x) {
    const float z = x * 0.70710678118654752440f;
    const float e = 0.5f * erfc_abs_f(fabsf(z));      // 0.5*erfc(|z|)
    const float phi = z >= 0.f ? 1.0f - e : e;         // Phi(x)
    return x * phi;
}
// gelu and its derivative together (they share erfc and the exp): a = x*Phi(x), g = Phi(x) + x*phi(x)
__device__ __forceinline__ void gelu_both_f(float x, float& a, float& g) {
    const float az = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float ex = __expf(-az * az);            // exp(-x^2/2)
    const float e = 0.5f * poly * ex;
    const float cdf = x >= 0.f ? 1.0f - e : e;
    a = x * cdf;
    g = cdf + x * (0.39894228040143267794f * ex);
}
// Two GELUs per instruction stream: the polynomial / scaling work on float2 ext-vectors compiles to v_pk_fma_f32 / v_pk_mul_f32
// (gfx950 packed fp32), so a pair costs ~13 packed VALU slots + 4 transcendentals instead of 2 x (16 + 2).  Same formula and
// constants as gelu_f (A&S 7.1.26 erfc), with the branch-free identity  x * Phi(x) = 0.5 x + |x| (0.5 - 0.5 erfc(|z|)).
typedef float lg_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lg_v2f gelu2_core(lg_v2f x, lg_v2f& absx, lg_v2f& ex) {
    absx = (lg_v2f){fabsf(x.x), fabsf(x.y)};
    const lg_v2f az = absx * 0.70710678118654752440f;
    const lg_v2f d = az * 0.3275911f + 1.0f;
    const lg_v2f t = (lg_v2f){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const lg_v2f poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const lg_v2f m = az * az * -1.44269504088896340736f;   // -az^2 * log2(e)
    ex = (lg_v2f){__builtin_amdgcn_exp2f(m.x), __builtin_amdgcn_exp2f(m.y)};   // exp(-x^2 / 2)
    return poly * ex * -0.5f + 0.5f;                       // 0.5 - 0.5 erfc(|z|) = Phi(|x|) - 0.5
}
__device__ __forceinline__ lg_v2f gelu2_f(lg_v2f x) {
    lg_v2f absx, ex;
    const lg_v2f sv = gelu2_core(x, absx, ex);
    return absx * sv + x * 0.5f;
}
// a = gelu(x), g = gelu'(x) = Phi(x) + x phi(x),  Phi(x) = 0.5 + copysign(Phi(|x|) - 0.5, x)
__device__ __forceinline__ void gelu2_both_f(lg_v2f x, lg_v2f& a, lg_v2f& g) {
    lg_v2f absx, ex;
    const lg_v2f sv = gelu2_core(x, absx, ex);
    a = absx * sv + x * 0.5f;
    const lg_v2f cs = (lg_v2f){copysignf(sv.x, x.x), copysignf(sv.y, x.y)};
    g = x * (ex * 0.39894228040143267794f) + (cs + 0.5f);
}
// gelu of a value that arrives SCALED by a power of two s_in and has to leave scaled by s_out (the f16-pair arithmetic of the FFN kernels,
// split_bf16.h NP = 2): returns s_out * gelu(x / s_in) for c1 = sqrt(1/2) / s_in, hr = 0.5 s_out / s_in.  The same instruction sequence as
// gelu2_f with two of its immediates replaced by these (power-of-two multiples of them): bit for bit s_out times gelu2_f's result.
__device__ __forceinline__ lg_v2f gelu2_scaled(lg_v2f x, float c1, float hr) {
    const lg_v2f absx = (lg_v2f){fabsf(x.x), fabsf(x.y)};
    const lg_v2f az = absx * c1;
    const lg_v2f d = az * 0.3275911f + 1.0f;
    const lg_v2f t = (lg_v2f){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const lg_v2f poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const lg_v2f m = az * az * -1.44269504088896340736f;
    const lg_v2f ex = (lg_v2f){__builtin_amdgcn_exp2f(m.x), __builtin_amdgcn_exp2f(m.y)};
    const lg_v2f sv = poly * ex * (0.0f - hr) + hr;
    return absx * sv + x * hr;
}
// The same function with every constant that CAN be folded folded (round 6: the GELUs are half of the fused FFN's vector instructions): for a kernel-wide
// pair (c1, hr) the caller builds a GeluK once -- w = |x| c1 sqrt(log2 e) serves both the exponential (exp2(-w^2): the negation is a source
// modifier) and, with the A&S constant divided by the same factor, the rational argument; the polynomial's coefficients carry -hr; |x| is a
// source modifier of the two scalar products that use it (no v_and).  15 issue slots per pair of values instead of 19 (81.6 against 95.5 SIMD
// cycles at the measured rates).  Same approximation (A&S 7.1.26), rounding differs in the last place.
struct GeluK { float k1, kd, hr, a1, a2, a3, a4, a5; };
__device__ __forceinline__ GeluK gelu_k(float c1, float hr) {
    GeluK k;
    k.k1 = c1 * 1.20112240878645f;                  // sqrt(log2 e)
    k.kd = 0.3275911f / 1.20112240878645f;
    k.hr = hr;
    k.a1 = -hr * 0.254829592f; k.a2 = -hr * -0.284496736f; k.a3 = -hr * 1.421413741f; k.a4 = -hr * -1.453152027f; k.a5 = -hr * 1.061405429f;
    return k;
}
__device__ __forceinline__ lg_v2f gelu2_k(lg_v2f x, const GeluK& k) {
    const lg_v2f w = (lg_v2f){fabsf(x.x) * k.k1, fabsf(x.y) * k.k1};
    const lg_v2f d = w * k.kd + 1.0f;
    const lg_v2f t = (lg_v2f){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const lg_v2f ph = ((((k.a5 * t + k.a4) * t + k.a3) * t + k.a2) * t + k.a1) * t;     // -hr erfc(|z|) exp(z^2)
    const lg_v2f w2 = w * w;
    const lg_v2f ex = (lg_v2f){__builtin_amdgcn_exp2f(-w2.x), __builtin_amdgcn_exp2f(-w2.y)};
    const lg_v2f sv = ph * ex + k.hr;                                                  // hr (1 - erfc(|z|)) = 2 hr (Phi(|x|) - 0.5)
    return (lg_v2f){__builtin_fmaf(fabsf(x.x), sv.x, x.x * k.hr), __builtin_fmaf(fabsf(x.y), sv.y, x.y * k.hr)};
}
// precision = 'bf16' (the NP = 1 instances of the FFN kernels): GELU in its tanh form, x * sigmoid(2 sqrt(2/pi) (x + 0.044715 x^3)) -- the
// nn.GELU(approximate='tanh') function.  Max deviation from the erf form 4.7e-4 (gelu) / 8.7e-4 (gelu'), i.e. below the resolution of the
// bf16 operands it is rounded to (3.9e-3 relative), at 5 packed instructions + 2 transcendentals per pair of values against 13 + 4: the
// erf form was ~45 % of that mode's FFN kernels.  Forward and backward instances use the same pair of functions (FAST = (NP == 1)).
__device__ __forceinline__ lg_v2f gelu2_fast_sig(lg_v2f x, lg_v2f& x2) {
    x2 = x * x;
    const lg_v2f v = x * (x2 * -0.1029432395800235f + -2.302208198144325f);      // -(2 sqrt(2/pi) (x + 0.044715 x^3)) log2(e)
    const lg_v2f d = (lg_v2f){__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)} + 1.0f;
    return (lg_v2f){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};    // sigmoid; exp2 -> inf gives 1 / inf = 0 for x << 0
}
template <bool FAST>
__device__ __forceinline__ lg_v2f gelu2_t(lg_v2f x) {
    if constexpr (FAST) {
        lg_v2f x2;
        return x * gelu2_fast_sig(x, x2);
    } else {
        return gelu2_f(x);
    }
}
template <bool FAST>
__device__ __forceinline__ void gelu2_both_t(lg_v2f x, lg_v2f& a, lg_v2f& g) {
    if constexpr (FAST) {
        lg_v2f x2;
        const lg_v2f sg = gelu2_fast_sig(x, x2);
        a = x * sg;
        g = a * (1.0f - sg) * (x2 * 0.21406444881780073f + 1.5957691216057308f) + sg;   // d/dx [x s(v(x))] = s + x s (1 - s) v'(x)
    } else {
        gelu2_both_f(x, a, g);
    }
}
// d gelu / dx = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float z = x * 0.70710678118654752440f;
    const float e = 0.5f * erfc_abs_f(fabsf(z));
    const float cdf = z >= 0.f ? 1.0f - e : e;
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// cubic-convolution weights, A = -0.75 (torch upsample_bicubic2d); taps at i0-1 .. i0+2.
// All t used here are dyadic, so these are exact in fp32.
__device__ __forceinline__ void cubic_w(float t, float w[4]) {
    const float A = -0.75f;
    float x;
    x = t + 1.0f; w[0] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
    x = t;        w[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 1.0f - t; w[2] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 2.0f - t; w[3] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
}
// 1-D resampling plan for output index o (F.interpolate bicubic, align_corners=False,
// scale given): MODE 0: x0.5, 1: x2, 2: x4.  i0 = floor(src), t = src - i0.
template <int MODE>
__device__ __forceinline__ void resample_plan(int o, int& i0, float w[4]) {
    if (MODE == 0) { i0 = 2 * o; cubic_w(0.5f, w); }
    else if (MODE == 1) {
        int m = o >> 1;
        if (o & 1) { i0 = m; cubic_w(0.25f, w); } else { i0 = m - 1; cubic_w(0.75f, w); }
    } else {
        int m = o >> 2, r = o & 3;
        if (r == 0) { i0 = m - 1; cubic_w(0.625f, w); }
        else if (r == 1) { i0 = m - 1; cubic_w(0.875f, w); }
        else if (r == 2) { i0 = m; cubic_w(0.125f, w); }
        else { i0 = m; cubic_w(0.375f, w); }
    }
}
// resampled value at output (oy,ox) of a plane [hi][wi]
template <int MODE>
__device__ __forceinline__ float resample_at(const float* __restrict__ p, int hi, int wi, int oy, int ox) {
    int iy0, ix0;
    float wy[4], wx[4];
    resample_plan<MODE>(oy, iy0, wy);
    resample_plan<MODE>(ox, ix0, wx);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        int yy = clampi(iy0 - 1 + a, 0, hi - 1);
        float r = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) r += wx[b] * p[yy * wi + clampi(ix0 - 1 + b, 0, wi - 1)];
        acc += wy[a] * r;
    }
    return acc;
}

template <int E>
__device__ __forceinline__ void ln_stats(const float (&x)[E], float& mu, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) s += x[i];
    mu = s * (1.0f / E);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) { float d = x[i] - mu; v += d * d; }
    rstd = __builtin_amdgcn_rsqf(v * (1.0f / E) + LG_EPS);   // v_rsq_f32 (1 ulp) instead of the IEEE sqrt + divide sequence
}

// sum over an aligned group of N = 4, 8 or 16 consecutive lanes (the lanes of one pixel in the lane = (pixel, channel quad) kernels):
// DPP quad_perm moves inside the quad (VALU only), one xor-4 shuffle on top for N = 8
template <int N>
__device__ __forceinline__ float lane_group_sum(float v) {
    static_assert(N == 4 || N == 8 || N == 16, "group of 4, 8 or 16 lanes");
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    if (N >= 8) v += __shfl_xor(v, 4);
    if (N >= 16) v += __shfl_xor(v, 8);
    return v;
}
// counter-hash RNG for dropout (keep probability 0.9, nn.Dropout(0.1) of LGT.py:197): a two-round 32-bit multiply-xorshift of a counter,
// keyed by both halves of the 64-bit per-(stage, block) seed.  Forward and backward call the same function, so the masks agree by construction.
// Round 5: ONE hash decides a PAIR of elements -- the hash of counter idx >> 1; element 2 j is dropped when its low 16 bits are below 6554,
// element 2 j + 1 when its high 16 bits are (6554 / 65536 = 0.100006).  Half the hashes per pixel: the hash was 10 % of k_attn_m<8> and 13 % of
// k_attn_m<16>.  dropout_scale() is the per-element view of the same function; dropout_scale2() returns both decisions of a pair.
#define LG_DROP_T16 6554u
__device__ __forceinline__ uint32_t dropout_hash(uint64_t seed, uint64_t pair) {
    uint32_t x = (uint32_t)pair + (uint32_t)seed;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= (uint32_t)(seed >> 32) ^ (uint32_t)(pair >> 32);
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t idx) {
    const uint32_t x = dropout_hash(seed, idx >> 1);
    return ((idx & 1) ? (x >> 16) : (x & 0xffffu)) < LG_DROP_T16 ? 0.0f : (1.0f / 0.9f);
}
__device__ __forceinline__ void dropout_scale2(uint64_t seed, uint64_t idx_even, float& s0, float& s1) {   // elements idx_even, idx_even + 1
    const uint32_t x = dropout_hash(seed, idx_even >> 1);
    s0 = (x & 0xffffu) < LG_DROP_T16 ? 0.0f : (1.0f / 0.9f);
    s1 = (x >> 16) < LG_DROP_T16 ? 0.0f : (1.0f / 0.9f);
}
// Small parameter tensors (weights / biases of the per-pixel matvecs) are staged in LDS once per workgroup and read back as
// broadcasts: indexed straight from global memory inside a per-pixel loop they compile to long chains of dependent vector
// loads, one s_waitcnt each.  Call from ALL threads of the block; __syncthreads() before use.
__device__ __forceinline__ void lds_stage(float* dst, const float* __restrict__ src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
// Several small arrays at once, with EVERY load in flight before the first store: one lds_stage call after another is one dependent L2 round
// trip after another (each loop's load sits in an exec-masked branch, whose join waits with vmcnt(0)) -- 3 to 5 of them at the start of
// kernels that run 12 - 40 us.  Counts are compile-time; loads are unconditional from clamped indices; NT = the block size.
template <int NT, int N>
__device__ __forceinline__ void lds_stage_ld(float (&v)[(N + NT - 1) / NT], const float* __restrict__ src) {
#pragma unroll
    for (int k = 0; k < (N + NT - 1) / NT; ++k) { const int i = k * NT + (int)threadIdx.x; v[k] = src[i < N ? i : N - 1]; }
}
template <int NT, int N>
__device__ __forceinline__ void lds_stage_st(float* dst, const float (&v)[(N + NT - 1) / NT]) {
#pragma unroll
    for (int k = 0; k < (N + NT - 1) / NT; ++k) { const int i = k * NT + (int)threadIdx.x; if (i < N) dst[i] = v[k]; }
}
#endif  // __HIPCC__
