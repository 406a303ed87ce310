// Power-of-two operand scales of the fused feed_forward kernels' f16-pair arithmetic (split_bf16.h, NP = 2) -- reference
// models/common/LGT.py:91-109 has no counterpart: this is bookkeeping of the number format.
//
// An f16 pair (hi, lo) carries 24 significant bits only while the value sits in f16's exponent range (|v| < 2^16, and down to 2^-17 of
// that before the low piece starts to lose bits), so every MFMA operand of the forward FFN is multiplied by a power of two -- exact, and
// undone exactly behind the accumulator -- chosen from a bound that holds for EVERY input, computed here from the block's weights:
//     |LN(x)_k|   <= sqrt(e) |gamma_k| + |beta_k|                         =: y_k          (LayerNorm output: |x^_k| <= sqrt(e - 1))
//     |h1_c|      <= |b1_c| + sum_k |W1_ck| y_k,   |gelu(h)| <= max(|h|, 0.17)          =: a1_c
//     |h2_c|      <= |b2_c| + sum_k |W2_ck| a1_k
//     |h3_c|      <= |dwb_c| + (sum_t |dww_ct|) |h2_c|                                 =: a3_c (same max)
// The bounds are loose by 2^3 .. 2^10 against typical activations (sums of absolute values); f16's 30 binades absorb that: an operand
// scaled so that its BOUND is below 2^15 keeps elements down to 2^-17 of the bound exact to 2^-24 and smaller ones to an absolute 2^-25
// (2^-40 of the bound).  One workgroup per block, all blocks of a forward call in ONE launch.
// out[job][8] = { s_x, s_a1, s_a3, s_w1, s_w2, s_w3, 0, 0 }: scales of LN(x), gelu(h1), gelu(h3) and the three weights.
#include "kernels.h"

namespace {
__device__ __forceinline__ float block_max(float v, float* red) {   // 256 threads: wave shuffles, then four values through LDS (round 6: the eight-level
#pragma unroll                                                        // barrier tree made this one-workgroup-per-block kernel a 16 us latency chain)
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    __syncthreads();                      // the previous call's readers are done
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float pow2_below(float bound) {   // the power of two s with bound * s in [2^14, 2^15)   (bound = 0: 2^15)
    // exponent clamped to [-100, 60]: a vanishing bound (weights of 1e-35) would otherwise make the PRODUCT of two scales overflow
    // (S1 = s_x s_w1 = inf, 1 / S1 = 0, NaN out), and a non-finite weight gives an arbitrary frexp exponent.  With the cap every scale product
    // stays below 2^120; operands below 2^-74 flush to zero in f16 -- 2^-50 of anything that can matter beside O(1) activations
    int ex = 15 - __builtin_amdgcn_frexp_expf(bound);
    ex = ex < -100 ? -100 : (ex > 60 ? 60 : ex);
    return __builtin_amdgcn_ldexpf(1.0f, ex);
}
}  // namespace

// blockIdx.y = 1: the LOCAL MIXER's static scales of the same block (round 6: to_qkv and Q K^T of k_attn_m on f16 pairs), from bounds of the same kind:
//     |LN1(x)_k| <= sqrt(e) |gamma_k| + |beta_k| =: y_k  (local half: k < e / 2);   |q_c|, |k_c| <= |b_c| + sum_k |W_ck| y_k
// attn_out[job][4] = { s_y, s_w, s_q, s_k }: scales of LN1(x), the to_qkv weights, q * D^-1/2 log2(e) and k
__global__ __launch_bounds__(256) void k_ffn_scales(FfnPrepTable tab, float* __restrict__ out, float* __restrict__ attn_out) {
    const FfnPrepJob& j = tab.j[blockIdx.x];
    const int e = j.e, n1 = 4 * e, t = threadIdx.x;
    __shared__ float ylim[64], a1lim[256], red[256];
    if (blockIdx.y == 1) {
        const int hc = e / 2, d = hc / 2;
        float by = 0.f;
        if (t < hc) { by = sqrtf((float)e) * fabsf(j.ln1g[t]) + fabsf(j.ln1b[t]); ylim[t] = by; }
        const float By = block_max(by, red);
        float bq = 0.f, bk = 0.f, mw = 0.f;
        if (t < 2 * hc) {      // rows [0, hc): q channels, [hc, 2 hc): k channels of to_qkv
            float b = fabsf(j.qkvb[t]);
            for (int k = 0; k < hc; ++k) b += fabsf(j.qkvw[(size_t)t * hc + k]) * ylim[k];
            if (t < hc) bq = b; else bk = b;
        }
        for (int i = t; i < 3 * hc * hc; i += 256) mw = fmaxf(mw, fabsf(j.qkvw[i]));
        const float Bq = block_max(bq, red) * (1.44269504088896340736f / sqrtf((float)d));
        const float Bk = block_max(bk, red);
        const float MW = block_max(mw, red);
        if (t == 0) {
            float* o = attn_out + (size_t)blockIdx.x * 4;
            o[0] = pow2_below(By); o[1] = pow2_below(MW); o[2] = pow2_below(Bq); o[3] = pow2_below(Bk);
        }
        return;
    }
    float by = 0.f;
    if (t < e) { by = sqrtf((float)e) * fabsf(j.ln2g[t]) + fabsf(j.ln2b[t]); ylim[t] = by; }
    const float By = block_max(by, red);
    float bh1 = 0.f, mw1 = 0.f;
    if (t < n1) {
        bh1 = fabsf(j.b1[t]);
        for (int k = 0; k < e; ++k) { const float w = fabsf(j.w1[(size_t)t * e + k]); bh1 += w * ylim[k]; mw1 = fmaxf(mw1, w); }
        bh1 = fmaxf(bh1, 0.17f);
        a1lim[t] = bh1;
    }
    const float A1 = block_max(bh1, red);
    const float MW1 = block_max(mw1, red);
    float bh3 = 0.f, mw2 = 0.f;
    if (t < n1) {
        float bh2 = fabsf(j.b2[t]);
        for (int k = 0; k < n1; ++k) { const float w = fabsf(j.w2[(size_t)t * n1 + k]); bh2 += w * a1lim[k]; mw2 = fmaxf(mw2, w); }
        float taps = 0.f;
        for (int k = 0; k < 9; ++k) taps += fabsf(j.dww[t * 9 + k]);
        bh3 = fmaxf(fabsf(j.dwb[t]) + taps * bh2, 0.17f);
    }
    const float A3 = block_max(bh3, red);
    const float MW2 = block_max(mw2, red);
    float mw3 = 0.f;
    for (int i = t; i < e * n1; i += 256) mw3 = fmaxf(mw3, fabsf(j.w3[i]));
    const float MW3 = block_max(mw3, red);
    if (t == 0) {
        float* o = out + (size_t)blockIdx.x * 8;
        o[0] = pow2_below(By); o[1] = pow2_below(A1); o[2] = pow2_below(A3);
        o[3] = pow2_below(MW1); o[4] = pow2_below(MW2); o[5] = pow2_below(MW3);
        o[6] = 0.f; o[7] = 0.f;
    }
}

int launch_ffn_scales(int n, const FfnPrepJob* jobs, float* out, hipStream_t s, float* attn_out) {
    for (int j0 = 0; j0 < n; j0 += LG_MAX_FFN_PREP_JOBS) {   // (the table travels as a kernel argument: at most LG_MAX_FFN_PREP_JOBS blocks per launch)
        const int m = n - j0 < LG_MAX_FFN_PREP_JOBS ? n - j0 : LG_MAX_FFN_PREP_JOBS;
        FfnPrepTable tab;
        for (int i = 0; i < LG_MAX_FFN_PREP_JOBS; ++i) tab.j[i] = jobs[j0 + (i < m ? i : 0)];
        for (int i = 0; i < m; ++i)
            if (tab.j[i].e != 16 && tab.j[i].e != 32 && tab.j[i].e != 64) { lg_set_error("ffn_scales: e=%d unsupported", tab.j[i].e); return -2; }
        k_ffn_scales<<<dim3(m, attn_out ? 2 : 1), 256, 0, s>>>(tab, out + (size_t)j0 * 8, attn_out ? attn_out + (size_t)j0 * 4 : nullptr);
        LG_CHECK_LAUNCH();
    }
    return 0;
}
