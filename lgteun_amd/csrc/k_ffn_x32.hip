// k_ffn_x32: the fused feed_forward half-block (reference models/common/LGT.py:91-109 + pre_norm / residual :45-61) at e = 32
// (hidden width 128: level 1 of the 4-band net, level 0 of the 8-band net) -- same strip walk and the same fp32-equivalent split-bf16
// GEMMs as k_ffn_xs (k_ffn_x.hip, split_bf16.h), sized for one 512-thread workgroup per CU:
//   * 8 waves, wave w owns hidden channels [16 w, 16 w + 16) of h1 / h2 for every pixel of a 48-pixel chunk (W1 / W2 fragments
//     register-resident: 12 + 48 VGPRs), and tile row w of the 8 x 16 output tile in the depthwise / GEMM3 phase;
//   * LDS 150 KB: h2 ring [10][18][132] fp32 (95 KB) | gelu(h1) pieces [3][48][128] bf16, 16-byte chunks XOR-swizzled by the pixel
//     index (conflict-free fragment reads without padding) | LN(x) pieces [2][3][48][32] ; the per-wave gelu(h3) pieces alias the
//     last two and hold one 64-channel K-half at a time ([8][3][16][64]);
//   * weight fragments arrive PRE-SPLIT: k_split_w (one tiny launch in front) writes the three bf16 pieces of W1 / W2 / W3 in
//     fragment order into the caller's workspace, so the kernel's weight loads are coalesced 16-byte reads and W3 (96 VGPRs if
//     resident) is simply re-read per K-half from L1 / L2.
// The f32-MFMA kernel it replaces (k_ffn_fused<32>) ran one wave per SIMD at MFMA cycles + VALU cycles.
#include <type_traits>
#include "kernels.h"

#include "hstore.h"
#include "split_bf16.h"

namespace {

#ifndef LG_X32_ALT
#define LG_X32_ALT 0   // 1 (round-4 experiment, parity-green, NOT kept): 32-pixel chunks, TWO gelu(h1) images; the two wave groups of the workgroup (waves
                       // 0 .. 3 / 4 .. 7 = the two waves of every SIMD) run GEMM2 of chunk c and GEMM1 + GELU of chunk c + 1 in OPPOSITE order, so
                       // that one wave's matrix burst could run under the other's vector burst, and a chunk costs one barrier.  Same-box A/B at c3
                       // (average launch of the fused FFN, all variants): 316.6 us (310.5 with the chunk loop unrolled and the x vectors requested two chunks ahead) against 280.5 - 282 for round 2's form (48-pixel chunks, every wave in
                       // the same phase, two barriers per chunk) -- with ONE wave of a SIMD in GEMM2 its two dependent accumulation chains do not
                       // fill the matrix pipe (six chains of two waves did), and five chunks per step pay the pipeline's fill and drain
#endif
constexpr int E = 32, N1 = 128, TX = 16, HX = 18, TY = 8, RING = 10, LDR = 132, CH = LG_X32_ALT ? 32 : 48, NPB = CH / 16;
constexpr int NA2 = LG_X32_ALT ? 2 : 1;          // gelu(h1) images
constexpr int A2_HALVES = 3 * CH * N1;           // halves per image (CH = 48: 36,864 B; 32: 24,576 B), rows of 256 B, swizzled
constexpr int XA_SLOT = 3 * CH * E;              // halves per LN(x) slot
constexpr int G3_WAVE = 3 * 16 * 64;             // 3072 halves per wave (one K-half), rows of 128 B, swizzled
constexpr size_t LDS_BYTES = (size_t)RING * HX * LDR * 4 + (size_t)(NA2 * A2_HALVES + 2 * XA_SLOT) * 2;
static_assert(8 * G3_WAVE <= NA2 * A2_HALVES + 2 * XA_SLOT, "gelu(h3) pieces must fit in the aliased region");
static_assert(LDS_BYTES + 4096 <= 160 * 1024, "LDS budget (dynamic + the static parameter / mask arrays)");
constexpr int NF_W1 = 8, NF_W2 = 32, NF_W3 = 8;  // 16 x 32 fragments: W1 [128][32], W2 [128][128], W3 [32][128]

__device__ __forceinline__ bf16x8_t lds_x8(const uint16_t* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(p)); }
__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return v;
}
// W [rows][K] fp32 -> fragments (mb, kb) of 16 x 32, three bf16 pieces each, in the order a wave loads them
__device__ __forceinline__ void split_w_body(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                                             u32x4_t* __restrict__ out, int e, int np, const float* __restrict__ scales) {
    const int n1 = 4 * e, kb1 = e / 32, kb2 = n1 / 32;
    const int nf1 = (n1 / 16) * kb1, nf2 = (n1 / 16) * kb2, nf3 = (e / 16) * kb2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int f = t >> 6, lane = t & 63, r = lane & 15, g = lane >> 4;
    if (f >= nf1 + nf2 + nf3) return;
    const float* W;
    int K, fl;
    if (f < nf1) { W = w1; K = e; fl = f; }
    else if (f < nf1 + nf2) { W = w2; K = n1; fl = f - nf1; }
    else { W = w3; K = n1; fl = f - nf1 - nf2; }
    const int kbn = K / 32, mb = fl / kbn, kb = fl - mb * kbn;
    const float* src = W + (size_t)(mb * 16 + r) * K + kb * 32 + 8 * g;
    const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
    float a[4] = {lo.x, lo.y, lo.z, lo.w}, b[4] = {hi.x, hi.y, hi.z, hi.w};
    u32x2_t a1, a2, a3, b1, b2, b3;
    if (np == 3) {
        split3_x4(a, a1, a2, a3);
        split3_x4(b, b1, b2, b3);
    } else if (np == 2) {   // f16 pairs of W * (the matrix's power-of-two scale, k_ffn_prep.hip)
        const float sw = scales[f < nf1 ? 3 : (f < nf1 + nf2 ? 4 : 5)];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] *= sw; b[i] *= sw; }
        split2_x4(a, a1, a2);
        split2_x4(b, b1, b2);
        a3 = a1; b3 = b1;
    } else {   // plain-bf16 mode: piece 0 rounded to nearest, the others unused
        split_x4<1>(a, a1, a2, a3);
        split_x4<1>(b, b1, b2, b3);
    }
    out[(f * 3 + 0) * 64 + lane] = (u32x4_t){a1.x, a1.y, b1.x, b1.y};
    out[(f * 3 + 1) * 64 + lane] = (u32x4_t){a2.x, a2.y, b2.x, b2.y};
    out[(f * 3 + 2) * 64 + lane] = (u32x4_t){a3.x, a3.y, b3.x, b3.y};
}
__global__ __launch_bounds__(256) void k_split_w(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ w3,
                                                 u32x4_t* __restrict__ out, int e, int np, const float* __restrict__ scales) {
    split_w_body(w1, w2, w3, out, e, np, scales);
}
// every e >= 32 block of a forward call in one launch: blockIdx.y = job (round 5: one k_split_w per FFN launch was 5 / 20 / 40 launches of 4.8 us per step)
__global__ __launch_bounds__(256) void k_split_w_jobs(SplitWTable tab) {
    const SplitWJob& j = tab.j[blockIdx.y];
    split_w_body(j.w1, j.w2, j.w3, reinterpret_cast<u32x4_t*>(j.out), j.e, j.np, j.scales);
}

// SAVE: 0 nothing; 1 gelu(h1), gelu'(h1), h2 and gelu(h3) / gelu'(h3) or the pre-activation h3 (a2.g3s null); 2 h2 and h3 only (the backward
// re-computes h1 from x: k_ffn1_bwd_xs<32>) -- a compile-time mode: as a run-time test of a1.a1s inside the unrolled stage the saving launch
// spilled and ran 570 us instead of 341
template <int SAVE, int NP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ffn_x32(Ffn1Args a1, Ffn2Args a2, const u32x4_t* __restrict__ wsp, int tiles_x, int strips_y, int nstrips, int SH) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* ring = reinterpret_cast<float*>(smem_raw);                                   // [RING*HX][LDR]
    uint16_t* A2 = reinterpret_cast<uint16_t*>(smem_raw + (size_t)RING * HX * LDR * 4);   // [NA2][3][CH][N1], chunk-swizzled
    uint16_t* XA = A2 + NA2 * A2_HALVES;                                                // [2][3][CH][E]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    uint16_t* G3 = A2 + wave * G3_WAVE;                                                 // [3][16][64], aliases A2 / XA
    const int h = a2.h, w = a2.w;
    __shared__ __attribute__((aligned(16))) float sPar[5 * E];
    __shared__ __attribute__((aligned(16))) float sMask[4][CH];   // halo-pixel masks of up to four chunks in flight (slot = chunk & 3; round 2's form uses two)
    float* sLn2g = sPar;            float* sLn2b = sPar + E;
    float* sN1g = sPar + 2 * E;     float* sN1b = sPar + 3 * E;
    float* sB3 = sPar + 4 * E;
    for (int i = threadIdx.x; i < E; i += 512) {
        sLn2g[i] = a1.ln2g[i]; sLn2b[i] = a1.ln2b[i]; sB3[i] = a2.b3[i];
        sN1g[i] = a2.g ? a2.n1g[i] : 0.f; sN1b[i] = a2.g ? a2.n1b[i] : 0.f;
    }
    constexpr bool BF = (NP == 1);                    // plain-bf16 mode: saved activations stored as bf16 (hstore.h)
    // NP = 2 (f16 pairs): the operand scales of k_ffn_prep.hip and how they leave again -- k_ffn_x.hip has the scheme
    float sx = 1.f, sa1 = 1.f, sa3 = 1.f, sw1 = 1.f, sw2 = 1.f, sw3 = 1.f;
    if (NP == 2) { sx = a1.scales[0]; sa1 = a1.scales[1]; sa3 = a1.scales[2]; sw1 = a1.scales[3]; sw2 = a1.scales[4]; sw3 = a1.scales[5]; }
    const float S1 = sx * sw1, S2 = sa1 * sw2, S3 = sa3 * sw3;
    const float inv1 = 1.0f / S1, g1c = 0.70710678118654752440f / S1, g1h = 0.5f * sa1 / S1, g3h = 0.5f * sa3, inv2 = 1.0f / S2, inv3 = 1.0f / S3;   // (powers of two: exact)
    const int c0 = wave * 16 + 4 * g;                 // first of the four h1 / h2 channels this lane holds after GEMM1 / GEMM2
    float4 b1v = *reinterpret_cast<const float4*>(a1.b1 + c0);
    float4 b2v = *reinterpret_cast<const float4*>(a1.b2 + c0);
    if (NP == 2) { b1v = make_float4(b1v.x * S1, b1v.y * S1, b1v.z * S1, b1v.w * S1); b2v = make_float4(b2v.x * S2, b2v.y * S2, b2v.z * S2, b2v.w * S2); }
    const WFrag32 w1f = ld_wfrag<NP>(wsp, wave);                                   // W1 rows 16 w .., K = 32
    WFrag32 w2f[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) w2f[kb] = ld_wfrag<NP>(wsp, NF_W1 + wave * 4 + kb);
    const u32x4_t* w3p = wsp + (size_t)(NF_W1 + NF_W2) * 3 * 64;             // W3 fragments (mb, kb): f = mb * 4 + kb
    // LayerNorm phase: thread t < 8 CH = (chunk pixel t / 8, channel quad t % 8)
    const int lpx = threadIdx.x >> 3, lq = threadIdx.x & 7;
    const bool ln_thread = threadIdx.x < 8 * CH;
    // depthwise phase: lane = (pixel slot lane / 16, channel quad q16 of the current K-half)
    const int q16 = lane & 15;
    __syncthreads();
    float4 lng = *reinterpret_cast<const float4*>(sLn2g + 4 * lq), lnb = *reinterpret_cast<const float4*>(sLn2b + 4 * lq);
    if (NP == 2) { lng = make_float4(lng.x * sx, lng.y * sx, lng.z * sx, lng.w * sx); lnb = make_float4(lnb.x * sx, lnb.y * sx, lnb.z * sx, lnb.w * sx); }

#pragma unroll 1
    for (int strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
    int t = strip;
    const int tx_i = t % tiles_x;
    t /= tiles_x;
    const int sy = t % strips_y;
    const long b = t / strips_y;
    const int x0 = tx_i * TX, Y0 = sy * SH, Yend = min(Y0 + SH, h);

    auto ln_fetch = [&](int ya, int npx, int c, float4& xv, bool& in) {
        const int m = c * CH + lpx;
        const int hy = m / HX, hx = m - hy * HX;
        const int y = ya + hy, x = x0 + hx - 1;
        in = ln_thread && (m < npx) && y >= 0 && y < h && x >= 0 && x < w;
        xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) xv = *reinterpret_cast<const float4*>(a1.x + ((b * h + y) * (long)w + x) * E + 4 * lq);
    };
    // LayerNorm over the 32 channels of a pixel = 8 consecutive lanes; pieces -> XA[slot], halo mask -> sMask[mslot]
    auto ln_store = [&](int slot, int mslot, const float4& xv, bool in) {
        if (!ln_thread) return;
        float s = quad_sum((xv.x + xv.y) + (xv.z + xv.w));
        s += __shfl_xor(s, 4);
        const float mu = s * (1.0f / E);
        const float d0 = xv.x - mu, d1 = xv.y - mu, d2 = xv.z - mu, d3 = xv.w - mu;
        float v = quad_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
        v += __shfl_xor(v, 4);
        const float rstd = __builtin_amdgcn_rsqf(v * (1.0f / E) + LG_EPS);
        const float m_ = in ? 1.0f : 0.0f;
        const float yv[4] = {(d0 * rstd * lng.x + lnb.x) * m_, (d1 * rstd * lng.y + lnb.y) * m_, (d2 * rstd * lng.z + lnb.z) * m_,
                             (d3 * rstd * lng.w + lnb.w) * m_};
        u32x2_t q1, q2, q3;
        split_x4<NP>(yv, q1, q2, q3);
        uint16_t* dst = XA + slot * XA_SLOT + lpx * E + 4 * lq;
        *reinterpret_cast<u32x2_t*>(dst) = q1;
        if (NP >= 2) *reinterpret_cast<u32x2_t*>(dst + CH * E) = q2;
        if (NP == 3) *reinterpret_cast<u32x2_t*>(dst + 2 * CH * E) = q3;
        if (lq == 0) sMask[mslot][lpx] = NP == 2 ? m_ * inv2 : m_;   // NP = 2: the mask also takes S2 out of GEMM2's accumulator
    };
    // ---- the two stages of a chunk.  stage1: GEMM1 (K = 32: h1[16 w .. +15][CH pixels] = W1 LN(x)) + GELU + split -> gelu(h1) image `img`;
    //      stage2: GEMM2 (K = 128: h2 = W2 gelu(h1)) from image `img` -> masked -> ring (+ saves)
    auto stage1 = [&](int ya, int npx, int c, int xslot, int img) {
        f32x4_t acc[NPB];
        const uint16_t* xa = XA + xslot * XA_SLOT + r * E + 8 * g;
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
            acc[pb] = (f32x4_t){b1v.x, b1v.y, b1v.z, b1v.w};
            const uint16_t* p = xa + pb * 16 * E;
            mfma_np32<NP>(acc[pb], w1f, lds_x8(p), lds_x8(p + CH * E), lds_x8(p + 2 * CH * E));
        }
        // GELU, split, -> image: logical 16-byte chunk 2 w + g / 2 of the pixel's row, stored at chunk ^ (pixel & 15)
        uint16_t* A2i = A2 + img * A2_HALVES;
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
            float av[4];
            if (SAVE == 1) {
                const int m = c * CH + pb * 16 + r;
                const int hy = m / HX, hx = m - hy * HX;
                const int y = ya + hy, x = x0 + hx - 1;
                const bool inner = m < npx && hx >= 1 && hx <= TX && x < w && y >= Y0 && y < Yend;
                const long prow = ((b * h + y) * (long)w + x) * N1 + c0;
                lg_v2f a01, a23, g01, g23;
                const float us = NP == 2 ? inv1 : 1.0f;   // NP = 2, saving launch: gelu / gelu' of the TRUE h1 (what the backward reads), scaled afterwards
                gelu2_both_t<NP == 1>((lg_v2f){acc[pb][0] * us, acc[pb][1] * us}, a01, g01);
                gelu2_both_t<NP == 1>((lg_v2f){acc[pb][2] * us, acc[pb][3] * us}, a23, g23);
                av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                if (inner) {
                    HS<BF>::st4_nt(a1.a1s, prow, make_float4(av[0], av[1], av[2], av[3]));
                    HS<BF>::st4_nt(a1.g1s, prow, make_float4(g01.x, g01.y, g23.x, g23.y));
                }
                if (NP == 2) { av[0] *= sa1; av[1] *= sa1; av[2] *= sa1; av[3] *= sa1; }
            } else {
                const lg_v2f a01 = NP == 2 ? gelu2_scaled((lg_v2f){acc[pb][0], acc[pb][1]}, g1c, g1h) : gelu2_t<NP == 1>((lg_v2f){acc[pb][0], acc[pb][1]});
                const lg_v2f a23 = NP == 2 ? gelu2_scaled((lg_v2f){acc[pb][2], acc[pb][3]}, g1c, g1h) : gelu2_t<NP == 1>((lg_v2f){acc[pb][2], acc[pb][3]});
                av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
            }
            u32x2_t q1, q2, q3;
            split_x4<NP>(av, q1, q2, q3);
            const int px = pb * 16 + r;
            uint16_t* dst = A2i + px * N1 + (((2 * wave + (g >> 1)) ^ (px & 15)) << 3) + 4 * (g & 1);
            *reinterpret_cast<u32x2_t*>(dst) = q1;
            if (NP >= 2) *reinterpret_cast<u32x2_t*>(dst + CH * N1) = q2;
            if (NP == 3) *reinterpret_cast<u32x2_t*>(dst + 2 * CH * N1) = q3;
        }
    };
    auto stage2 = [&](int ya, int npx, int c, int mslot, int img, int ring0) {
        f32x4_t acc[NPB];
        const uint16_t* A2i = A2 + img * A2_HALVES;
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
            acc[pb] = (f32x4_t){b2v.x, b2v.y, b2v.z, b2v.w};
            const int px = pb * 16 + r;
            const uint16_t* row = A2i + px * N1;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const uint16_t* p = row + (((4 * kb + g) ^ (px & 15)) << 3);
                mfma_np32<NP>(acc[pb], w2f[kb], lds_x8(p), lds_x8(p + CH * N1), lds_x8(p + 2 * CH * N1));
            }
        }
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
            const int m = c * CH + pb * 16 + r;
            const float mk = sMask[mslot][pb * 16 + r];
            const float4 hh = make_float4(acc[pb][0] * mk, acc[pb][1] * mk, acc[pb][2] * mk, acc[pb][3] * mk);
            if (SAVE) {
                const int hy = m / HX, hx = m - hy * HX;
                const int y = ya + hy, x = x0 + hx - 1;
                const bool inner = m < npx && hx >= 1 && hx <= TX && x < w && y >= Y0 && y < Yend;
                if (inner) HS<BF>::st4_nt(a1.h2, ((b * h + y) * (long)w + x) * N1 + c0, hh);
            }
            int rp = ring0 + m;
            rp = rp >= RING * HX ? rp - RING * HX : rp;
            if (m < npx) *reinterpret_cast<float4*>(ring + rp * LDR + c0) = hh;
        }
    };

#if LG_X32_ALT
    // dh.. h2 of halo rows [ya, ya + nr) -> ring.  pre0 / pre1: the x vectors of chunks 0 and 1, requested by the caller ahead of time.
    // after_first: runs right behind the consumption of the prefetched vectors (the compiler's wait there is vmcnt(0): loads issued before it
    // would be waited for on the spot -- k_ffn_x.hip).
    // Chunk pipeline: image c & 1 holds gelu(h1) of chunk c.  In iteration c the waves of group A (0 .. 3) run stage2(c) and THEN stage1(c + 1),
    // the waves of group B (4 .. 7) the other way round: the two waves of a SIMD (w and w + 4) are never in the same kind of burst, GEMM2's 48
    // MFMAs of one run under the other's GELU / splitting, and a chunk ends in ONE barrier.  LN(x) of chunk c + 2 is normalised in iteration c
    // (its slot c & 1 was read by stage1(c) in iteration c - 1; its mask slot (c + 2) & 3 is not the one stage2(c) reads).
    auto compute_rows = [&](int ya, auto nr_c, float4 pre0, bool pin0, float4 pre1, bool pin1, auto&& after_first) {
        constexpr int nr = decltype(nr_c)::value, npx = nr * HX, nchunks = (npx + CH - 1) / CH;   // compile-time: the chunk loop unrolls, so the
        const bool grpA = wave < 4;                                                                 // prefetch registers have static names
        __syncthreads();                 // the previous phase's readers of the aliased region (gelu(h3) pieces) are done
        ln_store(0, 0, pre0, pin0);
        if (nchunks > 1) ln_store(1, 1, pre1, pin1);
        after_first();
        // x vectors of chunks 2 .. nchunks - 1: requested TWO chunks ahead (chunk k's vector is normalised at the end of iteration k - 2)
        float4 nxv[nchunks > 2 ? nchunks - 2 : 1];
        bool ninv[nchunks > 2 ? nchunks - 2 : 1];
        if (nchunks > 2) ln_fetch(ya, npx, 2, nxv[0], ninv[0]);
        if (nchunks > 3) ln_fetch(ya, npx, 3, nxv[1], ninv[1]);
        __syncthreads();
        stage1(ya, npx, 0, 0, 0);
        __syncthreads();
        const int ring0 = ((ya - Y0 + 1) % RING) * HX;
#pragma unroll
        for (int c = 0; c < nchunks; ++c) {
            const bool more1 = c + 1 < nchunks, more2 = c + 2 < nchunks;
            if (c + 4 < nchunks) ln_fetch(ya, npx, c + 4, nxv[c + 2], ninv[c + 2]);
            if (grpA) {
                stage2(ya, npx, c, c & 3, c & 1, ring0);
                if (more1) stage1(ya, npx, c + 1, (c + 1) & 1, (c + 1) & 1);
            } else {
                if (more1) stage1(ya, npx, c + 1, (c + 1) & 1, (c + 1) & 1);
                stage2(ya, npx, c, c & 3, c & 1, ring0);
            }
            if (more2) ln_store(c & 1, (c + 2) & 3, nxv[c], ninv[c]);
            __syncthreads();
        }
    };

    {
        float4 pre0, pre1;
        bool pin0, pin1;
        ln_fetch(Y0 - 1, 2 * HX, 0, pre0, pin0);
        ln_fetch(Y0 - 1, 2 * HX, 1, pre1, pin1);
        compute_rows(Y0 - 1, std::integral_constant<int, 2>{}, pre0, pin0, pre1, pin1, [] {});
    }
    float4 pre, pre1;
    bool pin, pin1;
    ln_fetch(Y0 + 1, TY * HX, 0, pre, pin);
    ln_fetch(Y0 + 1, TY * HX, 1, pre1, pin1);
#else
    // after_first: runs right behind the consumption of the prefetched chunk-0 vector (the compiler's wait there is vmcnt(0): loads issued
    // before it would be waited for on the spot -- k_ffn_x.hip)
    auto compute_rows = [&](int ya, auto nr_c, float4 pre, bool pre_in, float4, bool, auto&& after_first) {
        constexpr int nr = decltype(nr_c)::value;
        const int npx = nr * HX, nchunks = (npx + CH - 1) / CH;
        __syncthreads();                 // the previous phase's readers of the aliased region (gelu(h3) pieces) are done
        ln_store(0, 0, pre, pre_in);
        after_first();
        __syncthreads();
        const int ring0 = ((ya - Y0 + 1) % RING) * HX;
        for (int c = 0; c < nchunks; ++c) {
            const int slot = c & 1;
            float4 nx;
            bool nin = false;
            const bool more = c + 1 < nchunks;
            if (more) ln_fetch(ya, npx, c + 1, nx, nin);
            stage1(ya, npx, c, slot, 0);
            __syncthreads();
            stage2(ya, npx, c, slot, 0, ring0);     // (the ring stores follow ln_store in round 2's order; the order of the two is free)
            if (more) ln_store(slot ^ 1, slot ^ 1, nx, nin);
            __syncthreads();
        }
    };

    {
        float4 pre;
        bool pin;
        ln_fetch(Y0 - 1, 2 * HX, 0, pre, pin);
        compute_rows(Y0 - 1, std::integral_constant<int, 2>{}, pre, pin, pre, pin, [] {});
    }
    float4 pre, pre1 = make_float4(0.f, 0.f, 0.f, 0.f);
    bool pin, pin1 = false;
    ln_fetch(Y0 + 1, TY * HX, 0, pre, pin);
#endif
#pragma unroll 1
    for (int y0 = Y0; y0 < Yend; y0 += TY) {
    // wave w owns tile row w; lane (r, g): pixel x0 + r, output channels 16 mb + 4 g .. + 3 (mb = 0, 1)
    const int ty = wave;
    float4 xres[2];
    compute_rows(y0 + 1, std::integral_constant<int, TY>{}, pre, pin, pre1, pin1, [&] {
        const int y = y0 + ty, x = x0 + r;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            xres[mb] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y < Yend && x < w) xres[mb] = *reinterpret_cast<const float4*>(a2.x + ((b * h + y) * (long)w + x) * E + 16 * mb + 4 * g);
        }
    });
    if (y0 + TY < Yend) {   // next step's first chunk(s): in flight during the output phase
        ln_fetch(y0 + TY + 1, TY * HX, 0, pre, pin);
        if (LG_X32_ALT) ln_fetch(y0 + TY + 1, TY * HX, 1, pre1, pin1);
    }
    // ---- output phase: per K-half: dw3x3 over the ring + GELU -> pieces -> GEMM3 partial ; then bias + residual -> y (+ planar LN half)
    {
        const int sbase = (y0 - Y0) % RING;            // ring slot of row y0 - 1
        f32x4_t o[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const float4 b3v = *reinterpret_cast<const float4*>(sB3 + 16 * mb + 4 * g);
            o[mb] = NP == 2 ? (f32x4_t){b3v.x * S3, b3v.y * S3, b3v.z * S3, b3v.w * S3} : (f32x4_t){b3v.x, b3v.y, b3v.z, b3v.w};
        }
#pragma unroll 1
        for (int kh = 0; kh < 2; ++kh) {
            const int qc = 16 * kh + q16;              // channel quad of this lane in this half: channels 4 qc .. 4 qc + 3
            float wq[4][9], bq[4];
            {
                const float* tp = a2.dww + 36 * qc;
                float t36[36];
#pragma unroll
                for (int k4 = 0; k4 < 9; ++k4) {
                    const float4 v = *reinterpret_cast<const float4*>(tp + 4 * k4);
                    t36[4 * k4] = v.x; t36[4 * k4 + 1] = v.y; t36[4 * k4 + 2] = v.z; t36[4 * k4 + 3] = v.w;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int kk = 0; kk < 9; ++kk) wq[u][kk] = t36[9 * u + kk];
                const float4 bv = *reinterpret_cast<const float4*>(a2.dwb + 4 * qc);
                bq[0] = bv.x; bq[1] = bv.y; bq[2] = bv.z; bq[3] = bv.w;
            }
#pragma unroll(SAVE ? 1 : 2)
            for (int it = 0; it < 4; ++it) {
                const int tx = (lane >> 4) + 4 * it;
                float4 acc = make_float4(bq[0], bq[1], bq[2], bq[3]);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    int sl = sbase + ty + dy;
                    sl = sl >= RING ? sl - RING : sl;
                    sl = sl >= RING ? sl - RING : sl;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float4 v = *reinterpret_cast<const float4*>(ring + (sl * HX + tx + dx) * LDR + 4 * qc);
                        acc.x += wq[0][dy * 3 + dx] * v.x; acc.y += wq[1][dy * 3 + dx] * v.y;
                        acc.z += wq[2][dy * 3 + dx] * v.z; acc.w += wq[3][dy * 3 + dx] * v.w;
                    }
                }
                float av[4];
                if (SAVE) {
                    lg_v2f a01, a23, g01, g23;
                    gelu2_both_t<NP == 1>((lg_v2f){acc.x, acc.y}, a01, g01);
                    gelu2_both_t<NP == 1>((lg_v2f){acc.z, acc.w}, a23, g23);
                    av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                    const int y = y0 + ty, x = x0 + tx;
                    if (y < Yend && x < w) {
                        const long off = ((b * h + y) * (long)w + x) * N1 + 4 * qc;
                        if (a2.g3s) {   // five-tensor form: gelu(h3), gelu'(h3)
                            HS<BF>::st4_nt(a2.a3s, off, make_float4(av[0], av[1], av[2], av[3]));
                            HS<BF>::st4_nt(a2.g3s, off, make_float4(g01.x, g01.y, g23.x, g23.y));
                        } else {        // k_ffn_dw_bwd_xs<32> re-evaluates both from the pre-activation
                            HS<BF>::st4_nt(a2.a3s, off, acc);
                        }
                    }
                    if (NP == 2) { av[0] *= sa3; av[1] *= sa3; av[2] *= sa3; av[3] *= sa3; }
                } else {
                    const lg_v2f a01 = NP == 2 ? gelu2_scaled((lg_v2f){acc.x, acc.y}, 0.70710678118654752440f, g3h) : gelu2_t<NP == 1>((lg_v2f){acc.x, acc.y});
                    const lg_v2f a23 = NP == 2 ? gelu2_scaled((lg_v2f){acc.z, acc.w}, 0.70710678118654752440f, g3h) : gelu2_t<NP == 1>((lg_v2f){acc.z, acc.w});
                    av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                }
                u32x2_t q1, q2, q3;
                split_x4<NP>(av, q1, q2, q3);
                // row of 64 halves = 8 chunks of 16 bytes: logical chunk q16 / 2, stored at chunk ^ (pixel & 7)
                uint16_t* dst = G3 + tx * 64 + ((((q16 >> 1)) ^ (tx & 7)) << 3) + 4 * (q16 & 1);
                *reinterpret_cast<u32x2_t*>(dst) = q1;
                if (NP >= 2) *reinterpret_cast<u32x2_t*>(dst + 16 * 64) = q2;
                if (NP == 3) *reinterpret_cast<u32x2_t*>(dst + 2 * 16 * 64) = q3;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- GEMM3 partial (K-half: 64): out[32 channels][16 pixels of tile row ty] += W3[:, 64 kh ..] gelu(h3)[64 kh ..]
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const uint16_t* p = G3 + r * 64 + (((4 * kb + g) ^ (r & 7)) << 3);
                const bf16x8_t x1 = lds_x8(p), x2 = lds_x8(p + 16 * 64), x3 = lds_x8(p + 2 * 16 * 64);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const WFrag32 wf = ld_wfrag<NP>(w3p, mb * 4 + 2 * kh + kb);
                    mfma_np32<NP>(o[mb], wf, x1, x2, x3);
                }
            }
            __builtin_amdgcn_wave_barrier();           // G3 is rewritten by the next half
        }
        // ---- epilogue in registers: residual, store, LayerNorm statistics of the next block across the four lane groups
        const int y = y0 + ty, x = x0 + r;
        const bool ok = y < Yend && x < w;
        float ov[8];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const float os = NP == 2 ? inv3 : 1.0f;
            ov[4 * mb + 0] = o[mb][0] * os + xres[mb].x; ov[4 * mb + 1] = o[mb][1] * os + xres[mb].y;
            ov[4 * mb + 2] = o[mb][2] * os + xres[mb].z; ov[4 * mb + 3] = o[mb][3] * os + xres[mb].w;
            if (ok) *reinterpret_cast<float4*>(a2.y + ((b * h + y) * (long)w + x) * E + 16 * mb + 4 * g) =
                        make_float4(ov[4 * mb], ov[4 * mb + 1], ov[4 * mb + 2], ov[4 * mb + 3]);
        }
        if (a2.g) {
            float s = ((ov[0] + ov[1]) + (ov[2] + ov[3])) + ((ov[4] + ov[5]) + (ov[6] + ov[7]));
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const float mu = s * (1.0f / E);
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { ov[i] -= mu; v += ov[i] * ov[i]; }
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const float rstd = __builtin_amdgcn_rsqf(v * (1.0f / E) + LG_EPS);
            if (ok) {      // channels 16..31 (output block mb = 1) = the global-mixer half, planar [B, e/2, h, w]
                const long hw = (long)h * w, sp = (long)y * w + x;
                const float4 ng = *reinterpret_cast<const float4*>(sN1g + 16 + 4 * g), nb = *reinterpret_cast<const float4*>(sN1b + 16 + 4 * g);
                float* dst = a2.g + (b * (E / 2) + 4 * g) * hw + sp;
                dst[0] = ov[4] * rstd * ng.x + nb.x;
                dst[hw] = ov[5] * rstd * ng.y + nb.y;
                dst[2 * hw] = ov[6] * rstd * ng.z + nb.z;
                dst[3 * hw] = ov[7] * rstd * ng.w + nb.w;
            }
        }
    }
    }   // steps of the strip
    }   // strips of this workgroup
}

}   // namespace

size_t ffn_wsplit_bytes(int e) {
    const int n1 = 4 * e, kb1 = (e + 31) / 32, kb2 = n1 / 32;
    const size_t nf = (size_t)(n1 / 16) * kb1 + (size_t)(n1 / 16) * kb2 + (size_t)(e / 16) * kb2;
    return nf * 3 * 64 * 16;
}

// fragments in the order W1 (mb, kb), W2 (mb, kb), W3 (mb, kb); e a multiple of 32
int launch_split_w(const float* w1, const float* w2, const float* w3, void* out, int e, int np, hipStream_t s, const float* scales) {
    if (np == 2 && !scales) { lg_set_error("split_w: f16 pairs need the operand scales"); return -2; }
    const int n1 = 4 * e, kb1 = e / 32, kb2 = n1 / 32;
    const int nfrag = (n1 / 16) * kb1 + (n1 / 16) * kb2 + (e / 16) * kb2;
    k_split_w<<<(nfrag * 64 + 255) / 256, 256, 0, s>>>(w1, w2, w3, reinterpret_cast<u32x4_t*>(out), e, np, scales);
    LG_CHECK_LAUNCH();
    return 0;
}

int launch_split_w_jobs(int n, const SplitWJob* jobs, hipStream_t s) {
    for (int j0 = 0; j0 < n; j0 += LG_MAX_FFN_PREP_JOBS) {
        const int m = n - j0 < LG_MAX_FFN_PREP_JOBS ? n - j0 : LG_MAX_FFN_PREP_JOBS;
        SplitWTable tab;
        int nfmax = 0;
        for (int i = 0; i < LG_MAX_FFN_PREP_JOBS; ++i) tab.j[i] = jobs[j0 + (i < m ? i : 0)];
        for (int i = 0; i < m; ++i) {
            const SplitWJob& q = tab.j[i];
            if (q.e % 32 || (q.np == 2 && !q.scales) || !q.out) { lg_set_error("split_w_jobs: job %d: e=%d np=%d", j0 + i, q.e, q.np); return -2; }
            const int n1 = 4 * q.e, kb1 = q.e / 32, kb2 = n1 / 32, nf = (n1 / 16) * kb1 + (n1 / 16) * kb2 + (q.e / 16) * kb2;
            if (nf > nfmax) nfmax = nf;
        }
        k_split_w_jobs<<<dim3((nfmax * 64 + 255) / 256, m), 256, 0, s>>>(tab);
        LG_CHECK_LAUNCH();
    }
    return 0;
}

int launch_ffn_x32(const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s) {
    ProfScope prof__(LG_K_FFN2, s);
    if (!a1.wsplit) { lg_set_error("ffn_x32: no weight-fragment scratch in the workspace"); return -3; }
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn_x32<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_x32<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_x32<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_x32<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_x32<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_x32<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_x32<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_x32<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_x32<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) { lg_set_error("ffn_x32: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    if (!a1.wsplit_ready) {
        const int rc = launch_split_w(a1.w1, a1.w2, a2.w3, a1.wsplit, E, a1.hbf ? 1 : (a1.scales ? 2 : 3), s, a1.scales);
        if (rc) return rc;
    }
    const int tiles_x = (a2.w + 15) / 16;
    // strip height: the tallest multiple of 8 rows that still yields >= 256 strips (one resident workgroup per CU), at least 16
    int SH = (a2.h + 7) / 8 * 8;
    while (SH > 16 && (long)a2.B * tiles_x * ((a2.h + SH - 1) / SH) < 256) SH = (SH / 2 + 7) / 8 * 8;
    const int strips_y = (a2.h + SH - 1) / SH;
    const int nstrips = a2.B * tiles_x * strips_y;
    const int grid = nstrips < 256 ? nstrips : 256;
    const u32x4_t* wsp = reinterpret_cast<const u32x4_t*>(a1.wsplit);
    const bool save = a1.h2 != nullptr;           // h2 leaves the chip only for the backward
    if (a1.hbf) {   // precision = 'bf16'
        if (save && !a1.a1s) k_ffn_x32<2, 1><<<grid, 512, LDS_BYTES, s>>>(a1, a2, wsp, tiles_x, strips_y, nstrips, SH);
        else if (save) k_ffn_x32<1, 1><<<grid, 512, LDS_BYTES, s>>>(a1, a2, wsp, tiles_x, strips_y, nstrips, SH);
        else k_ffn_x32<0, 1><<<grid, 512, LDS_BYTES, s>>>(a1, a2, wsp, tiles_x, strips_y, nstrips, SH);
    } else if (a1.scales) {   // f16 pairs (three products per block instead of six)
        if (save && !a1.a1s) k_ffn_x32<2, 2><<<grid, 512, LDS_BYTES, s>>>(a1, a2, wsp, tiles_x, strips_y, nstrips, SH);
        else if (save) k_ffn_x32<1, 2><<<grid, 512, LDS_BYTES, s>>>(a1, a2, wsp, tiles_x, strips_y, nstrips, SH);
        else k_ffn_x32<0, 2><<<grid, 512, LDS_BYTES, s>>>(a1, a2, wsp, tiles_x, strips_y, nstrips, SH);
    } else if (save && !a1.a1s) k_ffn_x32<2, 3><<<grid, 512, LDS_BYTES, s>>>(a1, a2, wsp, tiles_x, strips_y, nstrips, SH);
    else if (save) k_ffn_x32<1, 3><<<grid, 512, LDS_BYTES, s>>>(a1, a2, wsp, tiles_x, strips_y, nstrips, SH);
    else k_ffn_x32<0, 3><<<grid, 512, LDS_BYTES, s>>>(a1, a2, wsp, tiles_x, strips_y, nstrips, SH);
    LG_CHECK_LAUNCH();
    return 0;
}
