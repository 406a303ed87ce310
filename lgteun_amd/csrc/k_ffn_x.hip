// k_ffn_xs: the fused feed_forward half-block (reference models/common/LGT.py:91-109 + pre_norm / residual :45-61) at e = 16 with its
// three 1x1-conv GEMMs on the bf16 matrix pipe in fp32-equivalent arithmetic (split_bf16.h: three bf16 pieces per operand, six
// piece products, fp32 accumulation: measured error against fp64 no larger than the fp32 MFMA's).
//
//   y = x + W3 gelu( dw3x3( W2 gelu( W1 LN(x) + b1 ) + b2 ) ) + b3
//
// Same strip walk as k_ffn_strip (k_ffn.hip): a workgroup walks DOWN a 16-column strip in 8-row steps and keeps h2 of the last 10
// halo rows in an LDS ring (fp32: the depthwise conv and both GELUs stay fp32 VALU work), so a step computes only its 8 new halo
// rows (8 x 18 = 144 pixels).  What changed:
//   * the f32-input MFMA shares the SIMD's vector ALU time (the old kernel sat exactly at MFMA cycles + VALU cycles); the bf16
//     matrix pipe is separate, so the GEMMs now hide behind the VALU work (GELU, depthwise conv, LayerNorm, operand splitting);
//   * weights ride on the A side of the MFMA (rows = output channels), pixels on the B side: a lane holds FOUR CONSECUTIVE
//     CHANNELS of one pixel, so h1 / h2 / h3 / y leave its registers as 8- or 16-byte accesses (LDS and, for the saved
//     activations of the live stage, HBM) instead of four 4-byte ones, and masks / ring slots are computed once per pixel;
//   * the output tile needs no LDS round trip: the wave that ran GEMM3 adds bias + residual and does the next block's LayerNorm
//     statistics across its four lane groups in registers.
// Per step and wave: 54 + 108 + 24 MFMAs of 16 cycles (3.0 k matrix cycles, was 6.8 k vector-ALU cycles).
//
// LDS (78.9 KB, two workgroups per CU): ring [10][18][68] fp32 | A2: gelu(h1) pieces [3][48][72] bf16 | XA: LN(x) pieces
// [2][3][48][16] bf16 (double buffer: chunk c+1 is normalised while chunk c's GEMM2 runs) ; the per-wave gelu(h3) pieces
// [4][3][16][72] of the depthwise phase alias A2 + XA.
#include "kernels.h"

#include "hstore.h"
#include "split_bf16.h"

// In-kernel phase stamps (diagnostic build only, -DLG_STAMPS: tools/build_stamps.sh): the four waves of workgroup 0 write
// s_memtime at the phase boundaries of their third step into a buffer nothing else reads; no stamp executes in the product build.
#ifdef LG_STAMPS
__device__ unsigned long long g_ffn_stamps[4 * 32];
#define STAMP(i) do { if (blockIdx.x == 0 && lane == 0 && stamp_on) g_ffn_stamps[wave * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" __attribute__((visibility("default"))) int lg_debug_ffn_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ffn_stamps), sizeof(g_ffn_stamps));
}
#else
#define STAMP(i) do { } while (0)
#endif

#ifndef LG_XS_UNCOND
#define LG_XS_UNCOND 0   // 1: unconditional clamped x loads (the cure of k_ffn_dw_bwd_xs's vmcnt(0) waits); measured here: 114.6 vs 112.2 us per launch, off
#endif
#ifndef LG_XS_SAVE_UNROLL
#define LG_XS_SAVE_UNROLL 1
#endif
namespace {

constexpr int E = 16, N1 = 64, TX = 16, HX = 18, TY = 8, RING = 10, LDR = 68, CH = 48, LDP = 72, CQ = 16;
constexpr int A2_HALVES = 3 * CH * LDP;          // 10368
constexpr int XA_SLOT = 3 * CH * E;              // 2304 halves per slot
constexpr int G3_WAVE = 3 * 16 * LDP;            // 3456 halves per wave
constexpr size_t LDS_BYTES = (size_t)RING * HX * LDR * 4 + (size_t)(A2_HALVES + 2 * XA_SLOT) * 2;
static_assert(4 * G3_WAVE <= A2_HALVES + 2 * XA_SLOT, "gelu(h3) pieces must fit in the aliased region");

// sum over the four lanes of a quad (lanes 4k .. 4k+3) with DPP quad_perm moves: VALU only, no LDS crossbar round trip
__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return v;
}

__device__ __forceinline__ bf16x8_t lds_x8(const uint16_t* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(p)); }
__device__ __forceinline__ s16x4_t lds_x4(const uint16_t* p) { return __builtin_bit_cast(s16x4_t, *reinterpret_cast<const u32x2_t*>(p)); }

// SAVE: 0 nothing; 1 gelu(h1), gelu'(h1), h2, gelu(h3), gelu'(h3) (GELU-free backward: the precision = 'bf16' layout);
// 2 the PRE-ACTIVATIONS h1, h2, h3 only (three tensors instead of five: the backward kernels re-evaluate gelu / gelu' from
// them, bwd_kernels.h `pre`); 3 h2 and h3 only (the backward re-computes h1 from x: k_ffn1_bwd_xs)
template <int SAVE, int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_ffn_xs(Ffn1Args a1, Ffn2Args a2, int tiles_x, int strips_y, int nstrips,
                                                                                       int SH) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* ring = reinterpret_cast<float*>(smem_raw);                                   // [RING*HX][LDR]
    uint16_t* A2 = reinterpret_cast<uint16_t*>(smem_raw + (size_t)RING * HX * LDR * 4);   // [3][CH][LDP]
    uint16_t* XA = A2 + A2_HALVES;                                                      // [2][3][CH][E]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    uint16_t* G3 = A2 + wave * G3_WAVE;                                                 // [3][16][LDP], aliases A2 / XA
    const int h = a2.h, w = a2.w;
    __shared__ __attribute__((aligned(16))) float sPar[5 * E];
    __shared__ __attribute__((aligned(16))) float sMask[2][CH];   // 1 for halo pixels inside the image (dep_conv zero-pads h2)
    float* sLn2g = sPar;            float* sLn2b = sPar + E;
    float* sN1g = sPar + 2 * E;     float* sN1b = sPar + 3 * E;
    float* sB3 = sPar + 4 * E;
    for (int i = threadIdx.x; i < E; i += 256) {
        sLn2g[i] = a1.ln2g[i]; sLn2b[i] = a1.ln2b[i]; sB3[i] = a2.b3[i];
        sN1g[i] = a2.g ? a2.n1g[i] : 0.f; sN1b[i] = a2.g ? a2.n1b[i] : 0.f;
    }
    // ---- NP = 2 (two f16 pieces per operand): every matrix-core operand is scaled by a power of two that puts its PROVEN bound below 2^15
    // (k_ffn_prep.hip), exactly, and the product of the two scales is taken out again behind the accumulator -- by constants that were
    // multiplications already (the GELUs' two immediates, the halo mask, the residual add that becomes an fma): no instruction is added.
    //   GEMM1: (s_x LN(x)) x (s_w1 W1), bias b1 S1 as initial accumulator -> S1 h1, S1 = s_x s_w1;  gelu -> s_a1 gelu(h1)
    //   GEMM2: x (s_w2 W2) -> S2 h2, S2 = s_a1 s_w2; the halo mask carries 1 / S2: the ring (and the saved h2) hold h2 itself
    //   depthwise conv, gelu -> s_a3 gelu(h3);  GEMM3: x (s_w3 W3) -> S3 (...), S3 = s_a3 s_w3;  y = acc / S3 + x
    float sx = 1.f, sa1 = 1.f, sa3 = 1.f, sw1 = 1.f, sw2 = 1.f, sw3 = 1.f;
    if (NP == 2) { sx = a1.scales[0]; sa1 = a1.scales[1]; sa3 = a1.scales[2]; sw1 = a1.scales[3]; sw2 = a1.scales[4]; sw3 = a1.scales[5]; }
    const float S1 = sx * sw1, S2 = sa1 * sw2, S3 = sa3 * sw3;
    const float inv1 = 1.0f / S1, g1c = 0.70710678118654752440f / S1, g1h = 0.5f * sa1 / S1, g3h = 0.5f * sa3, inv2 = 1.0f / S2, inv3 = 1.0f / S3;   // (powers of two: exact)
    // ---- weights: split once, register-resident for every strip of this workgroup
    const int c0 = wave * 16 + 4 * g;                 // first of the four h1 / h2 channels this lane holds after GEMM1 / GEMM2
    float4 b1v = *reinterpret_cast<const float4*>(a1.b1 + c0);
    float4 b2v = *reinterpret_cast<const float4*>(a1.b2 + c0);
    if (NP == 2) { b1v = make_float4(b1v.x * S1, b1v.y * S1, b1v.z * S1, b1v.w * S1); b2v = make_float4(b2v.x * S2, b2v.y * S2, b2v.z * S2, b2v.w * S2); }
    constexpr bool BF = (NP == 1);                    // plain-bf16 mode: saved activations are stored as bf16 too (hstore.h)
    const WFrag16 w1f = NP == 3 ? load_wfrag16(a1.w1 + (size_t)(wave * 16) * E, E, 0) : (NP == 2 ? load_wfrag16_h2(a1.w1 + (size_t)(wave * 16) * E, E, 0, sw1) : load_wfrag16_rne(a1.w1 + (size_t)(wave * 16) * E, E, 0));
    const WFrag32 w2f0 = NP == 3 ? load_wfrag32(a1.w2 + (size_t)(wave * 16) * N1, N1, 0) : (NP == 2 ? load_wfrag32_h2(a1.w2 + (size_t)(wave * 16) * N1, N1, 0, sw2) : load_wfrag32_rne(a1.w2 + (size_t)(wave * 16) * N1, N1, 0));
    const WFrag32 w2f1 = NP == 3 ? load_wfrag32(a1.w2 + (size_t)(wave * 16) * N1, N1, 1) : (NP == 2 ? load_wfrag32_h2(a1.w2 + (size_t)(wave * 16) * N1, N1, 1, sw2) : load_wfrag32_rne(a1.w2 + (size_t)(wave * 16) * N1, N1, 1));
    const WFrag32 w3f0 = NP == 3 ? load_wfrag32(a2.w3, N1, 0) : (NP == 2 ? load_wfrag32_h2(a2.w3, N1, 0, sw3) : load_wfrag32_rne(a2.w3, N1, 0));
    const WFrag32 w3f1 = NP == 3 ? load_wfrag32(a2.w3, N1, 1) : (NP == 2 ? load_wfrag32_h2(a2.w3, N1, 1, sw3) : load_wfrag32_rne(a2.w3, N1, 1));
    // depthwise taps of the lane's four channels (phase P2: lane = (pixel slot lane / 16, channel quad q)): 36 + 4 contiguous
    // floats, re-read (L1 / L2 hits) at the top of every step instead of pinning 40 VGPRs through the GEMM phases
    const int q = lane % CQ;
    // LayerNorm phase: thread t < 192 = (chunk pixel t / 4, channel quad t % 4)
    const int lpx = threadIdx.x >> 2, lq = threadIdx.x & 3;
    const bool ln_thread = threadIdx.x < 4 * CH;
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
#ifdef LG_STAMPS
    bool stamp_on = false;
#endif

    // request the x vector of halo pixel m of the row block starting at ya (zeros outside the image / beyond npx)
    auto ln_fetch = [&](int ya, int npx, int c, float4& xv, bool& in) {
        const int m = c * CH + lpx;
        const int hy = m / HX, hx = m - hy * HX;
        const int y = ya + hy, x = x0 + hx - 1;
        in = ln_thread && (m < npx) && y >= 0 && y < h && x >= 0 && x < w;
#if LG_XS_UNCOND
        // unconditional, from a clamped (always valid) address: a load inside an exec-masked branch makes the compiler wait with vmcnt(0) at
        // the join, i.e. for every load in flight (k_ffn_dwbwd_x.hip; profiles/r03_ffn_bwd_phase_stamps.txt).  ln_store masks the result.
        xv = *reinterpret_cast<const float4*>(a1.x + ((b * h + clampi(y, 0, h - 1)) * (long)w + clampi(x, 0, w - 1)) * E + 4 * lq);
#else
        xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) xv = *reinterpret_cast<const float4*>(a1.x + ((b * h + y) * (long)w + x) * E + 4 * lq);
#endif
    };
    // LayerNorm of the fetched vector over its 16 channels (4 lanes of a quad), split into pieces -> XA[slot]
    auto ln_store = [&](int slot, const float4& xv, bool in) {
        if (!ln_thread) return;
        const float s = quad_sum((xv.x + xv.y) + (xv.z + xv.w));
        const float mu = s * (1.0f / E);
        const float d0 = xv.x - mu, d1 = xv.y - mu, d2 = xv.z - mu, d3 = xv.w - mu;
        const float v = quad_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
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
        if (lq == 0) sMask[slot][lpx] = NP == 2 ? m_ * inv2 : m_;   // NP = 2: the mask also takes S2 out of the accumulator
    };

    // h2 of halo rows [ya, ya + nr) x columns [x0 - 1, x0 + 17) -> ring   (nr = 2: strip prologue, 8: one step).
    // pre / pre_in: chunk 0's x vector, fetched by the caller ahead of time.
    // after_first: run right behind the consumption of the prefetched chunk-0 vector.  gfx950 counts loads AND stores in one in-order
    // counter, and with the exec-masked branches around them the compiler waits with vmcnt(0) there: anything issued before that point
    // (the step's residual-row loads, which need the whole step to land) would be waited for on the spot.
    auto compute_rows = [&](int ya, int nr, float4 pre, bool pre_in, auto&& after_first) {
        const int npx = nr * HX, nchunks = (npx + CH - 1) / CH;
        __syncthreads();                 // the previous phase's readers of the aliased region (gelu(h3) pieces) are done
        STAMP(1);
        ln_store(0, pre, pre_in);
        after_first();
        STAMP(2);
        __syncthreads();
        STAMP(3);
        const int ring0 = ((ya - Y0 + 1) % RING) * HX;
        for (int c = 0; c < nchunks; ++c) {
            const int slot = c & 1;
            // next chunk's x: requested now, normalised while GEMM2 runs
            float4 nx;
            bool nin = false;
            const bool more = c + 1 < nchunks;
            if (more) ln_fetch(ya, npx, c + 1, nx, nin);
            // per pixel block: image coordinates of this lane's pixel (SAVE) -- one pixel per block, not one per value
            long prow[3];
            bool inner[3];
            if (SAVE) {
#pragma unroll
                for (int pb = 0; pb < 3; ++pb) {
                    const int m = c * CH + pb * 16 + r;
                    const int hy = m / HX, hx = m - hy * HX;
                    const int y = ya + hy, x = x0 + hx - 1;
                    inner[pb] = m < npx && hx >= 1 && hx <= TX && x < w && y >= Y0 && y < Yend;
                    prow[pb] = ((b * h + y) * (long)w + x) * N1 + c0;
                }
            }
            // ---- GEMM1 (K = 16): h1[16 w .. +15][48 pixels] = W1 LN(x)
            f32x4_t acc[3];
            const uint16_t* xa = XA + slot * XA_SLOT + r * E + 4 * g;
#pragma unroll
            for (int pb = 0; pb < 3; ++pb) {
                acc[pb] = (f32x4_t){b1v.x, b1v.y, b1v.z, b1v.w};
                const uint16_t* p = xa + pb * 16 * E;
                mfma_np16<NP>(acc[pb], w1f, lds_x4(p), lds_x4(p + CH * E), lds_x4(p + 2 * CH * E));
            }
            // ---- GELU, split, -> A2 (8 bytes per piece and pixel)
#pragma unroll
            for (int pb = 0; pb < 3; ++pb) {
                float av[4];
                const float us = NP == 2 ? inv1 : 1.0f;   // NP = 2: what is SAVED is the true h1 (the accumulator holds S1 h1)
                if (SAVE == 1) {
                    lg_v2f a01, a23, g01, g23;
                    gelu2_both_t<NP == 1>((lg_v2f){acc[pb][0] * us, acc[pb][1] * us}, a01, g01);
                    gelu2_both_t<NP == 1>((lg_v2f){acc[pb][2] * us, acc[pb][3] * us}, a23, g23);
                    av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                    if (inner[pb]) {
                        HS<BF>::st4_nt(a1.a1s, prow[pb], make_float4(av[0], av[1], av[2], av[3]));
                        HS<BF>::st4_nt(a1.g1s, prow[pb], make_float4(g01.x, g01.y, g23.x, g23.y));
                    }
                    if (NP == 2) { av[0] *= sa1; av[1] *= sa1; av[2] *= sa1; av[3] *= sa1; }
                } else {
                    if (SAVE == 2 && inner[pb]) HS<BF>::st4_nt(a1.a1s, prow[pb], make_float4(acc[pb][0] * us, acc[pb][1] * us, acc[pb][2] * us, acc[pb][3] * us));
                    const lg_v2f a01 = NP == 2 ? gelu2_scaled((lg_v2f){acc[pb][0], acc[pb][1]}, g1c, g1h) : gelu2_t<NP == 1>((lg_v2f){acc[pb][0], acc[pb][1]});
                    const lg_v2f a23 = NP == 2 ? gelu2_scaled((lg_v2f){acc[pb][2], acc[pb][3]}, g1c, g1h) : gelu2_t<NP == 1>((lg_v2f){acc[pb][2], acc[pb][3]});
                    av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                }
                u32x2_t q1, q2, q3;
                split_x4<NP>(av, q1, q2, q3);
                uint16_t* dst = A2 + (pb * 16 + r) * LDP + c0;
                *reinterpret_cast<u32x2_t*>(dst) = q1;
                if (NP >= 2) *reinterpret_cast<u32x2_t*>(dst + CH * LDP) = q2;
                if (NP == 3) *reinterpret_cast<u32x2_t*>(dst + 2 * CH * LDP) = q3;
            }
            STAMP(4 + 4 * c);
            __syncthreads();
            STAMP(5 + 4 * c);
            // ---- GEMM2 (K = 64): h2[16 w .. +15][48 pixels] = W2 gelu(h1) ; beside it (VALU): LayerNorm of the next chunk
#pragma unroll
            for (int pb = 0; pb < 3; ++pb) {
                acc[pb] = (f32x4_t){b2v.x, b2v.y, b2v.z, b2v.w};
                const uint16_t* p = A2 + (pb * 16 + r) * LDP + 8 * g;
                mfma_np32<NP>(acc[pb], w2f0, lds_x8(p), lds_x8(p + CH * LDP), lds_x8(p + 2 * CH * LDP));
                mfma_np32<NP>(acc[pb], w2f1, lds_x8(p + 32), lds_x8(p + 32 + CH * LDP), lds_x8(p + 32 + 2 * CH * LDP));
            }
            if (more) ln_store(slot ^ 1, nx, nin);
#pragma unroll
            for (int pb = 0; pb < 3; ++pb) {
                const int m = c * CH + pb * 16 + r;
                const float mk = sMask[slot][pb * 16 + r];
                const float4 hh = make_float4(acc[pb][0] * mk, acc[pb][1] * mk, acc[pb][2] * mk, acc[pb][3] * mk);
                if (SAVE && inner[pb]) HS<BF>::st4_nt(a1.h2, prow[pb], hh);
                int rp = ring0 + m;
                rp = rp >= RING * HX ? rp - RING * HX : rp;
                if (m < npx) *reinterpret_cast<float4*>(ring + rp * LDR + c0) = hh;
            }
            STAMP(6 + 4 * c);
            __syncthreads();   // A2 / XA[slot] are rewritten by the next chunk; the ring rows are complete after the last one
            STAMP(7 + 4 * c);
        }
    };

    {
        float4 pre;
        bool pin;
        ln_fetch(Y0 - 1, 2 * HX, 0, pre, pin);
        compute_rows(Y0 - 1, 2, pre, pin, [] {});
    }
    float4 pre;
    bool pin;
    ln_fetch(Y0 + 1, TY * HX, 0, pre, pin);
#pragma unroll 1
    for (int y0 = Y0; y0 < Yend; y0 += TY) {
#ifdef LG_STAMPS
    stamp_on = (y0 == Y0 + 2 * TY) && strip == 0;
#endif
    STAMP(0);
    // the residual rows of the epilogue are requested early (right behind the first wait of the step, see compute_rows): their HBM
    // round trip hides under the whole step.  wave w owns tile rows 2 w and 2 w + 1; lane (r, g): pixel x0 + r, channels 4 g .. 4 g + 3
    float4 xres[2];
    compute_rows(y0 + 1, TY, pre, pin, [&] {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int y = y0 + 2 * wave + ch, x = x0 + r;
#if LG_XS_UNCOND
            xres[ch] = *reinterpret_cast<const float4*>(a2.x + ((b * h + min(y, h - 1)) * (long)w + min(x, w - 1)) * E + 4 * g);
#else
            xres[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y < Yend && x < w) xres[ch] = *reinterpret_cast<const float4*>(a2.x + ((b * h + y) * (long)w + x) * E + 4 * g);
#endif
        }
    });
    if (y0 + TY < Yend) ln_fetch(y0 + TY + 1, TY * HX, 0, pre, pin);   // next step's first chunk: in flight during P2
    float wq[4][9], bq[4];
    lg_v2f wq01[9], wq23[9];   // the taps as channel PAIRS: the depthwise sum runs on v_pk_fma_f32 (18 instead of 36 instructions per pixel quad)
    {
        const float* tp = a2.dww + 36 * q;
        asm volatile("" : "+v"(tp));            // keep the loads inside the step loop (not hoisted back into 40 live registers)
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
        const float* bp = a2.dwb + 4 * q;
        asm volatile("" : "+v"(bp));
        const float4 bv = *reinterpret_cast<const float4*>(bp);
        bq[0] = bv.x; bq[1] = bv.y; bq[2] = bv.z; bq[3] = bv.w;
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) { wq01[kk] = (lg_v2f){wq[0][kk], wq[1][kk]}; wq23[kk] = (lg_v2f){wq[2][kk], wq[3][kk]}; }
    }
    // ---- P2: per wave, 2 tile rows of 16 pixels: dw3x3 over the ring + GELU -> pieces -> GEMM3 -> bias + residual -> y (+ planar LN half)
    {
        const int sbase = (y0 - Y0) % RING;            // ring slot of row y0 - 1
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int ty = 2 * wave + ch;
#pragma unroll(SAVE == 1 ? LG_XS_SAVE_UNROLL : 4)
            for (int it = 0; it < 4; ++it) {
                const int tx = (lane >> 4) + 4 * it;
                lg_v2f acc01 = (lg_v2f){bq[0], bq[1]}, acc23 = (lg_v2f){bq[2], bq[3]};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    int sl = sbase + ty + dy;
                    sl = sl >= RING ? sl - RING : sl;
                    sl = sl >= RING ? sl - RING : sl;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float4 v = *reinterpret_cast<const float4*>(ring + (sl * HX + tx + dx) * LDR + 4 * q);
                        acc01 = wq01[dy * 3 + dx] * (lg_v2f){v.x, v.y} + acc01;
                        acc23 = wq23[dy * 3 + dx] * (lg_v2f){v.z, v.w} + acc23;
                    }
                }
                const float4 acc = make_float4(acc01.x, acc01.y, acc23.x, acc23.y);
                float av[4];
                if (SAVE == 1) {
                    lg_v2f a01, a23, g01, g23;
                    gelu2_both_t<NP == 1>((lg_v2f){acc.x, acc.y}, a01, g01);
                    gelu2_both_t<NP == 1>((lg_v2f){acc.z, acc.w}, a23, g23);
                    av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                    const int y = y0 + ty, x = x0 + tx;
                    if (y < Yend && x < w) {
                        const long o = ((b * h + y) * (long)w + x) * N1 + 4 * q;
                        HS<BF>::st4_nt(a2.a3s, o, make_float4(av[0], av[1], av[2], av[3]));
                        HS<BF>::st4_nt(a2.g3s, o, make_float4(g01.x, g01.y, g23.x, g23.y));
                    }
                    if (NP == 2) { av[0] *= sa3; av[1] *= sa3; av[2] *= sa3; av[3] *= sa3; }
                } else {
                    if (SAVE >= 2) {
                        const int y = y0 + ty, x = x0 + tx;
                        if (y < Yend && x < w) HS<BF>::st4_nt(a2.a3s, ((b * h + y) * (long)w + x) * N1 + 4 * q, acc);
                    }
                    const lg_v2f a01 = NP == 2 ? gelu2_scaled((lg_v2f){acc.x, acc.y}, 0.70710678118654752440f, g3h) : gelu2_t<NP == 1>((lg_v2f){acc.x, acc.y});
                    const lg_v2f a23 = NP == 2 ? gelu2_scaled((lg_v2f){acc.z, acc.w}, 0.70710678118654752440f, g3h) : gelu2_t<NP == 1>((lg_v2f){acc.z, acc.w});
                    av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                }
                u32x2_t q1, q2, q3;
                split_x4<NP>(av, q1, q2, q3);
                uint16_t* dst = G3 + tx * LDP + 4 * q;
                *reinterpret_cast<u32x2_t*>(dst) = q1;
                if (NP >= 2) *reinterpret_cast<u32x2_t*>(dst + 16 * LDP) = q2;
                if (NP == 3) *reinterpret_cast<u32x2_t*>(dst + 2 * 16 * LDP) = q3;
            }
            STAMP(16 + 3 * ch);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- GEMM3 (K = 64): out[16 channels][16 pixels of tile row ty]
            const float4 b3v = *reinterpret_cast<const float4*>(sB3 + 4 * g);
            f32x4_t o = NP == 2 ? (f32x4_t){b3v.x * S3, b3v.y * S3, b3v.z * S3, b3v.w * S3} : (f32x4_t){b3v.x, b3v.y, b3v.z, b3v.w};
            {
                const uint16_t* p = G3 + r * LDP + 8 * g;
                mfma_np32<NP>(o, w3f0, lds_x8(p), lds_x8(p + 16 * LDP), lds_x8(p + 2 * 16 * LDP));
                mfma_np32<NP>(o, w3f1, lds_x8(p + 32), lds_x8(p + 32 + 16 * LDP), lds_x8(p + 32 + 2 * 16 * LDP));
            }
            __builtin_amdgcn_wave_barrier();           // G3 is rewritten by the next tile row
            STAMP(17 + 3 * ch);
            // ---- epilogue in registers: residual, store, LayerNorm statistics of the next block across the four lane groups
            const int y = y0 + ty, x = x0 + r;
            const float os = NP == 2 ? inv3 : 1.0f;
            const float o0 = o[0] * os + xres[ch].x, o1 = o[1] * os + xres[ch].y, o2 = o[2] * os + xres[ch].z, o3 = o[3] * os + xres[ch].w;
            const bool ok = y < Yend && x < w;
            if (ok) *reinterpret_cast<float4*>(a2.y + ((b * h + y) * (long)w + x) * E + 4 * g) = make_float4(o0, o1, o2, o3);
            if (a2.g) {
                float s = (o0 + o1) + (o2 + o3);
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                const float mu = s * (1.0f / E);
                const float d0 = o0 - mu, d1 = o1 - mu, d2 = o2 - mu, d3 = o3 - mu;
                float v = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                const float rstd = __builtin_amdgcn_rsqf(v * (1.0f / E) + LG_EPS);
                if (ok && g >= 2) {      // channels 8..15 = the global-mixer half, planar [B, e/2, h, w]
                    const long hw = (long)h * w, sp = (long)y * w + x;
                    const float4 ng = *reinterpret_cast<const float4*>(sN1g + 4 * g), nb = *reinterpret_cast<const float4*>(sN1b + 4 * g);
                    float* dst = a2.g + (b * (E / 2) + (4 * g - E / 2)) * hw + sp;
                    dst[0] = d0 * rstd * ng.x + nb.x;
                    dst[hw] = d1 * rstd * ng.y + nb.y;
                    dst[2 * hw] = d2 * rstd * ng.z + nb.z;
                    dst[3 * hw] = d3 * rstd * ng.w + nb.w;
                }
            }
            STAMP(18 + 3 * ch);
        }
    }
    }   // steps of the strip
    }   // strips of this workgroup
}

}   // namespace

int launch_ffn_xs(const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s) {
    constexpr int NPF = 3;
    ProfScope prof__(LG_K_FFN2, s);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn_xs<0, NPF>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_xs<1, NPF>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_xs<2, NPF>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_xs<3, NPF>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_xs<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_xs<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_xs<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_xs<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_xs<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_xs<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_xs<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) { lg_set_error("ffn_xs: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    const int tiles_x = (a2.w + 15) / 16;
    // strip height: the tallest multiple of 8 rows that still yields >= 512 strips (two resident workgroups per CU), at least 16
    int SH = (a2.h + 7) / 8 * 8;
    while (SH > 16 && (long)a2.B * tiles_x * ((a2.h + SH - 1) / SH) < 512) SH = (SH / 2 + 7) / 8 * 8;
    const int strips_y = (a2.h + SH - 1) / SH;
    const int nstrips = a2.B * tiles_x * strips_y;
    const int grid = nstrips < 512 ? nstrips : 512;
    const bool save = a1.h2 != nullptr;           // h2 leaves the chip only for the backward
    const bool pre = save && a1.g1s == nullptr;   // pre-activation saves (h1 in a1s, h3 in a3s)
    const bool noh1 = pre && a1.a1s == nullptr;   // ... without h1 (the backward re-computes it)
    if (save && !pre && !a1.a1s) { lg_set_error("ffn_xs: the five-tensor save needs the gelu(h1) slot"); return -2; }
    if (pre && (!a2.a3s || a2.g3s)) { lg_set_error("ffn_xs: pre-activation saves need the h2 / h3 slots"); return -2; }
    if (pre && a1.hbf && !noh1) { lg_set_error("ffn_xs: bf16 storage keeps h2 / h3 (mode 2) or the five tensors (mode 5)"); return -2; }
    if (a1.hbf) {   // precision = 'bf16': plain bf16 operands, bf16 storage of the saved tensors
        if (noh1) k_ffn_xs<3, 1><<<grid, 256, LDS_BYTES, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
        else if (save) k_ffn_xs<1, 1><<<grid, 256, LDS_BYTES, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
        else k_ffn_xs<0, 1><<<grid, 256, LDS_BYTES, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
    } else if (a1.scales) {   // f16 pairs (three piece products per block instead of six)
        if (noh1) k_ffn_xs<3, 2><<<grid, 256, LDS_BYTES, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
        else if (pre) k_ffn_xs<2, 2><<<grid, 256, LDS_BYTES, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
        else if (save) k_ffn_xs<1, 2><<<grid, 256, LDS_BYTES, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
        else k_ffn_xs<0, 2><<<grid, 256, LDS_BYTES, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
    } else if (noh1) k_ffn_xs<3, NPF><<<grid, 256, LDS_BYTES, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
    else if (pre) k_ffn_xs<2, NPF><<<grid, 256, LDS_BYTES, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
    else if (save) k_ffn_xs<1, NPF><<<grid, 256, LDS_BYTES, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
    else k_ffn_xs<0, NPF><<<grid, 256, LDS_BYTES, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
    LG_CHECK_LAUNCH();
    return 0;
}
