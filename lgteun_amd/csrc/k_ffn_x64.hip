// feed_forward half-block at e = 64 (hidden width 256: level 1 of the 8-band net; reference models/common/LGT.py:91-109, :45-61) on
// the bf16 matrix pipe in fp32-equivalent split arithmetic (split_bf16.h).  At this width neither an fp32 ring of h2 (187 KB for a
// 16-column strip) nor the weights (W2 = 256 x 256: 192 VGPRs per wave as split fragments) fit on chip the way they do at e = 16 / 32,
// so the half-block stays split at the depthwise conv like the f32-MFMA pair it replaces (k_ffn1 + k_ffn2: 438 + 390 us at bs 32):
//   k_ffn1_x64 : LN -> GEMM1 (64 -> 256) -> GELU -> GEMM2 (256 -> 256) -> h2 to HBM.  One 512-thread workgroup per CU walks 64-pixel
//                tiles; wave w owns hidden channels [32 w, 32 w + 32) (two 16-row blocks): W1 fragments register-resident (48 VGPRs),
//                W2 fragments STREAMED per 32-deep K block from the pre-split scratch (k_split_w: coalesced 16-byte reads served by
//                L2, double-buffered in registers: 48 MFMAs per fetch).  LDS: LN(x) pieces [3][64][64] + gelu(h1) pieces [3][64][256]
//                bf16, 16-byte chunks XOR-swizzled by the pixel index = 123 KB.
//   k_ffn2_x64 : dw3x3 + GELU -> GEMM3 (256 -> 64) -> + bias + residual -> y (+ planar LN1 half).  8 x 16 output tile per 512-thread
//                workgroup, wave w = tile row w; the hidden width is walked in four 64-channel chunks: h2 halo tile [10][18][64] fp32
//                through LDS (49 KB), depthwise + GELU + split per wave into its own [3][16][64] pieces, GEMM3 partial sums in
//                registers (W3 fragments streamed per chunk).
#include "kernels.h"

#include "hstore.h"
#include "split_bf16.h"

namespace {

constexpr int E = 64, N1 = 256, TP = 64;          // k1: pixels per tile
constexpr int XA_HALVES = 3 * TP * E;            // 12288 halves (rows of 128 B, 8 chunks)
constexpr int A2_HALVES = 3 * TP * N1;           // 49152 halves (rows of 512 B, 32 chunks)
constexpr size_t LDS1_BYTES = (size_t)(XA_HALVES + A2_HALVES) * 2;
constexpr int NF_W1 = 32, NF_W2 = 128;           // fragments: W1 (16 mb x 2 kb), W2 (16 mb x 8 kb), W3 (4 mb x 8 kb)

__device__ __forceinline__ bf16x8_t lds_x8(const uint16_t* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(p)); }
__device__ __forceinline__ float oct_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return v;
}

// ------------------------------------------------------------------------------------------------
// k_ffn1_x64
// ------------------------------------------------------------------------------------------------
// NP = 3: three bf16 pieces per operand (six products); NP = 2: f16 pairs (three products) with the operand scales of k_ffn_prep.hip -- the
// scheme is k_ffn_x.hip's: operands scaled by proven powers of two, the product of the scales taken out behind the accumulator by constants
// that were multiplications already; what goes to HBM (h2, the saved activations) is always the true value.
template <bool SAVE, int NP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ffn1_x64(Ffn1Args a, const u32x4_t* __restrict__ wsp, long ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint16_t* XA = reinterpret_cast<uint16_t*>(smem_raw);      // [3][TP][E], chunk ^ (px & 7)
    uint16_t* A2 = XA + XA_HALVES;                             // [3][TP][N1], chunk ^ (px & 15) within its group of 16
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    __shared__ __attribute__((aligned(16))) float sLn[2 * E];
    float sx = 1.f, sa1 = 1.f, sw1 = 1.f, sw2 = 1.f;
    if (NP == 2) { sx = a.scales[0]; sa1 = a.scales[1]; sw1 = a.scales[3]; sw2 = a.scales[4]; }
    const float S1 = sx * sw1, S2 = sa1 * sw2, inv1 = 1.0f / S1, inv2 = 1.0f / S2, g1c = 0.70710678118654752440f / S1, g1h = 0.5f * sa1 / S1;   // (powers of two: exact)
    for (int i = threadIdx.x; i < E; i += 512) { sLn[i] = a.ln2g[i] * sx; sLn[E + i] = a.ln2b[i] * sx; }
    // W1 fragments of this wave's two row blocks (mb = 2 w, 2 w + 1), both K blocks
    WFrag32 w1f[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) w1f[m][kb] = ld_wfrag<NP>(wsp, (2 * wave + m) * 2 + kb);
    const u32x4_t* w2p = wsp + (size_t)NF_W1 * 3 * 64;        // W2 fragment (mb, kb): f = mb * 8 + kb
    float4 b1v[2], b2v[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        b1v[m] = *reinterpret_cast<const float4*>(a.b1 + 16 * (2 * wave + m) + 4 * g);
        b2v[m] = *reinterpret_cast<const float4*>(a.b2 + 16 * (2 * wave + m) + 4 * g);
        if (NP == 2) {
            b1v[m] = make_float4(b1v[m].x * S1, b1v[m].y * S1, b1v[m].z * S1, b1v[m].w * S1);
            b2v[m] = make_float4(b2v[m].x * S2, b2v[m].y * S2, b2v[m].z * S2, b2v[m].w * S2);
        }
    }
    // LayerNorm phase: thread t = (tile pixel t / 8, channel octet t % 8)
    const int lpx = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    __syncthreads();

    float4 xr[2];
    auto fetch_x = [&](long tile) {
        const long p = tile * TP + lpx;
        const long pc = p < a.P ? p : a.P - 1;
        xr[0] = *reinterpret_cast<const float4*>(a.x + pc * E + 8 * l8);
        xr[1] = *reinterpret_cast<const float4*>(a.x + pc * E + 8 * l8 + 4);
    };
    if ((long)blockIdx.x < ntiles) fetch_x(blockIdx.x);
#pragma unroll 1
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long p0 = tile * TP;
        // ---- LN(x) of the tile's 64 pixels (8 lanes per pixel), pieces -> XA
        {
            const float xv[8] = {xr[0].x, xr[0].y, xr[0].z, xr[0].w, xr[1].x, xr[1].y, xr[1].z, xr[1].w};
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += xv[k];
            const float mu = oct_sum(s) * (1.0f / E);
            float d[8], v = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { d[k] = xv[k] - mu; v += d[k] * d[k]; }
            const float rstd = __builtin_amdgcn_rsqf(oct_sum(v) * (1.0f / E) + LG_EPS);
            float lng[8], lnb[8];       // re-read per tile (two 32-byte LDS reads) instead of 16 registers pinned through the GEMMs
#pragma unroll
            for (int k = 0; k < 8; ++k) { lng[k] = sLn[8 * l8 + k]; lnb[k] = sLn[E + 8 * l8 + k]; }
            float y0[4], y1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { y0[k] = d[k] * rstd * lng[k] + lnb[k]; y1[k] = d[4 + k] * rstd * lng[4 + k] + lnb[4 + k]; }
            u32x2_t a1, a2, a3, c1, c2, c3;
            split_x4<NP>(y0, a1, a2, a3);
            split_x4<NP>(y1, c1, c2, c3);
            uint16_t* dst = XA + lpx * E + ((l8 ^ (lpx & 7)) << 3);
            *reinterpret_cast<u32x4_t*>(dst) = (u32x4_t){a1.x, a1.y, c1.x, c1.y};
            *reinterpret_cast<u32x4_t*>(dst + TP * E) = (u32x4_t){a2.x, a2.y, c2.x, c2.y};
            if (NP == 3) *reinterpret_cast<u32x4_t*>(dst + 2 * TP * E) = (u32x4_t){a3.x, a3.y, c3.x, c3.y};
        }
        if (tile + gridDim.x < ntiles) fetch_x(tile + gridDim.x);      // next tile's rows: in flight under the GEMMs
        __syncthreads();
        // ---- GEMM1 (K = 64): h1[32 w .. +31][64 pixels]
        f32x4_t acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) acc[m][pb] = (f32x4_t){b1v[m].x, b1v[m].y, b1v[m].z, b1v[m].w};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int px = pb * 16 + r;
                const uint16_t* p = XA + px * E + (((4 * kb + g) ^ (px & 7)) << 3);
                const bf16x8_t x1 = lds_x8(p), x2 = lds_x8(p + TP * E), x3 = lds_x8(p + 2 * TP * E);
#pragma unroll
                for (int m = 0; m < 2; ++m) mfma_np32<NP>(acc[m][pb], w1f[m][kb], x1, x2, x3);
            }
        // ---- GELU, split -> A2
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int px = pb * 16 + r;
                float av[4];
                if (SAVE) {
                    lg_v2f a01, a23, g01, g23;
                    const float us = NP == 2 ? inv1 : 1.0f;
                    gelu2_both_f((lg_v2f){acc[m][pb][0] * us, acc[m][pb][1] * us}, a01, g01);
                    gelu2_both_f((lg_v2f){acc[m][pb][2] * us, acc[m][pb][3] * us}, a23, g23);
                    av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                    if (p0 + px < a.P) {
                        const long o = (p0 + px) * N1 + 16 * (2 * wave + m) + 4 * g;
                        HS<false>::st4_nt(a.a1s, o, make_float4(av[0], av[1], av[2], av[3]));
                        HS<false>::st4_nt(a.g1s, o, make_float4(g01.x, g01.y, g23.x, g23.y));
                    }
                    if (NP == 2) { av[0] *= sa1; av[1] *= sa1; av[2] *= sa1; av[3] *= sa1; }
                } else {
                    const lg_v2f a01 = NP == 2 ? gelu2_scaled((lg_v2f){acc[m][pb][0], acc[m][pb][1]}, g1c, g1h) : gelu2_f((lg_v2f){acc[m][pb][0], acc[m][pb][1]});
                    const lg_v2f a23 = NP == 2 ? gelu2_scaled((lg_v2f){acc[m][pb][2], acc[m][pb][3]}, g1c, g1h) : gelu2_f((lg_v2f){acc[m][pb][2], acc[m][pb][3]});
                    av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                }
                u32x2_t q1, q2, q3;
                split_x4<NP>(av, q1, q2, q3);
                const int chunk = 2 * (2 * wave + m) + (g >> 1);           // logical 16-byte chunk of channels 16 mb + 4 g ..
                uint16_t* dst = A2 + px * N1 + ((chunk ^ (px & 15)) << 3) + 4 * (g & 1);
                *reinterpret_cast<u32x2_t*>(dst) = q1;
                *reinterpret_cast<u32x2_t*>(dst + TP * N1) = q2;
                if (NP == 3) *reinterpret_cast<u32x2_t*>(dst + 2 * TP * N1) = q3;
            }
        __syncthreads();
        // ---- GEMM2 (K = 256): h2[32 w .. +31][64 pixels]; W2 fragments streamed per K block, next block in flight under the MFMAs
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) acc[m][pb] = (f32x4_t){b2v[m].x, b2v[m].y, b2v[m].z, b2v[m].w};
        WFrag32 wc[2], wn[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) wc[m] = ld_wfrag<NP>(w2p, (2 * wave + m) * 8);
#pragma unroll 1
        for (int kb = 0; kb < 8; ++kb) {
            {
                const int kn = kb < 7 ? kb + 1 : 7;      // (the last iteration re-reads its own block: cached, unused)
#pragma unroll
                for (int m = 0; m < 2; ++m) wn[m] = ld_wfrag<NP>(w2p, (2 * wave + m) * 8 + kn);
            }
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const int px = pb * 16 + r;
                const uint16_t* p = A2 + px * N1 + (((4 * kb + g) ^ (px & 15)) << 3);
                const bf16x8_t x1 = lds_x8(p), x2 = lds_x8(p + TP * N1), x3 = lds_x8(p + 2 * TP * N1);
#pragma unroll
                for (int m = 0; m < 2; ++m) mfma_np32<NP>(acc[m][pb], wc[m], x1, x2, x3);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) wc[m] = wn[m];
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const long p = p0 + pb * 16 + r;
                const float us = NP == 2 ? inv2 : 1.0f;      // h2 itself goes to HBM
                if (p < a.P) HS<false>::st4(a.h2, p * N1 + 16 * (2 * wave + m) + 4 * g, make_float4(acc[m][pb][0] * us, acc[m][pb][1] * us, acc[m][pb][2] * us, acc[m][pb][3] * us));
            }
        __syncthreads();      // A2 / XA are rewritten by the next tile
    }
}

// ------------------------------------------------------------------------------------------------
// k_ffn2_x64
// ------------------------------------------------------------------------------------------------
constexpr int TX = 16, TY = 8, HX = 18, HY = 10, LDH = 68, CC = 64;      // CC = hidden channels per chunk
constexpr int G3_WAVE = 3 * 16 * CC;                                    // 3072 halves per wave
constexpr size_t LDS2_BYTES = (size_t)HY * HX * LDH * 4 + (size_t)8 * G3_WAVE * 2;
constexpr int NLD = (HY * HX * (CC / 4) + 511) / 512;                  // float4 items per thread of one halo chunk (6)

template <bool SAVE, int NP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ffn2_x64(Ffn2Args a, const u32x4_t* __restrict__ w3p, const float* __restrict__ scales, int tiles_x, int tiles_y, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* H = reinterpret_cast<float*>(smem_raw);                                         // [HY*HX][LDH] h2 halo tile, one chunk
    uint16_t* G3all = reinterpret_cast<uint16_t*>(smem_raw + (size_t)HY * HX * LDH * 4);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    uint16_t* G3 = G3all + wave * G3_WAVE;                                                 // [3][16][CC], chunk ^ (px & 7)
    const int h = a.h, w = a.w;
    __shared__ __attribute__((aligned(16))) float sPar[3 * E];
    for (int i = threadIdx.x; i < E; i += 512) { sPar[i] = a.b3[i]; sPar[E + i] = a.g ? a.n1g[i] : 0.f; sPar[2 * E + i] = a.g ? a.n1b[i] : 0.f; }
    const int q16 = lane & 15;
    float sa3 = 1.f, sw3 = 1.f;
    if (NP == 2) { sa3 = scales[2]; sw3 = scales[5]; }
    const float S3 = sa3 * sw3, inv3 = 1.0f / S3, g3h = 0.5f * sa3;
    __syncthreads();
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx_i = t % tiles_x;
        t /= tiles_x;
        const int ty_i = t % tiles_y;
        const long b = t / tiles_y;
        const int x0 = tx_i * TX, y0 = ty_i * TY;
        const int ty = wave;                             // wave w = tile row w
        // residual rows of the epilogue: lane (r, g) = pixel x0 + r, channels 16 mb + 4 g ..
        float4 xres[4];
        {
            const int y = y0 + ty, x = x0 + r;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                xres[mb] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (y < h && x < w) xres[mb] = *reinterpret_cast<const float4*>(a.x + ((b * h + y) * (long)w + x) * E + 16 * mb + 4 * g);
            }
        }
        f32x4_t o[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const float4 b3v = *reinterpret_cast<const float4*>(sPar + 16 * mb + 4 * g);
            o[mb] = (f32x4_t){b3v.x * S3, b3v.y * S3, b3v.z * S3, b3v.w * S3};
        }
        // halo chunk loader: item i = (halo pixel i / 16, channel quad i % 16) of chunk kc
        float4 hr[NLD];
        auto fetch_h = [&](int kc) {
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                const int i = threadIdx.x + it * 512;
                const int m = i >> 4, qq = i & 15;
                const int hy = m / HX, hx = m - hy * HX;
                const int y = y0 + hy - 1, x = x0 + hx - 1;
                hr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < HY * HX && y >= 0 && y < h && x >= 0 && x < w)
                    hr[it] = HS<false>::ld4(a.h2, ((b * h + y) * (long)w + x) * N1 + CC * kc + 4 * qq);
            }
        };
        fetch_h(0);
#pragma unroll 1
        for (int kc = 0; kc < 4; ++kc) {
            __syncthreads();                             // the previous chunk's readers of H are done
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                const int i = threadIdx.x + it * 512;
                const int m = i >> 4, qq = i & 15;
                if (m < HY * HX) *reinterpret_cast<float4*>(H + m * LDH + 4 * qq) = hr[it];
            }
            if (kc < 3) fetch_h(kc + 1);                 // next chunk: in flight under this chunk's arithmetic
            const int qc = 16 * kc + q16;                // channel quad of this lane in the full hidden width
            float wq[4][9], bq[4];
            {
                const float* tp = a.dww + 36 * qc;
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
                const float4 bv = *reinterpret_cast<const float4*>(a.dwb + 4 * qc);
                bq[0] = bv.x; bq[1] = bv.y; bq[2] = bv.z; bq[3] = bv.w;
            }
            __syncthreads();
            // ---- depthwise 3x3 + GELU + split of tile row ty, channels of this chunk -> G3
#pragma unroll(SAVE ? 1 : 2)
            for (int it = 0; it < 4; ++it) {
                const int tx = (lane >> 4) + 4 * it;
                float4 acc = make_float4(bq[0], bq[1], bq[2], bq[3]);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float4 v = *reinterpret_cast<const float4*>(H + ((ty + dy) * HX + tx + dx) * LDH + 4 * q16);
                        acc.x += wq[0][dy * 3 + dx] * v.x; acc.y += wq[1][dy * 3 + dx] * v.y;
                        acc.z += wq[2][dy * 3 + dx] * v.z; acc.w += wq[3][dy * 3 + dx] * v.w;
                    }
                float av[4];
                if (SAVE) {
                    lg_v2f a01, a23, g01, g23;
                    gelu2_both_f((lg_v2f){acc.x, acc.y}, a01, g01);
                    gelu2_both_f((lg_v2f){acc.z, acc.w}, a23, g23);
                    av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                    const int y = y0 + ty, x = x0 + tx;
                    if (y < h && x < w) {
                        const long off = ((b * h + y) * (long)w + x) * N1 + 4 * qc;
                        HS<false>::st4_nt(a.a3s, off, make_float4(av[0], av[1], av[2], av[3]));
                        HS<false>::st4_nt(a.g3s, off, make_float4(g01.x, g01.y, g23.x, g23.y));
                    }
                    if (NP == 2) { av[0] *= sa3; av[1] *= sa3; av[2] *= sa3; av[3] *= sa3; }
                } else {
                    const lg_v2f a01 = NP == 2 ? gelu2_scaled((lg_v2f){acc.x, acc.y}, 0.70710678118654752440f, g3h) : gelu2_f((lg_v2f){acc.x, acc.y});
                    const lg_v2f a23 = NP == 2 ? gelu2_scaled((lg_v2f){acc.z, acc.w}, 0.70710678118654752440f, g3h) : gelu2_f((lg_v2f){acc.z, acc.w});
                    av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                }
                u32x2_t q1, q2, q3;
                split_x4<NP>(av, q1, q2, q3);
                uint16_t* dst = G3 + tx * CC + (((q16 >> 1) ^ (tx & 7)) << 3) + 4 * (q16 & 1);
                *reinterpret_cast<u32x2_t*>(dst) = q1;
                *reinterpret_cast<u32x2_t*>(dst + 16 * CC) = q2;
                if (NP == 3) *reinterpret_cast<u32x2_t*>(dst + 2 * 16 * CC) = q3;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- GEMM3 partial: out[64 channels][16 pixels of tile row ty] += W3[:, 64 kc ..] gelu(h3)[64 kc ..]
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const uint16_t* p = G3 + r * CC + (((4 * kb + g) ^ (r & 7)) << 3);
                const bf16x8_t x1 = lds_x8(p), x2 = lds_x8(p + 16 * CC), x3 = lds_x8(p + 2 * 16 * CC);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const WFrag32 wf = ld_wfrag<NP>(w3p, mb * 8 + 2 * kc + kb);
                    mfma_np32<NP>(o[mb], wf, x1, x2, x3);
                }
            }
            __builtin_amdgcn_wave_barrier();             // G3 is rewritten by the next chunk
        }
        // ---- epilogue in registers: residual, store, LayerNorm statistics of the next block across the four lane groups
        const int y = y0 + ty, x = x0 + r;
        const bool ok = y < h && x < w;
        float ov[16];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            ov[4 * mb + 0] = o[mb][0] * inv3 + xres[mb].x; ov[4 * mb + 1] = o[mb][1] * inv3 + xres[mb].y;
            ov[4 * mb + 2] = o[mb][2] * inv3 + xres[mb].z; ov[4 * mb + 3] = o[mb][3] * inv3 + xres[mb].w;
            if (ok) *reinterpret_cast<float4*>(a.y + ((b * h + y) * (long)w + x) * E + 16 * mb + 4 * g) =
                        make_float4(ov[4 * mb], ov[4 * mb + 1], ov[4 * mb + 2], ov[4 * mb + 3]);
        }
        if (a.g) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s += ov[i];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const float mu = s * (1.0f / E);
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { ov[i] -= mu; v += ov[i] * ov[i]; }
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const float rstd = __builtin_amdgcn_rsqf(v * (1.0f / E) + LG_EPS);
            if (ok) {      // channels 32..63 (output blocks mb = 2, 3) = the global-mixer half, planar [B, e/2, h, w]
                const long hw = (long)h * w, sp = (long)y * w + x;
#pragma unroll
                for (int mb = 2; mb < 4; ++mb) {
                    const float4 ng = *reinterpret_cast<const float4*>(sPar + E + 16 * mb + 4 * g), nb = *reinterpret_cast<const float4*>(sPar + 2 * E + 16 * mb + 4 * g);
                    float* dst = a.g + (b * (E / 2) + 16 * (mb - 2) + 4 * g) * hw + sp;
                    dst[0] = ov[4 * mb] * rstd * ng.x + nb.x;
                    dst[hw] = ov[4 * mb + 1] * rstd * ng.y + nb.y;
                    dst[2 * hw] = ov[4 * mb + 2] * rstd * ng.z + nb.z;
                    dst[3 * hw] = ov[4 * mb + 3] * rstd * ng.w + nb.w;
                }
            }
        }
    }
}

}   // namespace

template <int NP>
static int launch_ffn_x64_t(const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s) {
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn1_x64<false, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS1_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn1_x64<true, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS1_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn2_x64<false, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS2_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn2_x64<true, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS2_BYTES);
        if (e != hipSuccess) { lg_set_error("ffn_x64: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    if (!a1.wsplit_ready) {
        const int rc = launch_split_w(a1.w1, a1.w2, a2.w3, a1.wsplit, E, NP, s, a1.scales);
        if (rc) return rc;
    }
    const u32x4_t* wsp = reinterpret_cast<const u32x4_t*>(a1.wsplit);
    const bool save = a1.a1s != nullptr;
    // one profiling scope over BOTH launches: the half-block's flops / bytes (bench.py algorithmic_per_launch) are charged to the pair
    ProfScope prof__(LG_K_FFN2, s);
    {
        const long ntiles = (a1.P + TP - 1) / TP;
        const int grid = (int)(ntiles < 256 ? ntiles : 256);
        if (save) k_ffn1_x64<true, NP><<<grid, 512, LDS1_BYTES, s>>>(a1, wsp, ntiles);
        else k_ffn1_x64<false, NP><<<grid, 512, LDS1_BYTES, s>>>(a1, wsp, ntiles);
        LG_CHECK_LAUNCH();
    }
    {
        const int tiles_x = (a2.w + TX - 1) / TX, tiles_y = (a2.h + TY - 1) / TY;
        const int ntiles = a2.B * tiles_x * tiles_y;
        const int grid = ntiles < 256 ? ntiles : 256;
        const u32x4_t* w3p = wsp + (size_t)(NF_W1 + NF_W2) * 3 * 64;
        if (save) k_ffn2_x64<true, NP><<<grid, 512, LDS2_BYTES, s>>>(a2, w3p, a1.scales, tiles_x, tiles_y, ntiles);
        else k_ffn2_x64<false, NP><<<grid, 512, LDS2_BYTES, s>>>(a2, w3p, a1.scales, tiles_x, tiles_y, ntiles);
        LG_CHECK_LAUNCH();
    }
    return 0;
}

int launch_ffn_x64(const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s) {
    if (!a1.wsplit || !a1.h2) { lg_set_error("ffn_x64: missing workspace (weight fragments / h2)"); return -3; }
    return a1.scales ? launch_ffn_x64_t<2>(a1, a2, s) : launch_ffn_x64_t<3>(a1, a2, s);
}
