// k_ffn_dw_bwd_xs<E>: the spatial half of the feed_forward backward at e = 16 and (round 4) e = 32 (reference models/common/LGT.py:91-109:
// net.4, the second GELU and the depthwise 3x3 of net.2), as a STRIP WALK like the forward's k_ffn_xs / k_ffn_x32:
//
//     dh3 = (W3^T dy) * gelu'(h3)          on the halo pixels, kept in a 10-row fp32 LDS ring (never stored)
//     dh2 = dw3x3^T dh3                    -> HBM, the one tensor k_ffn1_bwd_xs reads
//     d dww[c][k] += h2(q) dh3(q - off_k),  d dwb[c] += dh3,  dW3 += dy (x) gelu(h3),  db3 += dy        (pixel sums)
//
// A workgroup walks DOWN a 16-column strip in 8-row steps; a step computes dh3 only on its 8 NEW halo rows (8 x 18 = 144 pixels,
// three chunks of 48), so dy and h3 are read 1.125 x instead of the 1.40 x of the tile kernel it replaces (k_ffn_dw_bwd<16>), h2 is
// read exactly once (the tap products use the centre pixel's h2 against the NEIGHBOURS' dh3: d dww[c][k] = sum_q h2(q) dh3(q - off_k),
// the same neighbour value that dh2 needs), and the weight gradient of net.4 comes out of the same pass on the bf16 matrix pipe in
// split arithmetic (split_bf16.h) -- k_wgrad_t<1,4> and its second read of dy / h3 are gone.
// GEMM (W3^T dy): weights on the A side, pixels on the B side, wave w = hidden channels [16 w, 16 w + 16): a lane holds four
// consecutive channels of one pixel, so h3 arrives and dh3 leaves as 16-byte accesses.  dW3: the pixel axis is the K dimension; dy^T
// is read by columns from the chunk's dy image (ds_read_b64_tr_b16), gelu(h3) goes through a 4.5 KB per-wave [pixel][channel] image of the chunk.
// LDS 79 KB (two workgroups per CU): ring [10][18][68] fp32 | dy pieces [2][3][48][16] bf16 | per-wave gelu(h3) pieces [4][3][48][16] | depthwise taps [64][9] fp32.
// e = 32 (hidden width 128: level 0 of the 8-band net, level 1 of the 4-band net; round 4): everything behind W3^T dy is per hidden CHANNEL, so
// the 128 channels are two independent halves of 64: blockIdx.y = half, each half IS the e = 16 walk (four waves, two workgroups per CU, 74 KB:
// the gelu(h3) image is per 16-pixel block there) on channels [64 half, 64 half + 64) of h3 / h2 / dh2, with K = 32 in W3^T dy (one 32-deep block
// per piece product) and two 16-row tiles of dW3 per wave; dy is read by both halves (the second read is an L2 hit).  A first form with EIGHT
// waves in one 151 KB workgroup per CU ran 384 us against 314 + 52 for the kernels it replaces: every wave of the CU in the same phase, nothing
// beside the barriers, 66 spilled registers whose scratch traffic waits for the prefetches (one in-order counter) -- stamps: 56 k ticks per step
// against 2 x 12.5 k.  It replaces k_ffn_dw_bwd<32> (channel-split tiles: 1.40 x read amplification, h2 read twice) and the 32 x 128 k_wgrad_t
// launch with its second pass over dy and gelu(h3).
#include "kernels.h"
#include "bwd_kernels.h"
#include "split_bf16.h"
#include "hstore.h"

// In-kernel phase stamps (diagnostic build only, -DLG_STAMPS: tools/build_stamps.sh + tools/bwd_stamps.py)
#ifdef LG_STAMPS
__device__ unsigned long long g_ka_stamps[8 * 16];
#define STAMP(i) do { if (blockIdx.x == 0 && lane == 0 && stamp_on) g_ka_stamps[wave * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" __attribute__((visibility("default"))) int lg_debug_ka_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ka_stamps), sizeof(g_ka_stamps));
}
#else
#define STAMP(i) do { } while (0)
#endif

namespace {

constexpr int TX = 16, HX = 18, TY = 8, RING = 10, CH = 48;
constexpr int NH = 64, NW = 4, NT = 256, LDR = NH + 4, CQ = NH / 4;   // hidden channels / waves / threads of a workgroup, ring row, channel quads
constexpr int A3_PIECE = 16 * 16;            // halves, per wave and piece: one 16-pixel block x the wave's 16 channels
template <int E_>
struct KA {
    static constexpr int E = E_, N1 = 4 * E, NHALF = N1 / NH, LQ = E / 4, NV = (CH * LQ + NT - 1) / NT;   // NV: dy vectors per thread and chunk
    static constexpr int DY_PIECE = CH * E;      // halves
    static constexpr int DY_SLOT = 3 * DY_PIECE;
    static constexpr size_t OFF_DY = (size_t)RING * HX * LDR * 4;
    static constexpr size_t OFF_A3 = OFF_DY + (size_t)2 * DY_SLOT * 2;
    static constexpr size_t OFF_TAPS = OFF_A3 + (size_t)NW * 3 * A3_PIECE * 2;
    static constexpr size_t LDS_BYTES = OFF_TAPS + (size_t)NH * 9 * 4;
    // slab row of a workgroup: [d dww 64 x 9 | d dwb 64 | dW3 E x 64 | db3 E]  (its channel half)
    static constexpr int R_DB = NH * 9, R_W3 = NH * 10, R_B3 = R_W3 + E * NH, ROW = R_B3 + E;
    static_assert(OFF_DY % 16 == 0 && OFF_A3 % 16 == 0 && OFF_TAPS % 16 == 0, "16-byte aligned LDS regions");
    static_assert((size_t)(NW * CQ * 40 + NW * E) * 4 <= OFF_DY, "the end-of-kernel reduction rows alias the ring");
    static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
};
static_assert(KA<16>::R_DB == FFN_DW_BWD_X_DB && KA<16>::R_W3 == FFN_DW_BWD_X_W3 && KA<16>::R_B3 == FFN_DW_BWD_X_B3 && KA<16>::ROW == FFN_DW_BWD_X_ROW, "slab row (bwd_kernels.h)");
static_assert(KA<32>::ROW == FFN_DW_BWD_X32_ROW, "slab row (bwd_kernels.h)");

typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
__device__ __forceinline__ s16x4_t lds_x4(const uint16_t* p) { return __builtin_bit_cast(s16x4_t, *reinterpret_cast<const u32x2_t*>(p)); }
__device__ __forceinline__ s16x4_t lds_tr4(const uint16_t* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p); }
__device__ __forceinline__ void mfma6_16(f32x4_t& acc, const s16x4_t (&a)[3], const s16x4_t (&b)[3]) {
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b[0], acc, 0, 0, 0);
}

__device__ __forceinline__ bf16x8_t cat8(s16x4_t lo, s16x4_t hi) {
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
#ifndef LG_KA_ITEMFENCE
#define LG_KA_ITEMFENCE 1   // P2: a scheduling fence behind every LG_KA_ITEMFENCE-th item (0: none)
#endif
#ifndef LG_KA_PAIR
#define LG_KA_PAIR 0   // measured in THIS kernel (VALU / LDS bound, matrix pipe 10 % busy): the operand concatenation costs 133.7 vs 119.5 us per launch; off
#endif
// the six piece products as three 32-deep MFMAs, two products per instruction (k_ffn_bwd_x.hip: mfma3_16)
__device__ __forceinline__ void mfma3_16(f32x4_t& acc, const s16x4_t (&a)[3], const s16x4_t (&b)[3]) {
    const bf16x8_t b31 = cat8(b[2], b[0]), b21 = cat8(b[1], b[0]);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(a[0], a[2]), b31, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(a[1], a[1]), b21, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(a[0], a[0]), b21, acc, 0, 0, 0);
}
template <int NP>
__device__ __forceinline__ void mfmaN_16(f32x4_t& acc, const s16x4_t (&a)[3], const s16x4_t (&b)[3]) {
    if (NP == 3) { if (LG_KA_PAIR) mfma3_16(acc, a, b); else mfma6_16(acc, a, b); }
    else acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b[0], acc, 0, 0, 0);
}

__device__ __forceinline__ bf16x8_t lds_x8(const uint16_t* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(p)); }

// NP = 3: fp32 storage of h2 / h3 / dh2, fp32-equivalent split products; NP = 1 (precision = 'bf16'): bf16 storage (hstore.h), plain bf16 products
template <int E, int NP>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void k_ffn_dw_bwd_xs(FfnDwBwdXArgs a, int tiles_x, int strips_y, int nstrips, int SH, int dS) {
    using C = KA<E>;
    constexpr int N1 = C::N1, LQ = C::LQ, NV = C::NV, DY_PIECE = C::DY_PIECE, DY_SLOT = C::DY_SLOT;
    constexpr int PPL = 4, RPW = 2, NM = E / 16;       // P2: consecutive pixels of a tile row per lane, tile rows per wave; 16-row tiles of dW3
    constexpr bool BF = (NP == 1);
    const int hoff = blockIdx.y * NH;                  // this workgroup's channel half [hoff, hoff + 64) of the N1 hidden channels
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* ring = reinterpret_cast<float*>(smem_raw);                          // [RING*HX][LDR] dh3
    uint16_t* DY = reinterpret_cast<uint16_t*>(smem_raw + C::OFF_DY);          // [2][3][CH][E]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    uint16_t* A3 = reinterpret_cast<uint16_t*>(smem_raw + C::OFF_A3) + wave * 3 * A3_PIECE;   // [3][48 px][16 ch] of this wave
    const int h = a.h, w = a.w;
    const int c0 = wave * 16 + 4 * g;                 // first of the lane's four hidden channels after the GEMM (local to the half)
    // W3^T rows [16 w, 16 w + 16): one 16-deep block at e = 16, one 32-deep block at e = 32
    WFrag16 w3f16;
    WFrag32 w3f32;
    if constexpr (E == 16) w3f16 = NP == 3 ? load_wfrag16(a.w3t + (size_t)(hoff + wave * 16) * E, E, 0) : load_wfrag16_rne(a.w3t + (size_t)(hoff + wave * 16) * E, E, 0);
    else w3f32 = NP == 3 ? load_wfrag32(a.w3t + (size_t)(hoff + wave * 16) * E, E, 0) : load_wfrag32_rne(a.w3t + (size_t)(hoff + wave * 16) * E, E, 0);
    const int q = lane % CQ, lgrp = lane / CQ;        // P2: lane = (lane group, channel quad q)
    // dy role: vector v of thread t = (chunk pixel, channel quad) number t + 256 v of the chunk's CH * LQ (192 at e = 16: one per thread; 384 at
    // e = 32: threads 0 .. 127 carry a second one)
    const int lq = threadIdx.x % LQ;                                           // 256 % LQ == 0: the same quad for every vector of a thread
    auto dy_px = [&](int v) { return (threadIdx.x + NT * v) / LQ; };
    auto dy_has = [&](int v) { return threadIdx.x + NT * v < LQ * CH; };

    // the depthwise taps [N1][9], once per workgroup in LDS: re-read per step from there (from L1 / L2 the wait for them sat behind every
    // HBM load in flight -- one in-order counter -- 1.9 k ticks of a 24 k-tick step, profiles/r03_ffn_bwd_phase_stamps.txt)
    float* sTaps = reinterpret_cast<float*>(smem_raw + C::OFF_TAPS);
    for (int i = threadIdx.x; i < NH * 9; i += NT) sTaps[i] = a.dww[hoff * 9 + i];
    // gradient partials of the depthwise taps / bias of the lane's four P2 channels, as channel PAIRS (v_pk_fma_f32)
    lg_v2f pw01[10], pw23[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) { pw01[k] = (lg_v2f){0.f, 0.f}; pw23[k] = (lg_v2f){0.f, 0.f}; }
    f32x4_t acc3[NM];                                 // dW3[16 mt + 4 g + v][16 w + r]
#pragma unroll
    for (int mt = 0; mt < NM; ++mt) acc3[mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float4 sb3 = make_float4(0.f, 0.f, 0.f, 0.f);     // db3[4 lq ..], this thread's own pixels

    float dmx = 0.f;     // max |dh2| of this thread's stores (a.dh2_max: the operand scale of k_ffn1_bwd_xs's f16-pair products)
#pragma unroll 1
    for (int strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
    // dS != 0 (launcher: one strip per workgroup, two workgroups per CU): vertically adjacent strips are SH + dS rows for a workgroup of the first half of the grid and
    // SH - dS for its neighbour's in the second half -- the workgroup a CU received first runs faster (k_ffn_xr.hip strip_geo has the measurement)
    int tx_i, Y0, Yend;
    long b;
    if (dS) {
        const int half = nstrips >> 1, shortone = strip >= half ? 1 : 0;
        int p_ = strip - (shortone ? half : 0);
        tx_i = p_ % tiles_x;
        p_ /= tiles_x;
        const int hy2 = strips_y >> 1, ky = p_ % hy2;
        b = p_ / hy2;
        Y0 = ky * 2 * SH + (shortone ? SH + dS : 0);
        Yend = min(Y0 + (shortone ? SH - dS : SH + dS), h);
    } else {
        int t = strip;
        tx_i = t % tiles_x;
        t /= tiles_x;
        const int sy = t % strips_y;
        b = t / strips_y;
        Y0 = sy * SH;
        Yend = min(Y0 + SH, h);
    }
    const int x0 = tx_i * TX;
#ifdef LG_STAMPS
    bool stamp_on = false;
#endif

    // Every global load below is UNCONDITIONAL, from a clamped (always valid) address, and masked afterwards by a select: a load inside an
    // exec-masked branch makes the compiler wait with vmcnt(0) at the join (loads and stores share ONE in-order counter on gfx950), which
    // turns every prefetch into a full HBM round trip on the spot (in-kernel stamps, profiles/r03_ffn_bwd_phase_stamps.txt).
    // dy vector of halo pixel m of the row block starting at ya; in: inside the image, own: the pixel belongs to THIS strip
    struct DyVec { float4 dv[NV]; bool in[NV], own[NV]; };
    auto dy_fetch = [&](int ya, int npx, int c, DyVec& d) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int m = c * CH + dy_px(v);
            const int hy = m / HX, hx = m - hy * HX;
            const int y = ya + hy, x = x0 + hx - 1;
            d.in[v] = dy_has(v) && (m < npx) && y >= 0 && y < h && x >= 0 && x < w;
            d.own[v] = d.in[v] && hx >= 1 && hx <= TX && y >= Y0 && y < Yend;
            d.dv[v] = *reinterpret_cast<const float4*>(a.dy + ((b * h + clampi(y, 0, h - 1)) * (long)w + clampi(x, 0, w - 1)) * E + 4 * lq);
        }
    };
    auto dy_store = [&](int slot, const DyVec& d) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 dvr = d.dv[v];
            const bool in = d.in[v];
            const float4 dv = make_float4(in ? dvr.x : 0.f, in ? dvr.y : 0.f, in ? dvr.z : 0.f, in ? dvr.w : 0.f);   // dy = 0 outside the image: so is dh3
            if (d.own[v]) { sb3.x += dv.x; sb3.y += dv.y; sb3.z += dv.z; sb3.w += dv.w; }
            if (!dy_has(v)) continue;        // the last wave(s) hold no (second) dy vector
            const float vv[4] = {dv.x, dv.y, dv.z, dv.w};
            u32x2_t q1, q2, q3;
            split_x4<NP>(vv, q1, q2, q3);
            uint16_t* dst = DY + slot * DY_SLOT + dy_px(v) * E + 4 * lq;
            *reinterpret_cast<u32x2_t*>(dst) = q1;
            if (NP == 3) {
                *reinterpret_cast<u32x2_t*>(dst + DY_PIECE) = q2;
                *reinterpret_cast<u32x2_t*>(dst + 2 * DY_PIECE) = q3;
            }
        }
    };
    // h3 of the lane's pixel / channels in chunk c.  No mask: outside the image dy = 0 makes dh3 = 0 whatever gelu'(h3) is, and gelu(h3)
    // enters dW3 only for the strip's own pixels
    auto h3_fetch = [&](int ya, int c, typename HS<BF>::raw4 (&hv)[3]) {
#pragma unroll
        for (int pb = 0; pb < 3; ++pb) {
            const int m = c * CH + pb * 16 + r;
            const int hy = m / HX, hx = m - hy * HX;
            const int y = clampi(ya + hy, 0, h - 1), x = clampi(x0 + hx - 1, 0, w - 1);
            hv[pb] = HS<BF>::ldraw(a.h3, ((b * h + y) * (long)w + x) * N1 + hoff + c0);
        }
    };

    // dh3 of halo rows [ya, ya + nr) x columns [x0 - 1, x0 + 17) -> ring (nr = 2: strip prologue, 8: one step)
    // in_last: run at the top of the LAST chunk, where no next-chunk operands are in flight (registers and load slots are free)
    // pre / pre_in / pre_own, h3c: chunk 0's dy vector and h3 vectors, requested by the caller ahead of time
    auto compute_rows = [&](int ya, int nr, const DyVec& pre, typename HS<BF>::raw4 (&h3c)[3], auto&& in_last) {
        const int npx = nr * HX, nchunks = (npx + CH - 1) / CH;
        typename HS<BF>::raw4 h3n[3];
        dy_store(0, pre);
        STAMP(1);
        __syncthreads();                 // also: the previous phase (P2) is done reading the ring rows this call overwrites
        STAMP(2);
        const int ring0 = ((ya - Y0 + 1) % RING) * HX;
        for (int c = 0; c < nchunks; ++c) {
            const int slot = c & 1;
            const bool more = c + 1 < nchunks;
            DyVec nd;
            if (more) { dy_fetch(ya, npx, c + 1, nd); h3_fetch(ya, c + 1, h3n); }
            else in_last();
            const uint16_t* dyb = DY + slot * DY_SLOT;
#pragma unroll
            for (int pb = 0; pb < 3; ++pb) {
                const int m = c * CH + pb * 16 + r;
                const int hy = m / HX, hx = m - hy * HX;
                const int y = ya + hy, x = x0 + hx - 1;
                const bool own = m < npx && hx >= 1 && hx <= TX && x < w && y >= Y0 && y < Yend;
                // (W3^T dy)[16 w + 4 g + v][pixel r]
                f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                if constexpr (E == 16) {
                    const uint16_t* p = dyb + (pb * 16 + r) * E + 4 * g;
                    s16x4_t xb[3];
                    xb[0] = lds_x4(p);
                    if (NP == 3) { xb[1] = lds_x4(p + DY_PIECE); xb[2] = lds_x4(p + 2 * DY_PIECE); } else { xb[1] = xb[0]; xb[2] = xb[0]; }
                    mfmaN_16<NP>(acc, w3f16.p, xb);
                } else {
                    const uint16_t* p = dyb + (pb * 16 + r) * E + 8 * g;
                    const bf16x8_t x1 = lds_x8(p);
                    bf16x8_t x2 = x1, x3 = x1;
                    if (NP == 3) { x2 = lds_x8(p + DY_PIECE); x3 = lds_x8(p + 2 * DY_PIECE); }
                    mfma_np32<NP>(acc, w3f32, x1, x2, x3);
                }
                lg_v2f a01, a23, g01, g23;
                const float4 h3v = HS<BF>::widen(h3c[pb]);
                gelu2_both_t<NP == 1>((lg_v2f){h3v.x, h3v.y}, a01, g01);
                gelu2_both_t<NP == 1>((lg_v2f){h3v.z, h3v.w}, a23, g23);
                // dh3 (0 outside the image: dy is 0 there) -> ring
                int rp = ring0 + m;
                rp = rp >= RING * HX ? rp - RING * HX : rp;
                if (m < npx) *reinterpret_cast<float4*>(ring + rp * LDR + c0) = make_float4(acc[0] * g01.x, acc[1] * g01.y, acc[2] * g23.x, acc[3] * g23.y);
                // gelu(h3) of the strip's OWN pixels (every pixel of the image belongs to exactly one strip) -> the wave's [pixel][channel] image
                // of THIS 16-pixel block, consumed right away by dW3[.][16 w + .] += dy^T gelu(h3): both operands read by columns (K = the
                // block's 16 pixels); off the critical path of the ring
                const float mk = own ? 1.0f : 0.0f;
                const float av[4] = {a01.x * mk, a01.y * mk, a23.x * mk, a23.y * mk};
                u32x2_t q1, q2, q3;
                split_x4<NP>(av, q1, q2, q3);
                uint16_t* dst = A3 + r * 16 + 4 * g;
                *reinterpret_cast<u32x2_t*>(dst) = q1;
                if (NP == 3) {
                    *reinterpret_cast<u32x2_t*>(dst + A3_PIECE) = q2;
                    *reinterpret_cast<u32x2_t*>(dst + 2 * A3_PIECE) = q3;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                {
                    const uint16_t* pbp = A3 + (4 * g + (r >> 2)) * 16 + 4 * (r & 3);
                    s16x4_t at[3];
                    at[0] = lds_tr4(pbp);
                    if (NP == 3) { at[1] = lds_tr4(pbp + A3_PIECE); at[2] = lds_tr4(pbp + 2 * A3_PIECE); } else { at[1] = at[0]; at[2] = at[0]; }
#pragma unroll
                    for (int mt = 0; mt < NM; ++mt) {
                        const uint16_t* pa = dyb + (pb * 16 + 4 * g + (r >> 2)) * E + 16 * mt + 4 * (r & 3);
                        s16x4_t dt[3];
                        dt[0] = lds_tr4(pa);
                        if (NP == 3) { dt[1] = lds_tr4(pa + DY_PIECE); dt[2] = lds_tr4(pa + 2 * DY_PIECE); } else { dt[1] = dt[0]; dt[2] = dt[0]; }
                        mfmaN_16<NP>(acc3[mt], dt, at);
                    }
                }
                __builtin_amdgcn_wave_barrier();   // the image is rewritten by the next block
            }
            if (more) {
                dy_store(slot ^ 1, nd);
#pragma unroll
                for (int pb = 0; pb < 3; ++pb) h3c[pb] = h3n[pb];
            }
            STAMP(3 + 2 * c);
            __syncthreads();   // dy slot c + 1 complete; the readers of slot c are done before chunk c + 2 rewrites it; after the last chunk: ring rows complete
            STAMP(4 + 2 * c);
        }
    };

    DyVec pre;
    typename HS<BF>::raw4 h3p[3];
    dy_fetch(Y0 - 1, 2 * HX, 0, pre);
    h3_fetch(Y0 - 1, 0, h3p);
    // the first step's chunk-0 operands are requested at the top of the prologue's only chunk, the next step's before P2 (below): an HBM
    // round trip under a whole phase instead of in front of it
    DyVec npre;
    typename HS<BF>::raw4 nh3[3];
    compute_rows(Y0 - 1, 2, pre, h3p, [&] { dy_fetch(Y0 + 1, TY * HX, 0, npre); h3_fetch(Y0 + 1, 0, nh3); });
#pragma unroll 1
    for (int y0 = Y0; y0 < Yend; y0 += TY) {
#ifdef LG_STAMPS
        stamp_on = (y0 == Y0 + 2 * TY) && strip == 0;
#endif
        STAMP(0);
        // h2 of the step's output pixels: wave w owns tile rows RPW w .. RPW w + RPW - 1, a lane the PPL consecutive pixels
        // x0 + PPL lgrp + it of a row, channels 4 q ..  An HBM round trip is longer than one P2 item, so four vectors are kept in flight:
        // items 0 .. 3 are requested at the top of the LAST chunk of the halo pass, the item four places on while item i is worked on.
        // Every output pixel of a step is inside the image (launcher: h % 8 == 0, w % 16 == 0).
        const int yrow0 = y0 + RPW * wave, xl0 = x0 + PPL * lgrp;
        auto h2_at = [&](int y, int x) { return HS<BF>::ldraw(a.h2, ((b * h + y) * (long)w + x) * N1 + hoff + 4 * q); };
        typename HS<BF>::raw4 h2a, h2b, h2c, h2d;
        pre = npre;
#pragma unroll
        for (int pb = 0; pb < 3; ++pb) h3p[pb] = nh3[pb];
        compute_rows(y0 + 1, TY, pre, h3p, [&] { h2a = h2_at(yrow0, xl0); h2b = h2_at(yrow0, xl0 + 1); h2c = h2_at(yrow0, xl0 + 2); h2d = h2_at(yrow0, xl0 + 3); });
        // the depthwise taps of the lane's four channels as channel pairs: 36 contiguous floats from LDS per step instead of 36 VGPRs pinned
        // through the halo pass
        lg_v2f wq01[9], wq23[9];
        {
            const float* tp = sTaps + 36 * q;
            float t36[36];
#pragma unroll
            for (int k4 = 0; k4 < 9; ++k4) {
                const float4 v = *reinterpret_cast<const float4*>(tp + 4 * k4);
                t36[4 * k4] = v.x; t36[4 * k4 + 1] = v.y; t36[4 * k4 + 2] = v.z; t36[4 * k4 + 3] = v.w;
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) { wq01[k] = (lg_v2f){t36[k], t36[9 + k]}; wq23[k] = (lg_v2f){t36[18 + k], t36[27 + k]}; }
        }
        // next step's chunk-0 operands (clamped addresses: harmless behind the strip's last step), requested BEHIND the first h2 vectors: the
        // wait counter is in-order
        __builtin_amdgcn_sched_barrier(0);
        dy_fetch(y0 + TY + 1, TY * HX, 0, npre);
        h3_fetch(y0 + TY + 1, 0, nh3);
        STAMP(9);
        const int sbase = (y0 - Y0) % RING;            // ring slot of row y0 - 1
        // A lane works on PPL CONSECUTIVE pixels of a tile row and slides a 3 x 3 window of dh3 vectors over them: item `it` needs ring
        // columns it .. it + 2, so a row of PPL items reads (PPL + 2) x 3 vectors instead of PPL x 9 -- the P2 phase was LDS-read bound
        // (profiles/r03_ffn_bwd_phase_stamps.txt).  hreg holds an item's h2 vector and is re-loaded with the vector of the item four places
        // on in the lane's walk (no register rotation: a rotating set needs its moves in front of the loop's back edge, i.e. a wait for
        // the load issued in the same iteration).
        auto rowN = [&](int ch) {
            const int ty = RPW * wave + ch, y = y0 + ty, txb = PPL * lgrp;
            const float* rrow[3];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {       // ring row ty + rr (relative to row y0 - 1) meets tap dy = 2 - rr
                int sl = sbase + ty + rr;
                sl = sl >= RING ? sl - RING : sl;
                rrow[rr] = ring + (sl * HX + txb) * LDR + 4 * q;
            }
            float4 col[4][3];                      // the window in a ring of four column slots: item `it` reads slots it .. it + 2 (mod 4)
            auto ldcol = [&](int c) {
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) col[c & 3][rr] = *reinterpret_cast<const float4*>(rrow[rr] + c * LDR);
            };
            ldcol(0); ldcol(1); ldcol(2);
            auto item = [&](int it, typename HS<BF>::raw4& hreg) {
                if (it + 1 < PPL) ldcol(it + 3);    // the next item's new column, into the slot this item does not read: requested under its arithmetic
                const int x = x0 + txb + it;
                const float4 hc = HS<BF>::widen(hreg);
                {   // the item four places on: same row while it lasts, then the lane's next row (clamped to the step: a repeat nobody waits for)
                    const int nit = it + 4, ny = nit < PPL ? y : (ch + 1 < RPW ? y + 1 : y), nx = x0 + txb + (nit < PPL ? nit : nit - PPL);
                    hreg = h2_at(ny, nx);
                }
                const lg_v2f h01 = (lg_v2f){hc.x, hc.y}, h23 = (lg_v2f){hc.z, hc.w};
                lg_v2f acc01 = (lg_v2f){0.f, 0.f}, acc23 = (lg_v2f){0.f, 0.f};
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        // forward: h3(p) += w[dy][dx] h2(p + (dy-1, dx-1))  ->  h2(q) meets dh3(q - (dy-1, dx-1)) in both sums: halo column
                        // tx + 2 - dx = tx + cc, ring row ty + 2 - dy = ty + rr
                        const int k = (2 - rr) * 3 + (2 - cc);
                        const float4 gv = col[(it + cc) & 3][rr];
                        const lg_v2f g01 = (lg_v2f){gv.x, gv.y}, g23 = (lg_v2f){gv.z, gv.w};
                        acc01 = wq01[k] * g01 + acc01;
                        acc23 = wq23[k] * g23 + acc23;
                        pw01[k] = h01 * g01 + pw01[k];
                        pw23[k] = h23 * g23 + pw23[k];
                        if (k == 4) { pw01[9] += g01; pw23[9] += g23; }
                    }
                HS<BF>::st4(a.dh2, ((b * h + y) * (long)w + x) * N1 + hoff + 4 * q, make_float4(acc01.x, acc01.y, acc23.x, acc23.y));
                if (NP != 1) dmx = fmaxf(fmaxf(dmx, fmaxf(fabsf(acc01.x), fabsf(acc01.y))), fmaxf(fabsf(acc23.x), fabsf(acc23.y)));
                if (LG_KA_ITEMFENCE && ((it + 1) % LG_KA_ITEMFENCE) == 0) __builtin_amdgcn_sched_barrier(0);   // one item at a time: interleaved items need more registers than there are
            };
            item(0, h2a); item(1, h2b); item(2, h2c); item(3, h2d);
        };
#pragma unroll 1
        for (int ch = 0; ch < RPW; ++ch) rowN(ch);
        STAMP(10);
    }   // steps of the strip
    __syncthreads();   // the last step's P2 readers of the ring are done before the next strip's prologue writes it
    }   // strips of this workgroup

    if (NP != 1 && a.dh2_max) {     // one atomic per wave (non-negative floats order as their bit patterns)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dmx = fmaxf(dmx, __shfl_xor(dmx, off));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(a.dh2_max), __float_as_uint(dmx));
    }
    // ---- this workgroup's partial sums -> its slab row [d dww 64 x 9 | d dwb 64 | dW3 E x 64 | db3 E] of its channel half
    float* row = a.slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * C::ROW;
#pragma unroll
    for (int mt = 0; mt < NM; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) row[C::R_W3 + (16 * mt + 4 * g + v) * NH + 16 * wave + r] = acc3[mt][v];
    float* red = ring;    // [NW waves][CQ quads][40] | [NW waves][E]: the ring is dead (barrier at the end of the last strip)
    // depthwise partials: lanes with the same q (LGW per wave) hold the same channels
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            float v = u == 0 ? pw01[k].x : (u == 1 ? pw01[k].y : (u == 2 ? pw23[k].x : pw23[k].y));
#pragma unroll
            for (int off = CQ; off < 64; off <<= 1) v += __shfl_xor(v, off);
            if (lane < CQ) red[(wave * CQ + q) * 40 + u * 10 + k] = v;
        }
    // db3: dy threads with the same lq hold the same channels
    {
        float4 s4 = sb3;
#pragma unroll
        for (int off = LQ; off < 64; off <<= 1) {
            s4.x += __shfl_xor(s4.x, off); s4.y += __shfl_xor(s4.y, off); s4.z += __shfl_xor(s4.z, off); s4.w += __shfl_xor(s4.w, off);
        }
        if (lane < LQ) *reinterpret_cast<float4*>(red + NW * CQ * 40 + wave * E + 4 * lane) = s4;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CQ * 40; i += NT) {
        float v = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < NW; ++w8) v += red[w8 * CQ * 40 + i];
        const int qq = i / 40, rem = i - qq * 40, u = rem / 10, k = rem - u * 10;
        const int c = 4 * qq + u;
        if (k < 9) row[c * 9 + k] = v;
        else row[C::R_DB + c] = v;
    }
    if (threadIdx.x < E) {
        const float* sp = red + NW * CQ * 40 + threadIdx.x;
        float v = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < NW; ++w8) v += sp[w8 * E];
        row[C::R_B3 + threadIdx.x] = v;
    }
}

template <int E>
static int launch_dw_t(const FfnDwBwdXArgs& a, hipStream_t s) {
    using C = KA<E>;
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn_dw_bwd_xs<E, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_dw_bwd_xs<E, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) { lg_set_error("ffn_dw_bwd_xs: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    const int tiles_x = (a.w + 15) / 16;
    // strip height as in the forward: the tallest multiple of 8 rows that still yields a strip per resident workgroup (512 = two per CU; at
    // e = 32 the two channel halves share them), at least 16
    const int wgs = FFN_DW_BWD_X_WGS / C::NHALF;
    int SH = (a.h + 7) / 8 * 8;
    while (SH > 16 && (long)a.B * tiles_x * ((a.h + SH - 1) / SH) < wgs) SH = (SH / 2 + 7) / 8 * 8;
    const int strips_y = (a.h + SH - 1) / SH;
    const int nstrips = a.B * tiles_x * strips_y;
    const int gx = nstrips < wgs ? nstrips : wgs;
    const dim3 grid(gx, C::NHALF);
#ifndef LG_DWB_UNEVEN
#define LG_DWB_UNEVEN 0   // measured: even strips 110.5 us, 9 : 7 steps 113.2, 10 : 6 110.5 -- this kernel gains nothing from the uneven split (unlike k_ffn_xr, k_attn_m, k_ffn1_bwd_xs)
#endif
    int dS = 0;   // uneven strip pairs: the measured shape only (e = 16: one strip per workgroup, exactly two workgroups per CU)
    if (LG_DWB_UNEVEN && E == 16 && nstrips == wgs && wgs == 512 && (strips_y & 1) == 0 && SH >= 32 && a.h % (2 * SH) == 0) dS = (SH * LG_DWB_UNEVEN / 64 + 7) / 8 * 8;
    if (a.hbf) k_ffn_dw_bwd_xs<E, 1><<<grid, NT, C::LDS_BYTES, s>>>(a, tiles_x, strips_y, nstrips, SH, dS);   // precision = 'bf16'
    else k_ffn_dw_bwd_xs<E, 3><<<grid, NT, C::LDS_BYTES, s>>>(a, tiles_x, strips_y, nstrips, SH, dS);
    LG_CHECK_LAUNCH();
    // the slab rows of each channel half, summed in a fixed order by the deferred reduce launch
    ReduceJob j;
    j.dst2 = nullptr; j.nslices = gx; j.slice_stride = C::ROW;
    int rc = 0;
    for (int half = 0; half < C::NHALF && !rc; ++half) {
        const float* base = a.slab + (size_t)half * gx * C::ROW;
        auto job = [&](int off, float* dst, int rows, int cols, int ld) {
            j.slab = base + off; j.dst = dst; j.rows = rows; j.cols = cols; j.row_stride = cols; j.ld = ld; j.rows_valid = rows; j.cols_valid = cols;
            return launch_reduce_job(j, s);
        };
        rc = job(0, a.d_dww + (size_t)half * NH * 9, NH, 9, 9);
        if (!rc) rc = job(C::R_DB, a.d_dwb + half * NH, 1, NH, NH);
        if (!rc) rc = job(C::R_W3, a.d_w3 + half * NH, E, NH, C::N1);
        if (!rc && half == 0) rc = job(C::R_B3, a.d_b3, 1, E, E);     // db3 = sum of dy: every half sums it, one is used
    }
    return rc;
}

}   // namespace

int launch_ffn_dw_bwd_xs(int e, const FfnDwBwdXArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_FFN2_BWD, s);
    if (e != 16 && e != 32) { lg_set_error("ffn_dw_bwd_xs: e=%d unsupported", e); return -1; }
    if (!a.dy || !a.h3 || !a.h2 || !a.dh2 || !a.w3t || !a.dww || !a.slab) { lg_set_error("ffn_dw_bwd_xs: null argument"); return -2; }
    if ((a.h & 7) || (a.w & 15)) { lg_set_error("ffn_dw_bwd_xs: h must be a multiple of 8 and w of 16 (got %d x %d)", a.h, a.w); return -2; }
    return e == 16 ? launch_dw_t<16>(a, s) : launch_dw_t<32>(a, s);
}
