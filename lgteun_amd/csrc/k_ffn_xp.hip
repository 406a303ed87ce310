// k_ffn_xp: k_ffn_xs (k_ffn_x.hip: fused feed_forward half-block at e = 16, fp32-equivalent split-bf16 GEMMs, strip walk with an LDS
// ring of h2; reference models/common/LGT.py:91-109, :45-61) with the halo pass SOFTWARE-PIPELINED so that no phase between two
// barriers is matrix-only or vector-only:
//   k_ffn_xs ran  [GEMM1 + GELU(c)] | barrier | [GEMM2(c) + LN(c+1)] | barrier  per 48-pixel chunk; counters and in-kernel stamps
//   showed its VALU pipe 62 % busy: a wave alone issues one VALU instruction per ~4.7 cycles, the pipe takes one per ~2.6, so whenever
//   one of the two waves of a SIMD sits in an MFMA-only phase or at a barrier the other cannot fill the pipe.
//   Here chunks are 32 pixels, the gelu(h1) pieces are double-buffered, and ONE stage between two barriers does
//        LN(c+2) -> XA[c&1]   |   GEMM1(c+1) -> GELU + split -> A2[(c+1)&1]   |   GEMM2(c) from A2[c&1] -> ring
//   with GEMM2(c)'s MFMAs issued between the VALU instructions of GELU(c+1) (sched_group_barrier).  LayerNorm runs on all 256
//   threads (8 lanes per pixel), branch-free.  A2 / G3 rows are 128 bytes with their 16-byte chunks XOR-swizzled by the pixel index
//   (conflict-free fragment reads, no padding).
// MEASURED (bs 32, 128x128, e = 16): 130.3 us against k_ffn_xs's 127.1 us, outputs bitwise equal; LDS bank-conflict cycles 11.9 M vs
// 16.1 M, VALU instructions +4 %, wait cycles unchanged: at two waves per SIMD the bf16 MFMAs still cost vector-issue slots and the
// waits are LDS / memory latency, not the barriers.  Kept selectable (LG_FFN_IMPL=xp) as the A/B record; k_ffn_xs stays the default.
// LDS (79.7 KB, two workgroups per CU): ring [10][18][68] fp32 | A2 [2][3][32][64] bf16 | XA [2][3][32][16] bf16; the per-wave gelu(h3)
// pieces [4][3][16][64] of the depthwise phase alias A2.
#include "kernels.h"

#include <type_traits>

#include "hstore.h"
#include "split_bf16.h"

namespace {

constexpr int E = 16, N1 = 64, TX = 16, HX = 18, TY = 8, RING = 10, LDR = 68, CH = 32, CQ = 16;
constexpr int A2_BUF = 3 * CH * N1;              // 6144 halves per buffer
constexpr int XA_SLOT = 3 * CH * E;              // 1536 halves per slot
constexpr int G3_WAVE = 3 * 16 * N1;             // 3072 halves per wave
constexpr size_t LDS_BYTES = (size_t)RING * HX * LDR * 4 + (size_t)(2 * A2_BUF + 2 * XA_SLOT) * 2;
static_assert(4 * G3_WAVE <= 2 * A2_BUF, "gelu(h3) pieces must fit in the aliased region");

__device__ __forceinline__ bf16x8_t lds_x8(const uint16_t* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(p)); }
__device__ __forceinline__ s16x4_t lds_x4(const uint16_t* p) { return __builtin_bit_cast(s16x4_t, *reinterpret_cast<const u32x2_t*>(p)); }
// sum over 8 consecutive lanes with DPP moves only (quad_perm x 2, then row_half_mirror: lane i <-> 7 - i swaps the two quads)
__device__ __forceinline__ float oct_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return v;
}
// element offset (halves) of logical 16-byte chunk `c` in row `px` of a 64-channel bf16 row (8 chunks), swizzled
__device__ __forceinline__ int swz(int px, int c) { return px * N1 + ((c ^ (px & 7)) << 3); }

template <bool SAVE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_ffn_xp(Ffn1Args a1, Ffn2Args a2, int tiles_x, int strips_y, int nstrips,
                                                                                       int SH) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* ring = reinterpret_cast<float*>(smem_raw);                                   // [RING*HX][LDR]
    uint16_t* A2 = reinterpret_cast<uint16_t*>(smem_raw + (size_t)RING * HX * LDR * 4);   // [2][3][CH][N1], chunk-swizzled
    uint16_t* XA = A2 + 2 * A2_BUF;                                                     // [2][3][CH][E]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    uint16_t* G3 = A2 + wave * G3_WAVE;                                                 // [3][16][N1], aliases A2
    const int h = a2.h, w = a2.w;
    __shared__ __attribute__((aligned(16))) float sPar[5 * E];
    __shared__ __attribute__((aligned(16))) float sMask[4][CH];   // 1 for halo pixels inside the image (dep_conv zero-pads h2); slot = chunk & 3
    float* sLn2g = sPar;            float* sLn2b = sPar + E;
    float* sN1g = sPar + 2 * E;     float* sN1b = sPar + 3 * E;
    float* sB3 = sPar + 4 * E;
    for (int i = threadIdx.x; i < E; i += 256) {
        sLn2g[i] = a1.ln2g[i]; sLn2b[i] = a1.ln2b[i]; sB3[i] = a2.b3[i];
        sN1g[i] = a2.g ? a2.n1g[i] : 0.f; sN1b[i] = a2.g ? a2.n1b[i] : 0.f;
    }
    // ---- weights: split once, register-resident for every strip of this workgroup
    const int c0 = wave * 16 + 4 * g;                 // first of the four h1 / h2 channels this lane holds after GEMM1 / GEMM2
    const float4 b1v = *reinterpret_cast<const float4*>(a1.b1 + c0);
    const float4 b2v = *reinterpret_cast<const float4*>(a1.b2 + c0);
    const WFrag16 w1f = load_wfrag16(a1.w1 + (size_t)(wave * 16) * E, E, 0);
    const WFrag32 w2f0 = load_wfrag32(a1.w2 + (size_t)(wave * 16) * N1, N1, 0);
    const WFrag32 w2f1 = load_wfrag32(a1.w2 + (size_t)(wave * 16) * N1, N1, 1);
    const WFrag32 w3f0 = load_wfrag32(a2.w3, N1, 0);
    const WFrag32 w3f1 = load_wfrag32(a2.w3, N1, 1);
    const int q = lane % CQ;                          // depthwise phase: lane = (pixel slot lane / 16, channel quad q)
    // LayerNorm phase: thread t = (chunk pixel t / 8, channel pair t % 8)
    const int lpx = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    __syncthreads();
    const float2 lng = *reinterpret_cast<const float2*>(sLn2g + 2 * l8), lnb = *reinterpret_cast<const float2*>(sLn2b + 2 * l8);

#pragma unroll 1
    for (int strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
    int t = strip;
    const int tx_i = t % tiles_x;
    t /= tiles_x;
    const int sy = t % strips_y;
    const long b = t / strips_y;
    const int x0 = tx_i * TX, Y0 = sy * SH, Yend = min(Y0 + SH, h);

    // x pair of halo pixel m = 32 c + lpx of the row block starting at ya.  Branch-free: the address is clamped into the image and
    // `in` says whether the pixel exists (outside: LN output and mask are zero -- dep_conv zero-pads h2)
    auto ln_fetch = [&](int ya, int npx, int c, float2& xv, float& in) {
        const int m = c * CH + lpx;
        const int hy = m / HX, hx = m - hy * HX;
        const int y = ya + hy, x = x0 + hx - 1;
        in = ((m < npx) && y >= 0 && y < h && x >= 0 && x < w) ? 1.0f : 0.0f;
        const int yc = min(max(y, 0), h - 1), xc = min(max(x, 0), w - 1);
        xv = *reinterpret_cast<const float2*>(a1.x + ((b * h + yc) * (long)w + xc) * E + 2 * l8);
    };
    // LayerNorm over the 16 channels of a pixel = 8 consecutive lanes; pieces -> XA[c & 1], mask -> sMask[c & 3]
    auto ln_store = [&](int c, const float2& xv, float in) {
        const float mu = oct_sum(xv.x + xv.y) * (1.0f / E);
        const float d0 = xv.x - mu, d1 = xv.y - mu;
        const float rstd = __builtin_amdgcn_rsqf(oct_sum(d0 * d0 + d1 * d1) * (1.0f / E) + LG_EPS);
        const Split3 s0 = split3((d0 * rstd * lng.x + lnb.x) * in), s1 = split3((d1 * rstd * lng.y + lnb.y) * in);
        uint32_t* dst = reinterpret_cast<uint32_t*>(XA + (c & 1) * XA_SLOT + lpx * E + 2 * l8);
        dst[0] = pack_hi16(s0.p1, s1.p1);
        dst[CH * E / 2] = pack_hi16(s0.p2, s1.p2);
        dst[CH * E] = pack_hi16(s0.p3, s1.p3);
        if (l8 == 0) sMask[c & 3][lpx] = in;
    };

    // h2 of halo rows [ya, ya + nr) x columns [x0 - 1, x0 + 17) -> ring   (nr = 2: strip prologue, 8: one step).
    // pre / pre_in: chunk 0's x pair, fetched by the caller ahead of time.
    auto compute_rows = [&](int ya, int nr, float2 pre, float pre_in) {
        const int npx = nr * HX, nch = (npx + CH - 1) / CH;
        const int ring0 = ((ya - Y0 + 1) % RING) * HX;
        f32x4_t h1acc[2];

        // GEMM1 of chunk c (K = 16): h1[16 w .. +15][32 pixels] = W1 LN(x) + b1, left in h1acc
        auto gemm1 = [&](int c, auto nbt) {
            constexpr int NB = decltype(nbt)::value;    // pixel blocks of the chunk that hold halo pixels (the last chunk: 1)
            const uint16_t* xa = XA + (c & 1) * XA_SLOT + r * E + 4 * g;
#pragma unroll
            for (int pb = 0; pb < NB; ++pb) {
                h1acc[pb] = (f32x4_t){b1v.x, b1v.y, b1v.z, b1v.w};
                const uint16_t* p = xa + pb * 16 * E;
                mfma_split16(h1acc[pb], w1f, lds_x4(p), lds_x4(p + CH * E), lds_x4(p + 2 * CH * E));
            }
        };
        // GELU + split of h1acc -> A2[c & 1]
        auto gelu1 = [&](int c, auto nbt) {
            constexpr int NB = decltype(nbt)::value;
            uint16_t* a2b = A2 + (c & 1) * A2_BUF;
#pragma unroll
            for (int pb = 0; pb < NB; ++pb) {
                float av[4];
                if (SAVE) {
                    const int m = c * CH + pb * 16 + r;
                    const int hy = m / HX, hx = m - hy * HX;
                    const int y = ya + hy, x = x0 + hx - 1;
                    const bool inner = m < npx && hx >= 1 && hx <= TX && x < w && y >= Y0 && y < Yend;
                    lg_v2f a01, a23, g01, g23;
                    gelu2_both_f((lg_v2f){h1acc[pb][0], h1acc[pb][1]}, a01, g01);
                    gelu2_both_f((lg_v2f){h1acc[pb][2], h1acc[pb][3]}, a23, g23);
                    av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                    if (inner) {
                        const long prow = ((b * h + y) * (long)w + x) * N1 + c0;
                        HS<false>::st4(a1.a1s, prow, make_float4(av[0], av[1], av[2], av[3]));
                        HS<false>::st4(a1.g1s, prow, make_float4(g01.x, g01.y, g23.x, g23.y));
                    }
                } else {
                    const lg_v2f a01 = gelu2_f((lg_v2f){h1acc[pb][0], h1acc[pb][1]}), a23 = gelu2_f((lg_v2f){h1acc[pb][2], h1acc[pb][3]});
                    av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                }
                u32x2_t q1, q2, q3;
                split3_x4(av, q1, q2, q3);
                const int px = pb * 16 + r;
                uint16_t* dst = a2b + swz(px, 2 * wave + (g >> 1)) + 4 * (g & 1);
                *reinterpret_cast<u32x2_t*>(dst) = q1;
                *reinterpret_cast<u32x2_t*>(dst + CH * N1) = q2;
                *reinterpret_cast<u32x2_t*>(dst + 2 * CH * N1) = q3;
            }
        };
        // GEMM2 of chunk c (K = 64) from A2[c & 1]: accumulators only
        auto gemm2 = [&](int c, f32x4_t (&acc)[2], auto nbt) {
            constexpr int NB = decltype(nbt)::value;
            const uint16_t* a2b = A2 + (c & 1) * A2_BUF;
#pragma unroll
            for (int pb = 0; pb < NB; ++pb) acc[pb] = (f32x4_t){b2v.x, b2v.y, b2v.z, b2v.w};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int pb = 0; pb < NB; ++pb) {
                    const int px = pb * 16 + r;
                    const uint16_t* p = a2b + swz(px, 4 * kb + g);
                    mfma_split32(acc[pb], kb ? w2f1 : w2f0, lds_x8(p), lds_x8(p + CH * N1), lds_x8(p + 2 * CH * N1));
                }
        };
        // mask, (save,) -> ring
        auto ring_store = [&](int c, const f32x4_t (&acc)[2], auto nbt) {
            constexpr int NB = decltype(nbt)::value;
#pragma unroll
            for (int pb = 0; pb < NB; ++pb) {
                const int m = c * CH + pb * 16 + r;
                const float mk = sMask[c & 3][pb * 16 + r];
                const float4 hh = make_float4(acc[pb][0] * mk, acc[pb][1] * mk, acc[pb][2] * mk, acc[pb][3] * mk);
                if (SAVE) {
                    const int hy = m / HX, hx = m - hy * HX;
                    const int y = ya + hy, x = x0 + hx - 1;
                    if (m < npx && hx >= 1 && hx <= TX && x < w && y >= Y0 && y < Yend)
                        HS<false>::st4(a1.h2, ((b * h + y) * (long)w + x) * N1 + c0, hh);
                }
                int rp = ring0 + m;
                rp = rp >= RING * HX ? rp - RING * HX : rp;
                if (m < npx) *reinterpret_cast<float4*>(ring + rp * LDR + c0) = hh;
            }
        };

        constexpr std::integral_constant<int, 1> one{};
        constexpr std::integral_constant<int, 2> two{};
        // one stage between two barriers: LN(i+2) | GEMM1(i+1) -> GELU(i+1) | GEMM2(i) -> ring ; nb1 = pixel blocks of chunk i+1
        auto stage = [&](int i, float2& nx, float& nin, auto nb1) {
            if (i + 2 < nch) ln_store(i + 2, nx, nin);
            ln_fetch(ya, npx, i + 3, nx, nin);
            f32x4_t acc[2];
            gemm1(i + 1, nb1);
            gemm2(i, acc, two);
            gelu1(i + 1, nb1);
            // interleave: one MFMA, an LDS read if one is due, then a few VALU instructions of GELU(i+1)
#pragma unroll
            for (int k = 0; k < 24 + 6 * decltype(nb1)::value; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
            ring_store(i, acc, two);
            __syncthreads();
        };
        const bool last_half = npx - (nch - 1) * CH <= 16;      // the last chunk holds one pixel block (both row counts in use: yes)
        float2 nx;
        float nin;
        __syncthreads();                 // the previous phase's readers of the aliased region (gelu(h3) pieces) are done
        ln_store(0, pre, pre_in);
        ln_fetch(ya, npx, 1, nx, nin);
        __syncthreads();
        ln_store(1, nx, nin);            // nch >= 2 always (36 or 144 halo pixels)
        ln_fetch(ya, npx, 2, nx, nin);
        gemm1(0, two);
        gelu1(0, two);
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i + 2 < nch; ++i) stage(i, nx, nin, two);
        f32x4_t acc[2];
        if (last_half) {
            stage(nch - 2, nx, nin, one);
            gemm2(nch - 1, acc, one);
            ring_store(nch - 1, acc, one);
        } else {
            stage(nch - 2, nx, nin, two);
            gemm2(nch - 1, acc, two);
            ring_store(nch - 1, acc, two);
        }
    };

    {
        float2 pre;
        float pin;
        ln_fetch(Y0 - 1, 2 * HX, 0, pre, pin);
        compute_rows(Y0 - 1, 2, pre, pin);
    }
    float2 pre;
    float pin;
    ln_fetch(Y0 + 1, TY * HX, 0, pre, pin);
#pragma unroll 1
    for (int y0 = Y0; y0 < Yend; y0 += TY) {
    // the residual rows of the epilogue are requested first: their HBM round trip hides under the whole step.
    // wave w owns tile rows 2 w and 2 w + 1; lane (r, g): pixel x0 + r, channels 4 g .. 4 g + 3
    float4 xres[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const int y = y0 + 2 * wave + ch, x = x0 + r;
        xres[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y < Yend && x < w) xres[ch] = *reinterpret_cast<const float4*>(a2.x + ((b * h + y) * (long)w + x) * E + 4 * g);
    }
    compute_rows(y0 + 1, TY, pre, pin);
    if (y0 + TY < Yend) ln_fetch(y0 + TY + 1, TY * HX, 0, pre, pin);   // next step's first chunk: in flight during the output phase
    float wq[4][9], bq[4];
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
    }
    __syncthreads();                                  // the ring rows of this step are complete
    // ---- output phase: per wave, 2 tile rows of 16 pixels: dw3x3 over the ring + GELU -> pieces -> GEMM3 -> bias + residual -> y (+ planar LN half)
    {
        const int sbase = (y0 - Y0) % RING;            // ring slot of row y0 - 1
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int ty = 2 * wave + ch;
#pragma unroll(SAVE ? 1 : 4)
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
                        const float4 v = *reinterpret_cast<const float4*>(ring + (sl * HX + tx + dx) * LDR + 4 * q);
                        acc.x += wq[0][dy * 3 + dx] * v.x; acc.y += wq[1][dy * 3 + dx] * v.y;
                        acc.z += wq[2][dy * 3 + dx] * v.z; acc.w += wq[3][dy * 3 + dx] * v.w;
                    }
                }
                float av[4];
                if (SAVE) {
                    lg_v2f a01, a23, g01, g23;
                    gelu2_both_f((lg_v2f){acc.x, acc.y}, a01, g01);
                    gelu2_both_f((lg_v2f){acc.z, acc.w}, a23, g23);
                    av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                    const int y = y0 + ty, x = x0 + tx;
                    if (y < Yend && x < w) {
                        const long o = ((b * h + y) * (long)w + x) * N1 + 4 * q;
                        HS<false>::st4(a2.a3s, o, make_float4(av[0], av[1], av[2], av[3]));
                        HS<false>::st4(a2.g3s, o, make_float4(g01.x, g01.y, g23.x, g23.y));
                    }
                } else {
                    const lg_v2f a01 = gelu2_f((lg_v2f){acc.x, acc.y}), a23 = gelu2_f((lg_v2f){acc.z, acc.w});
                    av[0] = a01.x; av[1] = a01.y; av[2] = a23.x; av[3] = a23.y;
                }
                u32x2_t q1, q2, q3;
                split3_x4(av, q1, q2, q3);
                uint16_t* dst = G3 + swz(tx, q >> 1) + 4 * (q & 1);
                *reinterpret_cast<u32x2_t*>(dst) = q1;
                *reinterpret_cast<u32x2_t*>(dst + 16 * N1) = q2;
                *reinterpret_cast<u32x2_t*>(dst + 2 * 16 * N1) = q3;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- GEMM3 (K = 64): out[16 channels][16 pixels of tile row ty]
            const float4 b3v = *reinterpret_cast<const float4*>(sB3 + 4 * g);
            f32x4_t o = (f32x4_t){b3v.x, b3v.y, b3v.z, b3v.w};
            {
                const uint16_t* p0 = G3 + swz(r, g), *p1 = G3 + swz(r, 4 + g);
                mfma_split32(o, w3f0, lds_x8(p0), lds_x8(p0 + 16 * N1), lds_x8(p0 + 2 * 16 * N1));
                mfma_split32(o, w3f1, lds_x8(p1), lds_x8(p1 + 16 * N1), lds_x8(p1 + 2 * 16 * N1));
            }
            __builtin_amdgcn_wave_barrier();           // G3 is rewritten by the next tile row
            // ---- epilogue in registers: residual, store, LayerNorm statistics of the next block across the four lane groups
            const int y = y0 + ty, x = x0 + r;
            const float o0 = o[0] + xres[ch].x, o1 = o[1] + xres[ch].y, o2 = o[2] + xres[ch].z, o3 = o[3] + xres[ch].w;
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
        }
    }
    }   // steps of the strip
    }   // strips of this workgroup
}

}   // namespace

int launch_ffn_xp(const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s) {
    ProfScope prof__(LG_K_FFN2, s);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn_xp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_xp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) { lg_set_error("ffn_xp: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    const int tiles_x = (a2.w + 15) / 16;
    // strip height: the tallest multiple of 8 rows that still yields >= 512 strips (two resident workgroups per CU), at least 16
    int SH = (a2.h + 7) / 8 * 8;
    while (SH > 16 && (long)a2.B * tiles_x * ((a2.h + SH - 1) / SH) < 512) SH = (SH / 2 + 7) / 8 * 8;
    const int strips_y = (a2.h + SH - 1) / SH;
    const int nstrips = a2.B * tiles_x * strips_y;
    const int grid = nstrips < 512 ? nstrips : 512;
    if (a1.a1s != nullptr) k_ffn_xp<true><<<grid, 256, LDS_BYTES, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
    else k_ffn_xp<false><<<grid, 256, LDS_BYTES, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
    LG_CHECK_LAUNCH();
    return 0;
}
