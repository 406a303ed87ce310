// feed_forward of an LGB block for gfx950 -- reference models/common/LGT.py:91-109 (+ pre_norm/residual 45-61).
//   y = x + W3 gelu( dw3x3( W2 gelu( W1 LN(x) + b1 ) + b2 ) ) + b3
// The three 1x1 convs are per-pixel GEMMs ([pixels, K] x [K, N], K,N in 16..256) and run on the matrix cores
// with the exact-fp32 MFMA v_mfma_f32_16x16x4_f32 (parity mode: bit-for-bit an fp32 fma chain).
// Two kernels, split at the depthwise conv (the only spatial coupling):
//   k_ffn1: LN -> GEMM1 -> GELU -> GEMM2 -> h2            (hidden a1 never leaves LDS)
//   k_ffn2: dw3x3 + GELU (halo read from h2 through L2) -> GEMM3 -> +residual, and emits the LayerNorm-ed
//           global half the next block's FFT mixer consumes.
// Operand staging: activations sit in LDS as [pixel][K] rows; a lane fetches 4 consecutive k (one ds_read_b128 /
// global_load_dwordx4 of the [N][K] weight row) and feeds them to 4 consecutive MFMAs -- the k order inside a
// 16-deep block is permuted identically for A and B, which a dot product does not care about.
#include "kernels.h"

#include "mfma.h"
#include "hstore.h"

// ------------------------------------------------------------------------------------------------
// k_ffn1: each wave owns MW = 16*MT pixels end to end (no inter-wave dependency)
// ------------------------------------------------------------------------------------------------
template <int E, int MT, bool BF>
__global__ __launch_bounds__(256) void k_ffn1(Ffn1Args a) {
    constexpr int N1 = 4 * E, MW = 16 * MT, LDA = E + 4, LDH = N1 + 4;
    constexpr int LPP = 64 / MW;       // lanes per pixel in the load/LN phase (2 or 4)
    constexpr int CPL = E / LPP;       // channels per lane
    extern __shared__ float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    float* bufA = smem + wave * (MW * LDA);
    float* bufH = smem + 4 * MW * LDA + wave * (MW * LDH);
    const long p0 = ((long)blockIdx.x * 4 + wave) * MW;
    // ---- load + LayerNorm (LGT.py:58)
    {
        const int m = lane % MW, part = lane / MW;
        long p = p0 + m;
        if (p >= a.P) p = a.P - 1;
        float xv[CPL];
        const float4* src = reinterpret_cast<const float4*>(a.x + p * E + part * CPL);
#pragma unroll
        for (int k = 0; k < CPL / 4; ++k) {
            float4 v = src[k];
            xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) s += xv[k];
#pragma unroll
        for (int off = MW; off < 64; off <<= 1) s += __shfl_xor(s, off);
        const float mu = s * (1.0f / E);
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) { float d = xv[k] - mu; v += d * d; }
#pragma unroll
        for (int off = MW; off < 64; off <<= 1) v += __shfl_xor(v, off);
        const float rstd = 1.0f / sqrtf(v * (1.0f / E) + LG_EPS);
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const int c = part * CPL + k;
            bufA[m * LDA + c] = (xv[k] - mu) * rstd * a.ln2g[c] + a.ln2b[c];
        }
    }
    __syncthreads();
    // ---- GEMM1 (K = E) + bias + GELU -> bufH, 64 output channels at a time
    for (int nc = 0; nc < N1; nc += 64) {
        f32x4 acc[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        wave_gemm<MT, 4, E>(acc, bufA, LDA, a.w1 + (size_t)nc * E);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = nc + nt * 16 + r;
                const float bias = a.b1[col];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = mt * 16 + 4 * g + v;
                    float h = acc[mt][nt][v] + bias;
                    if (a.a1s) {
                        float av, gv;
                        gelu_both_f(h, av, gv);
                        if (p0 + row < a.P) { HS<BF>::st1(a.a1s, (p0 + row) * N1 + col, av); HS<BF>::st1(a.g1s, (p0 + row) * N1 + col, gv); }
                        bufH[row * LDH + col] = av;
                    } else {
                        bufH[row * LDH + col] = gelu_f(h);
                    }
                }
            }
    }
    __syncthreads();
    // ---- GEMM2 (K = 4E) + bias -> h2
    for (int nc = 0; nc < N1; nc += 64) {
        f32x4 acc[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        wave_gemm<MT, 4, N1>(acc, bufH, LDH, a.w2 + (size_t)nc * N1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = nc + nt * 16 + r;
                const float bias = a.b2[col];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = mt * 16 + 4 * g + v;
                    if (p0 + row < a.P) HS<BF>::st1(a.h2, (p0 + row) * N1 + col, acc[mt][nt][v] + bias);
                }
            }
    }
}

template <int E, int MT>
static int launch_ffn1_t(const Ffn1Args& a, hipStream_t s) {
    ProfScope prof__(LG_K_FFN1, s);
    constexpr int N1 = 4 * E, MW = 16 * MT;
    size_t lds = (size_t)4 * MW * ((E + 4) + (N1 + 4)) * sizeof(float);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn1<E, MT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn1<E, MT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) { lg_set_error("ffn1: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    long per_wg = 4L * MW;
    int grid = (int)((a.P + per_wg - 1) / per_wg);
    if (a.hbf) k_ffn1<E, MT, true><<<grid, 256, lds, s>>>(a);
    else k_ffn1<E, MT, false><<<grid, 256, lds, s>>>(a);
    LG_CHECK_LAUNCH();
    return 0;
}

int launch_ffn1(int e, const Ffn1Args& a, hipStream_t s) {
    if (e == 16) return launch_ffn1_t<16, 2>(a, s);
    if (e == 32) return launch_ffn1_t<32, 2>(a, s);
    if (e == 64) return launch_ffn1_t<64, 1>(a, s);
    lg_set_error("ffn1: e=%d unsupported", e);
    return -1;
}

// ------------------------------------------------------------------------------------------------
// k_ffn2: workgroup = TY x TX pixel tile (M = TY*TX = 64*MT)
// ------------------------------------------------------------------------------------------------
template <int E, int MT, int TY, int TX, bool BF>
__global__ __launch_bounds__(256) void k_ffn2(Ffn2Args a, int tiles_x, int tiles_y) {
    constexpr int N1 = 4 * E, M = TY * TX, LDH = N1 + 4, LDO = E + 1, CQ = N1 / 4, NT3 = E / 16;
    static_assert(M == 64 * MT, "tile");
    extern __shared__ float smem[];
    float* bufH = smem;            // [M][LDH]  gelu(dw(h2))
    float* bufO = smem + M * LDH;  // [M][LDO]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    int t = blockIdx.x;
    const int tx_i = t % tiles_x;
    t /= tiles_x;
    const int ty_i = t % tiles_y;
    const long b = t / tiles_y;
    const int y0 = ty_i * TY, x0 = tx_i * TX;
    // ---- depthwise 3x3 (zero padding) + bias + GELU; thread <-> (pixel, channel quad), quad fixed per thread
    {
        const int q = threadIdx.x % CQ;
        float wq[4][9], bq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wq[u][k] = a.dww[(4 * q + u) * 9 + k];
            bq[u] = a.dwb[4 * q + u];
        }
        for (int m = threadIdx.x / CQ; m < M; m += 256 / CQ) {
            const int y = y0 + m / TX, x = x0 + m % TX;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool valid = (y < a.h) && (x < a.w);
            if (valid) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = y + dy - 1;
                    if (yy < 0 || yy >= a.h) continue;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int xx = x + dx - 1;
                        if (xx < 0 || xx >= a.w) continue;
                        const float4 v = HS<BF>::ld4(a.h2, ((b * a.h + yy) * (long)a.w + xx) * N1 + 4 * q);
                        acc.x += wq[0][dy * 3 + dx] * v.x;
                        acc.y += wq[1][dy * 3 + dx] * v.y;
                        acc.z += wq[2][dy * 3 + dx] * v.z;
                        acc.w += wq[3][dy * 3 + dx] * v.w;
                    }
                }
                acc.x += bq[0]; acc.y += bq[1]; acc.z += bq[2]; acc.w += bq[3];
                if (a.a3s) {
                    float4 av, gv;
                    gelu_both_f(acc.x, av.x, gv.x); gelu_both_f(acc.y, av.y, gv.y);
                    gelu_both_f(acc.z, av.z, gv.z); gelu_both_f(acc.w, av.w, gv.w);
                    const long o = ((b * a.h + y) * (long)a.w + x) * N1 + 4 * q;
                    HS<BF>::st4(a.a3s, o, av);
                    HS<BF>::st4(a.g3s, o, gv);
                    acc = av;
                } else {
                    acc = make_float4(gelu_f(acc.x), gelu_f(acc.y), gelu_f(acc.z), gelu_f(acc.w));
                }
            }
            *reinterpret_cast<float4*>(bufH + m * LDH + 4 * q) = acc;
        }
    }
    __syncthreads();
    // ---- GEMM3 (K = 4E, N = E): wave owns rows [wave*16*MT, +16*MT)
    {
        f32x4 acc[MT][NT3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT3; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        wave_gemm<MT, NT3, N1>(acc, bufH + wave * 16 * MT * LDH, LDH, a.w3);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT3; ++nt) {
                const int col = nt * 16 + r;
                const float bias = a.b3[col];
#pragma unroll
                for (int v = 0; v < 4; ++v) bufO[(wave * 16 * MT + mt * 16 + 4 * g + v) * LDO + col] = acc[mt][nt][v] + bias;
            }
    }
    __syncthreads();
    // ---- residual, store, emit LN-ed global half for the next block
    for (int m = threadIdx.x; m < M; m += 256) {
        const int y = y0 + m / TX, x = x0 + m % TX;
        if (y >= a.h || x >= a.w) continue;
        const long p = (b * a.h + y) * (long)a.w + x;
        float o[E];
        const float4* xs = reinterpret_cast<const float4*>(a.x + p * E);
        float4* yo = reinterpret_cast<float4*>(a.y + p * E);
#pragma unroll
        for (int k = 0; k < E / 4; ++k) {
            float4 xr = xs[k];
            o[4 * k] = xr.x + bufO[m * LDO + 4 * k];
            o[4 * k + 1] = xr.y + bufO[m * LDO + 4 * k + 1];
            o[4 * k + 2] = xr.z + bufO[m * LDO + 4 * k + 2];
            o[4 * k + 3] = xr.w + bufO[m * LDO + 4 * k + 3];
            yo[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
        }
        if (a.g) {
            float mu, rstd;
            ln_stats<E>(o, mu, rstd);
            const long hw = (long)a.h * a.w, s = (long)y * a.w + x;
#pragma unroll
            for (int n = E / 2; n < E; ++n) a.g[(b * (E / 2) + (n - E / 2)) * hw + s] = (o[n] - mu) * rstd * a.n1g[n] + a.n1b[n];
        }
    }
}

template <int E, int MT, int TY, int TX>
static int launch_ffn2_t(const Ffn2Args& a, hipStream_t s) {
    ProfScope prof__(LG_K_FFN2, s);
    constexpr int N1 = 4 * E, M = TY * TX;
    size_t lds = (size_t)(M * (N1 + 4) + M * (E + 1)) * sizeof(float);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn2<E, MT, TY, TX, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn2<E, MT, TY, TX, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) { lg_set_error("ffn2: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    int tiles_x = (a.w + TX - 1) / TX, tiles_y = (a.h + TY - 1) / TY;
    int grid = a.B * tiles_x * tiles_y;
    if (a.hbf) k_ffn2<E, MT, TY, TX, true><<<grid, 256, lds, s>>>(a, tiles_x, tiles_y);
    else k_ffn2<E, MT, TY, TX, false><<<grid, 256, lds, s>>>(a, tiles_x, tiles_y);
    LG_CHECK_LAUNCH();
    return 0;
}

int launch_ffn2(int e, const Ffn2Args& a, hipStream_t s) {
    if (e == 16) return launch_ffn2_t<16, 2, 8, 16>(a, s);
    if (e == 32) return launch_ffn2_t<32, 2, 8, 16>(a, s);
    if (e == 64) return launch_ffn2_t<64, 1, 8, 8>(a, s);
    lg_set_error("ffn2: e=%d unsupported", e);
    return -1;
}

// ------------------------------------------------------------------------------------------------
// k_ffn_fused: the whole feed_forward half-block in one kernel.  A workgroup owns an 8x16 pixel tile; the two
// channel-mixing GEMMs are recomputed on the 10x18 halo tile (x1.41 of their flops) so that h2 -- the only tensor with
// spatial coupling (depthwise 3x3) -- never leaves LDS.  HBM traffic drops from {x, h2 write, h2 read x taps, x, y}
// to the algorithmic {x (+halo), y}.  LDS: LN(x) halo tile [192][e+4], h2 halo tile [180][4e+4], one [16][4e+4]
// scratch per wave (a1 chunk, later gelu(dw(h2)) chunk): 81.7 KB at e=16 -> two workgroups per CU.
// Optionally saves h1/h2/h3 of the inner pixels for the backward (live stage).
// ------------------------------------------------------------------------------------------------
struct FfnFusedArgs {
    Ffn1Args a1;
    Ffn2Args a2;
};

template <int E, bool SAVE, bool BF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(E == 16 ? 2 : 1))) void k_ffn_fused(Ffn1Args a1, Ffn2Args a2, int tiles_x, int tiles_y, int ntiles) {
    constexpr int N1 = 4 * E, TY = 8, TX = 16, HX = TX + 2, HY = TY + 2, NH = HX * HY /*180*/, MH = 192, M = TY * TX;
    // LDA: at E = 16 the unpadded 64-byte row makes the A-fragment float4 reads one contiguous 1 KB (conflict-free) AND brings
    // the workgroup under 80 KB of LDS, so two workgroups share a CU and one's MFMA phases overlap the other's GELU/LDS phases
    constexpr int LDA = (E == 16 ? E : E + 4), LDH = N1 + 4, LDO = E + 1, CQ = N1 / 4, NT3 = E / 16;
    static_assert(CQ <= 64 && 64 % CQ == 0, "quad mapping");
    extern __shared__ float smem[];
    float* bufA = smem;                       // [MH][LDA]   LN2(x) on the halo tile; later [M][LDO] output tile
    float* bufH2 = smem + MH * LDA;           // [NH][LDH]   h2 on the halo tile (0 outside the image)
    float* scr = bufH2 + NH * LDH;            // [4][16][LDH] per-wave chunk
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    float* my = scr + wave * 16 * LDH;
    const int h = a2.h, w = a2.w;
    // small parameters of the half-block, staged once per (persistent) workgroup: as global loads inside the tile loop every one
    // of them was a dependent L1/L2 round trip on the critical path (~100 per tile).  LN / depthwise / b3 go to LDS (broadcast
    // reads), the two wide biases this lane needs in the MFMA epilogues to registers.
    __shared__ __attribute__((aligned(16))) float sPar[5 * E + 10 * N1];
    float* sLn2g = sPar;            float* sLn2b = sPar + E;
    float* sN1g = sPar + 2 * E;     float* sN1b = sPar + 3 * E;
    float* sB3 = sPar + 4 * E;      float* sDww = sPar + 5 * E;     // [N1][9]
    float* sDwb = sDww + 9 * N1;                                    // [N1]
    for (int i = threadIdx.x; i < E; i += 256) {
        sLn2g[i] = a1.ln2g[i]; sLn2b[i] = a1.ln2b[i]; sB3[i] = a2.b3[i];
        sN1g[i] = a2.g ? a2.n1g[i] : 0.f; sN1b[i] = a2.g ? a2.n1b[i] : 0.f;
    }
    for (int i = threadIdx.x; i < 9 * N1; i += 256) sDww[i] = a2.dww[i];
    for (int i = threadIdx.x; i < N1; i += 256) sDwb[i] = a2.dwb[i];
    float b1r[4], b2r[4];   // biases of the h1 / h2 columns this lane holds in the MFMA C layout
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = (E == 32 ? wave * 32 : 0) + nt * 16 + r;
        b1r[nt] = (nt < (E == 32 ? 2 : 4)) ? a1.b1[col] : 0.f;
        b2r[nt] = (nt < (E == 32 ? 2 : 4)) ? a1.b2[col] : 0.f;
    }
    // weights as MFMA B fragments, resident in registers for ALL tiles of this (persistent) workgroup when they fit
    // (e = 16: 16 + 64 + 16 VGPRs); otherwise every row chunk re-reads them through L1
    constexpr bool RB = (E == 16);
    float4 w1f[RB ? 4 : 1][1], w2f[RB ? 4 : 1][RB ? 4 : 1];
    if (RB) {
        load_bfrag<4, 1>(reinterpret_cast<float4(&)[4][1]>(w1f), a1.w1, E);
        load_bfrag<4, 4>(reinterpret_cast<float4(&)[4][4]>(w2f), a1.w2, N1);
    }
    // e = 32: the weights do not fit one wave's registers (W2 alone is 64 KB), so P1 splits the OUTPUT COLUMNS across the four
    // waves instead of the rows: wave w owns columns [32w, 32w + 32) of h1 and h2 for all 192 halo rows and keeps exactly its
    // slices of W1 (16 VGPRs) and W2 (64 VGPRs) resident for the whole kernel -- no weight traffic in the tile loop
    constexpr bool NS = (E == 32);
    float4 w1p[NS ? 2 : 1][NS ? 2 : 1], w2p[NS ? 2 : 1][NS ? 8 : 1];
    if (NS) {
        load_bfrag<2, 2>(reinterpret_cast<float4(&)[2][2]>(w1p), a1.w1 + (size_t)(wave * 32) * E, E);
        load_bfrag<2, 8>(reinterpret_cast<float4(&)[2][8]>(w2p), a1.w2 + (size_t)(wave * 32) * N1, N1);
    }
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int tx_i = t % tiles_x;
    t /= tiles_x;
    const int ty_i = t % tiles_y;
    const long b = t / tiles_y;
    const int y0 = ty_i * TY, x0 = tx_i * TX;
    __syncthreads();   // previous tile's readers of bufA (output tile) / bufH2 are done
    // ---- P0: halo tile load + LayerNorm (one pixel per thread, 192 rows; rows >= 180 and out-of-image pixels are zero)
    if (threadIdx.x < MH) {
        const int m = threadIdx.x;
        const int hy = m / HX, hx = m - hy * HX;
        const int y = y0 + hy - 1, x = x0 + hx - 1;
        float xv[E];
        const bool in = (m < NH) && y >= 0 && y < h && x >= 0 && x < w;
        if (in) {
            const float4* src = reinterpret_cast<const float4*>(a1.x + ((b * h + y) * (long)w + x) * E);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) {
                float4 v = src[k];
                xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
            }
            float mu, rstd;
            ln_stats<E>(xv, mu, rstd);
#pragma unroll
            for (int c = 0; c < E; ++c) xv[c] = (xv[c] - mu) * rstd * sLn2g[c] + sLn2b[c];
        } else {
#pragma unroll
            for (int c = 0; c < E; ++c) xv[c] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < E / 4; ++k)
            *reinterpret_cast<float4*>(bufA + m * LDA + 4 * k) = make_float4(xv[4 * k], xv[4 * k + 1], xv[4 * k + 2], xv[4 * k + 3]);
    }
    __syncthreads();
    // ---- P1: per wave, 3 chunks of 16 halo pixels: GEMM1 -> GELU -> GEMM2 -> h2 tile
    if constexpr (NS) {
        // four row blocks of 48 halo rows; per block: GEMM1 + GELU -> A2 (gelu(h1), shared by the waves) | barrier | GEMM2 -> h2 tile
        float* A2 = scr;   // [48][LDH]
        const int cw0 = wave * 32;
        for (int rb = 0; rb < 4; ++rb) {
            const int row0 = rb * 48;
            long prow[3][4];
            bool inner[3][4], inimg[3][4];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int m = row0 + mt * 16 + 4 * g + v;
                    const int hy = m / HX, hx = m - hy * HX;
                    const int y = y0 + hy - 1, x = x0 + hx - 1;
                    inimg[mt][v] = (m < NH) && y >= 0 && y < h && x >= 0 && x < w;
                    inner[mt][v] = inimg[mt][v] && hy >= 1 && hy <= TY && hx >= 1 && hx <= TX;
                    prow[mt][v] = (b * h + y) * (long)w + x;
                }
            f32x4 acc[3][2];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            wave_gemm_rb<3, 2, 2>(acc, bufA + row0 * LDA, LDA, reinterpret_cast<const float4(&)[2][2]>(w1p));
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int col = cw0 + nt * 16 + r;
                const float bias = b1r[nt];
#pragma unroll
                for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                    for (int v = 0; v < 4; v += 2) {
                        const lg_v2f hh = (lg_v2f){acc[mt][nt][v] + bias, acc[mt][nt][v + 1] + bias};
                        lg_v2f av;
                        if (SAVE) {
                            lg_v2f gv;
                            gelu2_both_f(hh, av, gv);
                            if (inner[mt][v]) { HS<BF>::st1(a1.a1s, prow[mt][v] * N1 + col, av.x); HS<BF>::st1(a1.g1s, prow[mt][v] * N1 + col, gv.x); }
                            if (inner[mt][v + 1]) { HS<BF>::st1(a1.a1s, prow[mt][v + 1] * N1 + col, av.y); HS<BF>::st1(a1.g1s, prow[mt][v + 1] * N1 + col, gv.y); }
                        } else {
                            av = gelu2_f(hh);
                        }
                        A2[(mt * 16 + 4 * g + v) * LDH + col] = av.x;
                        A2[(mt * 16 + 4 * g + v + 1) * LDH + col] = av.y;
                    }
            }
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            wave_gemm_rb<3, 2, 8>(acc, A2, LDH, reinterpret_cast<const float4(&)[2][8]>(w2p));
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int col = cw0 + nt * 16 + r;
                const float bias = b2r[nt];
#pragma unroll
                for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int m = row0 + mt * 16 + 4 * g + v;
                        const float hh = inimg[mt][v] ? acc[mt][nt][v] + bias : 0.f;   // dep_conv zero-pads h2
                        if (SAVE && inner[mt][v]) HS<BF>::st1(a1.h2, prow[mt][v] * N1 + col, hh);
                        if (m < NH) bufH2[m * LDH + col] = hh;
                    }
            }
            __syncthreads();   // A2 is rewritten by the next row block
        }
    } else {
    static_assert(NS || N1 == 64, "register-resident biases assume one 64-column block");
    for (int ch = 0; ch < 3; ++ch) {
        const int row0 = (wave * 3 + ch) * 16;
        // validity / global pixel index of the 4 rows this lane owns in the C layout
        long prow[4];
        bool inner[4], inimg[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int m = row0 + 4 * g + v;
            const int hy = m / HX, hx = m - hy * HX;
            const int y = y0 + hy - 1, x = x0 + hx - 1;
            inimg[v] = (m < NH) && y >= 0 && y < h && x >= 0 && x < w;
            inner[v] = inimg[v] && hy >= 1 && hy <= TY && hx >= 1 && hx <= TX;
            prow[v] = (b * h + y) * (long)w + x;
        }
        for (int nc = 0; nc < N1; nc += 64) {
            f32x4 acc[1][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (RB) wave_gemm_rb<1, 4, 1>(acc, bufA + row0 * LDA, LDA, reinterpret_cast<const float4(&)[4][1]>(w1f));
            else wave_gemm<1, 4, E>(acc, bufA + row0 * LDA, LDA, a1.w1 + (size_t)nc * E);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = nc + nt * 16 + r;
                const float bias = b1r[nt];
#pragma unroll
                for (int v = 0; v < 4; v += 2) {   // packed pairs (v_pk_fma_f32)
                    const lg_v2f hh = (lg_v2f){acc[0][nt][v] + bias, acc[0][nt][v + 1] + bias};
                    lg_v2f av;
                    if (SAVE) {
                        lg_v2f gv;
                        gelu2_both_f(hh, av, gv);
                        if (inner[v]) { HS<BF>::st1(a1.a1s, prow[v] * N1 + col, av.x); HS<BF>::st1(a1.g1s, prow[v] * N1 + col, gv.x); }
                        if (inner[v + 1]) { HS<BF>::st1(a1.a1s, prow[v + 1] * N1 + col, av.y); HS<BF>::st1(a1.g1s, prow[v + 1] * N1 + col, gv.y); }
                    } else {
                        av = gelu2_f(hh);
                    }
                    my[(4 * g + v) * LDH + col] = av.x;
                    my[(4 * g + v + 1) * LDH + col] = av.y;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int nc = 0; nc < N1; nc += 64) {
            f32x4 acc[1][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (RB) wave_gemm_rb<1, 4, 4>(acc, my, LDH, reinterpret_cast<const float4(&)[4][4]>(w2f));
            else wave_gemm<1, 4, N1>(acc, my, LDH, a1.w2 + (size_t)nc * N1);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = nc + nt * 16 + r;
                const float bias = b2r[nt];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int m = row0 + 4 * g + v;
                    const float hh = inimg[v] ? acc[0][nt][v] + bias : 0.f;   // dep_conv zero-pads h2 (basic_module_unformer_v2.py:18)
                    if (SAVE && inner[v]) HS<BF>::st1(a1.h2, prow[v] * N1 + col, hh);
                    if (m < NH) bufH2[m * LDH + col] = hh;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    }
    __syncthreads();
    // ---- P2: per wave, 2 chunks of 16 inner pixels: dw3x3 + GELU -> scratch -> GEMM3 -> output tile (in bufA's space)
    float* bufO = bufA;
    // the residual rows of P3 are requested now: their HBM round trip hides under the depthwise phase (e = 16 only: registers)
    constexpr bool XPRE = (E == 16);
    float4 xres[XPRE ? E / 4 : 1];
    if (XPRE && threadIdx.x < M) {
        const int ym = y0 + threadIdx.x / TX, xm = x0 + threadIdx.x % TX;
        if (ym < h && xm < w) {
            const float4* xs0 = reinterpret_cast<const float4*>(a2.x + ((b * h + ym) * (long)w + xm) * E);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) xres[k] = xs0[k];
        }
    }
    {
        // depthwise taps and the W3 fragments are (re)loaded per tile: keeping them live across P1 costs ~56 VGPRs
        const int q = lane % CQ;
        float wq[4][9], bq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wq[u][k] = sDww[(4 * q + u) * 9 + k];
            bq[u] = sDwb[4 * q + u];
        }
        float4 w3f[1][RB ? 4 : 1];
        if (RB) load_bfrag<1, 4>(reinterpret_cast<float4(&)[1][4]>(w3f), a2.w3, N1);
        for (int ch = 0; ch < 2; ++ch) {
            const int m0 = (wave * 2 + ch) * 16;
            for (int mm = lane / CQ; mm < 16; mm += 64 / CQ) {
                const int m = m0 + mm;
                const int ty = m / TX, tx = m - ty * TX;
                float4 acc = make_float4(bq[0], bq[1], bq[2], bq[3]);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float4 v = *reinterpret_cast<const float4*>(bufH2 + ((ty + dy) * HX + tx + dx) * LDH + 4 * q);
                        acc.x += wq[0][dy * 3 + dx] * v.x; acc.y += wq[1][dy * 3 + dx] * v.y;
                        acc.z += wq[2][dy * 3 + dx] * v.z; acc.w += wq[3][dy * 3 + dx] * v.w;
                    }
                const int y = y0 + ty, x = x0 + tx;
                float4 av;
                if (SAVE) {
                    lg_v2f a01, a23, g01, g23;
                    gelu2_both_f((lg_v2f){acc.x, acc.y}, a01, g01);
                    gelu2_both_f((lg_v2f){acc.z, acc.w}, a23, g23);
                    av = make_float4(a01.x, a01.y, a23.x, a23.y);
                    if (y < h && x < w) {
                        const long o = ((b * h + y) * (long)w + x) * N1 + 4 * q;
                        HS<BF>::st4(a2.a3s, o, av);
                        HS<BF>::st4(a2.g3s, o, make_float4(g01.x, g01.y, g23.x, g23.y));
                    }
                } else {
                    const lg_v2f a01 = gelu2_f((lg_v2f){acc.x, acc.y}), a23 = gelu2_f((lg_v2f){acc.z, acc.w});
                    av = make_float4(a01.x, a01.y, a23.x, a23.y);
                }
                *reinterpret_cast<float4*>(my + mm * LDH + 4 * q) = av;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            f32x4 acc3[1][NT3];
#pragma unroll
            for (int nt = 0; nt < NT3; ++nt) acc3[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (RB) wave_gemm_rb<1, 1, 4>(reinterpret_cast<f32x4(&)[1][1]>(acc3), my, LDH, reinterpret_cast<const float4(&)[1][4]>(w3f));
            else wave_gemm<1, NT3, N1>(acc3, my, LDH, a2.w3);
#pragma unroll
            for (int nt = 0; nt < NT3; ++nt) {
                const int col = nt * 16 + r;
                const float bias = sB3[col];
#pragma unroll
                for (int v = 0; v < 4; ++v) bufO[(m0 + 4 * g + v) * LDO + col] = acc3[0][nt][v] + bias;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    // ---- P3: residual, store, planar LN1 half for the next block
    if (threadIdx.x < M) {
        const int m = threadIdx.x;
        const int y = y0 + m / TX, x = x0 + m % TX;
        if (y < h && x < w) {
            const long p = (b * h + y) * (long)w + x;
            float o[E];
            const float4* xs = reinterpret_cast<const float4*>(a2.x + p * E);
            float4* yo = reinterpret_cast<float4*>(a2.y + p * E);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) {
                float4 xr = XPRE ? xres[XPRE ? k : 0] : xs[k];
                o[4 * k] = xr.x + bufO[m * LDO + 4 * k];
                o[4 * k + 1] = xr.y + bufO[m * LDO + 4 * k + 1];
                o[4 * k + 2] = xr.z + bufO[m * LDO + 4 * k + 2];
                o[4 * k + 3] = xr.w + bufO[m * LDO + 4 * k + 3];
                yo[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
            }
            if (a2.g) {
                float mu, rstd;
                ln_stats<E>(o, mu, rstd);
                const long hw = (long)h * w, s = (long)y * w + x;
#pragma unroll
                for (int n = E / 2; n < E; ++n) a2.g[(b * (E / 2) + (n - E / 2)) * hw + s] = (o[n] - mu) * rstd * sN1g[n] + sN1b[n];
            }
        }
    }
    }   // tiles of this workgroup
}

// ------------------------------------------------------------------------------------------------
// k_ffn_strip (e = 16, fp32): the fused feed_forward half-block walking DOWN a 16-column strip.  k_ffn_fused recomputes h1 / h2
// on the full 10x18 halo of every 8x16 tile (192 MFMA rows per 128 output pixels); here consecutive tiles of a strip share
// their halo rows through a 10-row LDS ring of h2, so a step only computes the 8 NEW halo rows (8 x 18 = 144 pixels = exactly
// 9 MFMA row blocks).  144 rows do not split evenly over 4 waves by rows, so the OUTPUT COLUMNS are split instead (as in the
// e = 32 path): wave w owns columns [16 w, 16 w + 16) of h1 and h2 for all rows and keeps its W1 / W2 slices in 5 VGPR quads.
// GEMM1/GEMM2 MFMAs per 128 pixels: 720 instead of 960; GELU-1 evaluations 9 216 instead of 12 288.
// LDS: {LN(x) rows [144][16] + gelu(h1) chunk [48][68]} aliased with {per-wave chunk [4][16][68] + output tile [128][17]}
// = 26 KB, h2 ring [10][18][68] = 49 KB -> 75 KB, two workgroups per CU.
// ------------------------------------------------------------------------------------------------
template <bool SAVE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_ffn_strip(Ffn1Args a1, Ffn2Args a2, int tiles_x, int strips_y, int nstrips,
                                                                                          int SH) {
    constexpr int E = 16, N1 = 64, TX = 16, HX = 18, TY = 8, RING = 10, LDA = 16, LDH = 68, LDO = 17, CQ = 16, M = 128;
    constexpr int R1 = 4 * 16 * LDH + M * LDO;   // 6528 floats >= 144 * LDA + 48 * LDH
    extern __shared__ float smem[];
    float* bufA = smem;                  // [144][LDA]  LN2(x) of the new halo rows
    float* A2 = smem + 144 * LDA;        // [48][LDH]   gelu(h1) chunk, shared by the waves
    float* scr = smem;                   // P2: [4][16][LDH] per-wave gelu(dw(h2)) chunk
    float* bufO = smem + 4 * 16 * LDH;   // P2/P3: [M][LDO] output tile
    float* ring = smem + R1;             // [RING][HX][LDH] h2 rows y, slot (y - (Y0 - 1)) % RING
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    float* my = scr + wave * 16 * LDH;
    const int h = a2.h, w = a2.w;
    __shared__ __attribute__((aligned(16))) float sPar[5 * E];
    float* sLn2g = sPar;            float* sLn2b = sPar + E;
    float* sN1g = sPar + 2 * E;     float* sN1b = sPar + 3 * E;
    float* sB3 = sPar + 4 * E;
    for (int i = threadIdx.x; i < E; i += 256) {
        sLn2g[i] = a1.ln2g[i]; sLn2b[i] = a1.ln2b[i]; sB3[i] = a2.b3[i];
        sN1g[i] = a2.g ? a2.n1g[i] : 0.f; sN1b[i] = a2.g ? a2.n1b[i] : 0.f;
    }
    __shared__ __attribute__((aligned(16))) float sMask[144];   // 1 for halo pixels inside the image (h2 is zero-padded outside)
    const int col = wave * 16 + r;                 // the h1 / h2 column this lane holds in the MFMA C layout
    const float b1c = a1.b1[col], b2c = a1.b2[col];
    float4 w1p[1][1], w2p[1][4];
    load_bfrag<1, 1>(w1p, a1.w1 + (size_t)(wave * 16) * E, E);
    load_bfrag<1, 4>(w2p, a1.w2 + (size_t)(wave * 16) * N1, N1);
    // the column split leaves P1 with 20 weight registers (the row-split tile kernel holds 80), so the depthwise taps and the W3
    // fragments stay resident too: no weight load -- LDS or global -- inside a step
    const int q = lane % CQ;
    float wq[4][9], bq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) wq[u][kk] = a2.dww[(4 * q + u) * 9 + kk];
        bq[u] = a2.dwb[4 * q + u];
    }
    float4 w3f[1][4];
    load_bfrag<1, 4>(w3f, a2.w3, N1);
#pragma unroll 1
    for (int strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
    int t = strip;
    const int tx_i = t % tiles_x;
    t /= tiles_x;
    const int sy = t % strips_y;
    const long b = t / strips_y;
    const int x0 = tx_i * TX, Y0 = sy * SH, Yend = min(Y0 + SH, h);

    // h2 of halo rows [ya, ya + nr) x columns [x0 - 1, x0 + 17) -> ring   (nr = 2: strip prologue, 8: one step)
    auto compute_rows = [&](int ya, int nr) {
        const int npx = nr * HX, nchunks = (npx + 47) / 48;
        __syncthreads();   // readers of bufA / A2 space (previous step's output tile) are done
        if (threadIdx.x < nchunks * 48) {
            const int m = threadIdx.x;
            const int hy = m / HX, hx = m - hy * HX;
            const int y = ya + hy, x = x0 + hx - 1;
            float xv[E];
            const bool in = (m < npx) && y >= 0 && y < h && x >= 0 && x < w;
            sMask[m] = in ? 1.0f : 0.0f;
            if (in) {
                const float4* src = reinterpret_cast<const float4*>(a1.x + ((b * h + y) * (long)w + x) * E);
#pragma unroll
                for (int k = 0; k < E / 4; ++k) {
                    float4 v = src[k];
                    xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
                }
                float mu, rstd;
                ln_stats<E>(xv, mu, rstd);
#pragma unroll
                for (int c = 0; c < E; ++c) xv[c] = (xv[c] - mu) * rstd * sLn2g[c] + sLn2b[c];
            } else {
#pragma unroll
                for (int c = 0; c < E; ++c) xv[c] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < E / 4; ++k)
                *reinterpret_cast<float4*>(bufA + m * LDA + 4 * k) = make_float4(xv[4 * k], xv[4 * k + 1], xv[4 * k + 2], xv[4 * k + 3]);
        }
        __syncthreads();
        // ring pixel of halo pixel m: ((s0 + hy) % RING) * HX + hx = (s0 * HX + m) mod (RING * HX) -- no division per pixel
        const int ring0 = ((ya - Y0 + 1) % RING) * HX;
        for (int ch = 0; ch < nchunks; ++ch) {
            const int row0 = ch * 48;
            long prow[3][4];
            int rslot[3][4];
            bool inner[3][4];
            float mk[3][4];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                const float4 m4 = *reinterpret_cast<const float4*>(sMask + row0 + mt * 16 + 4 * g);
                mk[mt][0] = m4.x; mk[mt][1] = m4.y; mk[mt][2] = m4.z; mk[mt][3] = m4.w;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int m = row0 + mt * 16 + 4 * g + v;
                    int rs = ring0 + m;
                    rs = rs >= RING * HX ? rs - RING * HX : rs;
                    rslot[mt][v] = (m < npx) ? rs : -1;
                    if (SAVE) {
                        const int hy = m / HX, hx = m - hy * HX;
                        const int y = ya + hy, x = x0 + hx - 1;
                        inner[mt][v] = mk[mt][v] != 0.f && hx >= 1 && hx <= TX && y >= Y0 && y < Yend;
                        prow[mt][v] = (b * h + y) * (long)w + x;
                    }
                }
            }
            f32x4 acc[3][1];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) acc[mt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            wave_gemm_rb<3, 1, 1>(acc, bufA + row0 * LDA, LDA, w1p);
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int v = 0; v < 4; v += 2) {
                    const lg_v2f hh = (lg_v2f){acc[mt][0][v] + b1c, acc[mt][0][v + 1] + b1c};
                    lg_v2f av;
                    if (SAVE) {
                        lg_v2f gv;
                        gelu2_both_f(hh, av, gv);
                        if (inner[mt][v]) { HS<false>::st1(a1.a1s, prow[mt][v] * N1 + col, av.x); HS<false>::st1(a1.g1s, prow[mt][v] * N1 + col, gv.x); }
                        if (inner[mt][v + 1]) { HS<false>::st1(a1.a1s, prow[mt][v + 1] * N1 + col, av.y); HS<false>::st1(a1.g1s, prow[mt][v + 1] * N1 + col, gv.y); }
                    } else {
                        av = gelu2_f(hh);
                    }
                    A2[(mt * 16 + 4 * g + v) * LDH + col] = av.x;
                    A2[(mt * 16 + 4 * g + v + 1) * LDH + col] = av.y;
                }
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) acc[mt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            wave_gemm_rb<3, 1, 4>(acc, A2, LDH, w2p);
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float hh = (acc[mt][0][v] + b2c) * mk[mt][v];   // dep_conv zero-pads h2 (basic_module_unformer_v2.py:18)
                    if (SAVE && inner[mt][v]) HS<false>::st1(a1.h2, prow[mt][v] * N1 + col, hh);
                    if (rslot[mt][v] >= 0) ring[rslot[mt][v] * LDH + col] = hh;
                }
            __syncthreads();   // A2 is rewritten by the next chunk; the ring rows are complete after the last one
        }
    };

    compute_rows(Y0 - 1, 2);
#pragma unroll 1
    for (int y0 = Y0; y0 < Yend; y0 += TY) {
    // the residual rows of P3 are requested first: their HBM round trip hides under the whole step
    float4 xres[E / 4];
    if (threadIdx.x < M) {
        const int ym = y0 + threadIdx.x / TX, xm = x0 + threadIdx.x % TX;
        if (ym < h && xm < w) {
            const float4* xs0 = reinterpret_cast<const float4*>(a2.x + ((b * h + ym) * (long)w + xm) * E);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) xres[k] = xs0[k];
        }
    }
    compute_rows(y0 + 1, TY);
    // ---- P2: per wave, 2 chunks of 16 inner pixels: dw3x3 over the ring + GELU -> scratch -> GEMM3 -> output tile
    {
        const int sbase = (y0 - Y0) % RING;            // ring slot of row y0 - 1
        for (int ch = 0; ch < 2; ++ch) {
            const int m0 = (wave * 2 + ch) * 16;
            for (int mm = lane / CQ; mm < 16; mm += 64 / CQ) {
                const int m = m0 + mm;
                const int ty = m / TX, tx = m - ty * TX;
                float4 acc = make_float4(bq[0], bq[1], bq[2], bq[3]);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    int sl = sbase + ty + dy;
                    sl = sl >= RING ? sl - RING : sl;
                    sl = sl >= RING ? sl - RING : sl;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float4 v = *reinterpret_cast<const float4*>(ring + (sl * HX + tx + dx) * LDH + 4 * q);
                        acc.x += wq[0][dy * 3 + dx] * v.x; acc.y += wq[1][dy * 3 + dx] * v.y;
                        acc.z += wq[2][dy * 3 + dx] * v.z; acc.w += wq[3][dy * 3 + dx] * v.w;
                    }
                }
                const int y = y0 + ty, x = x0 + tx;
                float4 av;
                if (SAVE) {
                    lg_v2f a01, a23, g01, g23;
                    gelu2_both_f((lg_v2f){acc.x, acc.y}, a01, g01);
                    gelu2_both_f((lg_v2f){acc.z, acc.w}, a23, g23);
                    av = make_float4(a01.x, a01.y, a23.x, a23.y);
                    if (y < Yend && x < w) {
                        const long o = ((b * h + y) * (long)w + x) * N1 + 4 * q;
                        HS<false>::st4(a2.a3s, o, av);
                        HS<false>::st4(a2.g3s, o, make_float4(g01.x, g01.y, g23.x, g23.y));
                    }
                } else {
                    const lg_v2f a01 = gelu2_f((lg_v2f){acc.x, acc.y}), a23 = gelu2_f((lg_v2f){acc.z, acc.w});
                    av = make_float4(a01.x, a01.y, a23.x, a23.y);
                }
                *reinterpret_cast<float4*>(my + mm * LDH + 4 * q) = av;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            f32x4 acc3[1][1];
            acc3[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            wave_gemm_rb<1, 1, 4>(acc3, my, LDH, w3f);
            const float bias = sB3[r];
#pragma unroll
            for (int v = 0; v < 4; ++v) bufO[(m0 + 4 * g + v) * LDO + r] = acc3[0][0][v] + bias;
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    // ---- P3: residual, store, planar LN1 half for the next block
    if (threadIdx.x < M) {
        const int m = threadIdx.x;
        const int y = y0 + m / TX, x = x0 + m % TX;
        if (y < Yend && x < w) {
            const long p = (b * h + y) * (long)w + x;
            float o[E];
            float4* yo = reinterpret_cast<float4*>(a2.y + p * E);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) {
                const float4 xr = xres[k];
                o[4 * k] = xr.x + bufO[m * LDO + 4 * k];
                o[4 * k + 1] = xr.y + bufO[m * LDO + 4 * k + 1];
                o[4 * k + 2] = xr.z + bufO[m * LDO + 4 * k + 2];
                o[4 * k + 3] = xr.w + bufO[m * LDO + 4 * k + 3];
                yo[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
            }
            if (a2.g) {
                float mu, rstd;
                ln_stats<E>(o, mu, rstd);
                const long hw = (long)h * w, sp = (long)y * w + x;
#pragma unroll
                for (int n = E / 2; n < E; ++n) a2.g[(b * (E / 2) + (n - E / 2)) * hw + sp] = (o[n] - mu) * rstd * sN1g[n] + sN1b[n];
            }
        }
    }
    }   // steps of the strip
    }   // strips of this workgroup
}

static int launch_ffn_strip(const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s) {
    ProfScope prof__(LG_K_FFN2, s);
    const size_t lds = (size_t)(4 * 16 * 68 + 128 * 17 + 10 * 18 * 68) * sizeof(float);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn_strip<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_strip<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) { lg_set_error("ffn_strip: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    const int tiles_x = (a2.w + 15) / 16;
    // strip height: the tallest multiple of 8 rows that still yields >= 512 strips (two resident workgroups per CU), at least 16
    int SH = (a2.h + 7) / 8 * 8;
    while (SH > 16 && (long)a2.B * tiles_x * ((a2.h + SH - 1) / SH) < 512) SH = (SH / 2 + 7) / 8 * 8;
    const int strips_y = (a2.h + SH - 1) / SH;
    const int nstrips = a2.B * tiles_x * strips_y;
    const int grid = nstrips < 512 ? nstrips : 512;
    if (a1.a1s != nullptr) k_ffn_strip<true><<<grid, 256, lds, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
    else k_ffn_strip<false><<<grid, 256, lds, s>>>(a1, a2, tiles_x, strips_y, nstrips, SH);
    LG_CHECK_LAUNCH();
    return 0;
}

template <int E>
static int launch_ffn_fused_t(const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s) {
    ProfScope prof__(LG_K_FFN2, s);
    constexpr int N1 = 4 * E;
    size_t lds = (size_t)(192 * (E == 16 ? E : E + 4) + 180 * (N1 + 4) + 4 * 16 * (N1 + 4)) * sizeof(float);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn_fused<E, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_fused<E, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_fused<E, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) { lg_set_error("ffn_fused: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    int tiles_x = (a2.w + 15) / 16, tiles_y = (a2.h + 7) / 8;
    const int ntiles = a2.B * tiles_x * tiles_y;
    const int cap = (E == 32) ? 256 : 512;   // persistent: resident workgroups per CU (2 at e = 16, 1 at e = 32) walk the tiles, weights stay in registers
    const int grid = ntiles < cap ? ntiles : cap;
    const bool save = a1.a1s != nullptr;   // forward of the live stage: keep gelu / gelu' / h2 for the backward
    if (save && a1.hbf) k_ffn_fused<E, true, true><<<grid, 256, lds, s>>>(a1, a2, tiles_x, tiles_y, ntiles);
    else if (save) k_ffn_fused<E, true, false><<<grid, 256, lds, s>>>(a1, a2, tiles_x, tiles_y, ntiles);
    else k_ffn_fused<E, false, false><<<grid, 256, lds, s>>>(a1, a2, tiles_x, tiles_y, ntiles);   // nothing stored: storage type irrelevant
    LG_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------------------------------------
// k_ffn_fused_bf: throughput-mode variant of k_ffn_fused (lg_config.precision = 1): the three 1x1-conv GEMMs run on the
// bf16 matrix cores (v_mfma_f32_16x16x32_bf16 / 16x16x16 for K = 16) with fp32 accumulation; their LDS operand tiles
// (LN(x), gelu(h1) chunk, gelu(dw(h2)) chunk) are bf16; LayerNorm, bias, GELU, the h2 tile, the depthwise 3x3, the
// residual and everything the backward saves are computed in fp32.  Weights stay fp32 in HBM and are converted to bf16
// B fragments once per (persistent) workgroup.  Same tiling / phases as the fp32 kernel.
// ------------------------------------------------------------------------------------------------
template <int E, bool SAVE, bool BF>
__global__ __launch_bounds__(256) void k_ffn_fused_bf(Ffn1Args a1, Ffn2Args a2, int tiles_x, int tiles_y, int ntiles) {
    constexpr int N1 = 4 * E, TY = 8, TX = 16, HX = TX + 2, HY = TY + 2, NH = HX * HY /*180*/, MH = 192, M = TY * TX;
    constexpr int LDA = E + 8, LDS16 = N1 + 8 /* halves */, LDH = N1 + 4, LDO = E + 1, CQ = N1 / 4, NT3 = E / 16;
    constexpr int KB2 = N1 / 32;   // 32-deep k blocks of GEMM2 / GEMM3
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bufH2 = smem;                                   // [NH][LDH] fp32  h2 on the halo tile (0 outside the image)
    float* bufO = bufH2 + NH * LDH;                        // [M][LDO]  fp32  output tile
    __bf16* bufA = reinterpret_cast<__bf16*>(bufO + M * LDO + 3);   // [MH][LDA] bf16 LN2(x) on the halo tile (16-byte aligned below)
    bufA = reinterpret_cast<__bf16*>((reinterpret_cast<uintptr_t>(bufA) + 15) & ~(uintptr_t)15);
    __bf16* scr = bufA + MH * LDA;                         // [4][16][LDS16] bf16 per-wave chunk
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    __bf16* my = scr + wave * 16 * LDS16;
    const int h = a2.h, w = a2.w;
    // small parameters staged once per workgroup (see k_ffn_fused); the wide biases stay global here (col varies per nc block)
    __shared__ __attribute__((aligned(16))) float sPar[5 * E + 10 * N1 + 2 * N1];
    float* sLn2g = sPar;            float* sLn2b = sPar + E;
    float* sN1g = sPar + 2 * E;     float* sN1b = sPar + 3 * E;
    float* sB3 = sPar + 4 * E;      float* sDww = sPar + 5 * E;     // [N1][9]
    float* sDwb = sDww + 9 * N1;    float* sB1 = sDwb + N1;         float* sB2 = sB1 + N1;
    for (int i = threadIdx.x; i < E; i += 256) {
        sLn2g[i] = a1.ln2g[i]; sLn2b[i] = a1.ln2b[i]; sB3[i] = a2.b3[i];
        sN1g[i] = a2.g ? a2.n1g[i] : 0.f; sN1b[i] = a2.g ? a2.n1b[i] : 0.f;
    }
    for (int i = threadIdx.x; i < 9 * N1; i += 256) sDww[i] = a2.dww[i];
    for (int i = threadIdx.x; i < N1; i += 256) { sDwb[i] = a2.dwb[i]; sB1[i] = a1.b1[i]; sB2[i] = a1.b2[i]; }
    // B fragments (bf16) for the whole life of the workgroup
    s16x4 w1k16[4];
    bf16x8 w2f[4][KB2], w3f[NT3][KB2];
    if constexpr (E == 16) {
        load_bfrag_bf16_k16<4>(w1k16, a1.w1);
        load_bfrag_bf16<4, KB2>(w2f, a1.w2, N1);
        load_bfrag_bf16<NT3, KB2>(w3f, a2.w3, N1);
    }
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int tx_i = t % tiles_x;
    t /= tiles_x;
    const int ty_i = t % tiles_y;
    const long b = t / tiles_y;
    const int y0 = ty_i * TY, x0 = tx_i * TX;
    __syncthreads();
    // ---- P0: halo tile load + LayerNorm (fp32) -> bf16 rows
    if (threadIdx.x < MH) {
        const int m = threadIdx.x;
        const int hy = m / HX, hx = m - hy * HX;
        const int y = y0 + hy - 1, x = x0 + hx - 1;
        float xv[E];
        const bool in = (m < NH) && y >= 0 && y < h && x >= 0 && x < w;
        if (in) {
            const float4* src = reinterpret_cast<const float4*>(a1.x + ((b * h + y) * (long)w + x) * E);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) {
                float4 v = src[k];
                xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
            }
            float mu, rstd;
            ln_stats<E>(xv, mu, rstd);
#pragma unroll
            for (int c = 0; c < E; ++c) xv[c] = (xv[c] - mu) * rstd * sLn2g[c] + sLn2b[c];
        } else {
#pragma unroll
            for (int c = 0; c < E; ++c) xv[c] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < E / 4; ++k) {
            const bf16x4 hv = (bf16x4){(__bf16)xv[4 * k], (__bf16)xv[4 * k + 1], (__bf16)xv[4 * k + 2], (__bf16)xv[4 * k + 3]};
            *reinterpret_cast<bf16x4*>(bufA + m * LDA + 4 * k) = hv;
        }
    }
    __syncthreads();
    // ---- P1: per wave, 3 chunks of 16 halo pixels: GEMM1 -> GELU -> GEMM2 -> h2 tile
    for (int ch = 0; ch < 3; ++ch) {
        const int row0 = (wave * 3 + ch) * 16;
        long prow[4];
        bool inner[4], inimg[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int m = row0 + 4 * g + v;
            const int hy = m / HX, hx = m - hy * HX;
            const int y = y0 + hy - 1, x = x0 + hx - 1;
            inimg[v] = (m < NH) && y >= 0 && y < h && x >= 0 && x < w;
            inner[v] = inimg[v] && hy >= 1 && hy <= TY && hx >= 1 && hx <= TX;
            prow[v] = (b * h + y) * (long)w + x;
        }
        for (int nc = 0; nc < N1; nc += 64) {
            f32x4 acc[1][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (E == 16) {
                wave_gemm_bf_k16<1, 4>(acc, bufA + row0 * LDA, LDA, w1k16);
            } else {
                constexpr int KB1 = E / 32;
                bf16x8 wf[4][KB1];
                load_bfrag_bf16<4, KB1>(wf, a1.w1 + (size_t)nc * E, E);
                wave_gemm_bf<1, 4, KB1>(acc, bufA + row0 * LDA, LDA, wf);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = nc + nt * 16 + r;
                const float bias = sB1[col];
#pragma unroll
                for (int v = 0; v < 4; v += 2) {   // packed pairs (v_pk_fma_f32)
                    const lg_v2f hh = (lg_v2f){acc[0][nt][v] + bias, acc[0][nt][v + 1] + bias};
                    lg_v2f av;
                    if (SAVE) {
                        lg_v2f gv;
                        gelu2_both_f(hh, av, gv);
                        if (inner[v]) { HS<BF>::st1(a1.a1s, prow[v] * N1 + col, av.x); HS<BF>::st1(a1.g1s, prow[v] * N1 + col, gv.x); }
                        if (inner[v + 1]) { HS<BF>::st1(a1.a1s, prow[v + 1] * N1 + col, av.y); HS<BF>::st1(a1.g1s, prow[v + 1] * N1 + col, gv.y); }
                    } else {
                        av = gelu2_f(hh);
                    }
                    my[(4 * g + v) * LDS16 + col] = (__bf16)av.x;
                    my[(4 * g + v + 1) * LDS16 + col] = (__bf16)av.y;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int nc = 0; nc < N1; nc += 64) {
            f32x4 acc[1][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (E == 16) {
                wave_gemm_bf<1, 4, KB2>(acc, my, LDS16, w2f);
            } else {
                bf16x8 wf[4][KB2];
                load_bfrag_bf16<4, KB2>(wf, a1.w2 + (size_t)nc * N1, N1);
                wave_gemm_bf<1, 4, KB2>(acc, my, LDS16, wf);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = nc + nt * 16 + r;
                const float bias = sB2[col];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int m = row0 + 4 * g + v;
                    const float hh = inimg[v] ? acc[0][nt][v] + bias : 0.f;   // dep_conv zero-pads h2 (basic_module_unformer_v2.py:18)
                    if (SAVE && inner[v]) HS<BF>::st1(a1.h2, prow[v] * N1 + col, hh);
                    if (m < NH) bufH2[m * LDH + col] = hh;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- P2: per wave, 2 chunks of 16 inner pixels: dw3x3 + GELU (fp32) -> bf16 scratch -> GEMM3 -> output tile
    {
        const int q = lane % CQ;
        float wq[4][9], bq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 9; ++k) wq[u][k] = sDww[(4 * q + u) * 9 + k];
            bq[u] = sDwb[4 * q + u];
        }
        for (int ch = 0; ch < 2; ++ch) {
            const int m0 = (wave * 2 + ch) * 16;
            for (int mm = lane / CQ; mm < 16; mm += 64 / CQ) {
                const int m = m0 + mm;
                const int ty = m / TX, tx = m - ty * TX;
                float4 acc = make_float4(bq[0], bq[1], bq[2], bq[3]);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float4 v = *reinterpret_cast<const float4*>(bufH2 + ((ty + dy) * HX + tx + dx) * LDH + 4 * q);
                        acc.x += wq[0][dy * 3 + dx] * v.x; acc.y += wq[1][dy * 3 + dx] * v.y;
                        acc.z += wq[2][dy * 3 + dx] * v.z; acc.w += wq[3][dy * 3 + dx] * v.w;
                    }
                const int y = y0 + ty, x = x0 + tx;
                float4 av;
                if (SAVE) {
                    lg_v2f a01, a23, g01, g23;
                    gelu2_both_f((lg_v2f){acc.x, acc.y}, a01, g01);
                    gelu2_both_f((lg_v2f){acc.z, acc.w}, a23, g23);
                    av = make_float4(a01.x, a01.y, a23.x, a23.y);
                    if (y < h && x < w) {
                        const long o = ((b * h + y) * (long)w + x) * N1 + 4 * q;
                        HS<BF>::st4(a2.a3s, o, av);
                        HS<BF>::st4(a2.g3s, o, make_float4(g01.x, g01.y, g23.x, g23.y));
                    }
                } else {
                    const lg_v2f a01 = gelu2_f((lg_v2f){acc.x, acc.y}), a23 = gelu2_f((lg_v2f){acc.z, acc.w});
                    av = make_float4(a01.x, a01.y, a23.x, a23.y);
                }
                *reinterpret_cast<bf16x4*>(my + mm * LDS16 + 4 * q) = (bf16x4){(__bf16)av.x, (__bf16)av.y, (__bf16)av.z, (__bf16)av.w};
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            f32x4 acc3[1][NT3];
#pragma unroll
            for (int nt = 0; nt < NT3; ++nt) acc3[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (E == 16) {
                wave_gemm_bf<1, NT3, KB2>(acc3, my, LDS16, w3f);
            } else {
                bf16x8 wf[NT3][KB2];
                load_bfrag_bf16<NT3, KB2>(wf, a2.w3, N1);
                wave_gemm_bf<1, NT3, KB2>(acc3, my, LDS16, wf);
            }
#pragma unroll
            for (int nt = 0; nt < NT3; ++nt) {
                const int col = nt * 16 + r;
                const float bias = sB3[col];
#pragma unroll
                for (int v = 0; v < 4; ++v) bufO[(m0 + 4 * g + v) * LDO + col] = acc3[0][nt][v] + bias;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    // ---- P3: residual, store, planar LN1 half for the next block (fp32)
    if (threadIdx.x < M) {
        const int m = threadIdx.x;
        const int y = y0 + m / TX, x = x0 + m % TX;
        if (y < h && x < w) {
            const long p = (b * h + y) * (long)w + x;
            float o[E];
            const float4* xs = reinterpret_cast<const float4*>(a2.x + p * E);
            float4* yo = reinterpret_cast<float4*>(a2.y + p * E);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) {
                float4 xr = xs[k];
                o[4 * k] = xr.x + bufO[m * LDO + 4 * k];
                o[4 * k + 1] = xr.y + bufO[m * LDO + 4 * k + 1];
                o[4 * k + 2] = xr.z + bufO[m * LDO + 4 * k + 2];
                o[4 * k + 3] = xr.w + bufO[m * LDO + 4 * k + 3];
                yo[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
            }
            if (a2.g) {
                float mu, rstd;
                ln_stats<E>(o, mu, rstd);
                const long hw = (long)h * w, s = (long)y * w + x;
#pragma unroll
                for (int n = E / 2; n < E; ++n) a2.g[(b * (E / 2) + (n - E / 2)) * hw + s] = (o[n] - mu) * rstd * sN1g[n] + sN1b[n];
            }
        }
    }
    }   // tiles of this workgroup
}

template <int E>
static int launch_ffn_fused_bf_t(const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s) {
    ProfScope prof__(LG_K_FFN2, s);
    constexpr int N1 = 4 * E;
    const size_t lds = (size_t)(180 * (N1 + 4) + 128 * (E + 1) + 8) * sizeof(float) + (size_t)(192 * (E + 8) + 4 * 16 * (N1 + 8)) * 2;
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn_fused_bf<E, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_fused_bf<E, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) { lg_set_error("ffn_fused_bf: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    int tiles_x = (a2.w + 15) / 16, tiles_y = (a2.h + 7) / 8;
    const int ntiles = a2.B * tiles_x * tiles_y;
    const int grid = ntiles < 512 ? ntiles : 512;
    if (a1.a1s != nullptr) k_ffn_fused_bf<E, true, true><<<grid, 256, lds, s>>>(a1, a2, tiles_x, tiles_y, ntiles);
    else k_ffn_fused_bf<E, false, true><<<grid, 256, lds, s>>>(a1, a2, tiles_x, tiles_y, ntiles);
    LG_CHECK_LAUNCH();
    return 0;
}

// returns LG_FFN_NOT_FUSED if the fused kernels do not cover this size (caller falls back to k_ffn1 + k_ffn2)
int launch_ffn_fused(int e, const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s) {
    if (a1.hbf) {   // throughput mode: bf16 matrix cores for the three GEMMs
#ifdef LG_BUILD_AB   // round 1's bf16 tile kernels: A/B builds only (lg_plan_create rejects precision = 1 with an FFN variant otherwise)
        if (e == 16 && a1.tile16 != 0) return launch_ffn_fused_bf_t<16>(a1, a2, s);
        if (e == 32 && (a1.tile16 != 0 || !a1.wsplit)) return launch_ffn_fused_bf_t<32>(a1, a2, s);
#endif
        if (e == 16 && (a1.tile16 == 0 || a1.tile16 == 4)) {                            // k_ffn_xr<., NP = 1> (round 6) / k_ffn_xs<., NP = 1> (the other save modes, LG_VAR_FFN_XS)
            const bool save = a1.h2 != nullptr, h2h3 = save && !a1.a1s && !a1.g1s && a2.a3s && !a2.g3s;
            if (a1.tile16 == 0 && (!save || h2h3)) return launch_ffn_xr(a1, a2, s);
            return launch_ffn_xs(a1, a2, s);
        }
        if (e == 32 && a1.tile16 == 0 && a1.wsplit) return launch_ffn_x32(a1, a2, s);   // k_ffn_x32<., NP = 1>
        if (e == 64) return LG_FFN_NOT_FUSED;
        lg_set_error("ffn: precision = 1 with an FFN variant needs a `make AB=1` build");
        return -2;
    }
    // e = 16: the strip kernel on the bf16 matrix pipe in fp32-equivalent split arithmetic (k_ffn_x.hip); for A/B runs the plan's
    // switch (lg_config.variant & LG_VAR_FFN_IMPL_MASK) selects the f32-MFMA strip kernel (1), the per-tile kernel (2: `make AB=1` builds)
    // it replaced, or the software-pipelined variant k_ffn_xp (3: same results bit for bit, measured 2.5 % slower)
#ifdef LG_BUILD_AB
    if (e == 16 && a1.tile16 == 3) return launch_ffn_xp(a1, a2, s);
#else
    if (e == 16 && a1.tile16 == 3) { lg_set_error("LG_FFN_IMPL=xp: k_ffn_xp is an A/B kernel, build the library with `make AB=1`"); return -2; }
#endif
#ifdef LG_BUILD_AB
    if (e == 16 && a1.tile16 == 2) return launch_ffn_fused_t<16>(a1, a2, s);   // round 1's per-tile kernel at e = 16: A/B builds only
#endif
    if (e == 16 && a1.tile16 == 1) return launch_ffn_strip(a1, a2, s);
    if (e == 16) {
        // round 6: the register chain k_ffn_xr where it exists (f16 pairs, nothing or h2 / h3 saved); k_ffn_xs for the bf16 x 3 arithmetic, the
        // other save modes and as the A/B variant LG_VAR_FFN_XS
        const bool save = a1.h2 != nullptr, h2h3 = save && !a1.a1s && !a1.g1s && !a2.g3s;   // (a3s null: h2 alone, the backward re-computes h3)
        if (a1.tile16 == 0 && a1.scales && (!save || h2h3)) return launch_ffn_xr(a1, a2, s);
        return launch_ffn_xs(a1, a2, s);
    }
    if (e == 32) return (a1.tile16 == 0 && a1.wsplit) ? launch_ffn_x32(a1, a2, s) : launch_ffn_fused_t<32>(a1, a2, s);
    if (e == 64 && a1.tile16 == 0 && a1.wsplit && a1.h2) return launch_ffn_x64(a1, a2, s);   // two kernels, split-bf16 GEMMs
    return LG_FFN_NOT_FUSED;
}
