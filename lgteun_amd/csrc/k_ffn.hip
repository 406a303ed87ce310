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

// ------------------------------------------------------------------------------------------------
// k_ffn1: each wave owns MW = 16*MT pixels end to end (no inter-wave dependency)
// ------------------------------------------------------------------------------------------------
template <int E, int MT>
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
                    if (a.h1 && p0 + row < a.P) a.h1[(p0 + row) * N1 + col] = h;
                    bufH[row * LDH + col] = gelu_f(h);
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
                    if (p0 + row < a.P) a.h2[(p0 + row) * N1 + col] = acc[mt][nt][v] + bias;
                }
            }
    }
}

template <int E, int MT>
static int launch_ffn1_t(const Ffn1Args& a, hipStream_t s) {
    ProfScope prof__(LG_K_FFN1, s);
    constexpr int N1 = 4 * E, MW = 16 * MT;
    size_t lds = (size_t)4 * MW * ((E + 4) + (N1 + 4)) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn1<E, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) { lg_set_error("ffn1: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done = true;
    }
    long per_wg = 4L * MW;
    int grid = (int)((a.P + per_wg - 1) / per_wg);
    k_ffn1<E, MT><<<grid, 256, lds, s>>>(a);
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
template <int E, int MT, int TY, int TX>
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
                        const float4 v = *reinterpret_cast<const float4*>(a.h2 + ((b * a.h + yy) * (long)a.w + xx) * N1 + 4 * q);
                        acc.x += wq[0][dy * 3 + dx] * v.x;
                        acc.y += wq[1][dy * 3 + dx] * v.y;
                        acc.z += wq[2][dy * 3 + dx] * v.z;
                        acc.w += wq[3][dy * 3 + dx] * v.w;
                    }
                }
                acc.x += bq[0]; acc.y += bq[1]; acc.z += bq[2]; acc.w += bq[3];
                if (a.h3) *reinterpret_cast<float4*>(a.h3 + ((b * a.h + y) * (long)a.w + x) * N1 + 4 * q) = acc;
                acc = make_float4(gelu_f(acc.x), gelu_f(acc.y), gelu_f(acc.z), gelu_f(acc.w));
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
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn2<E, MT, TY, TX>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) { lg_set_error("ffn2: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done = true;
    }
    int tiles_x = (a.w + TX - 1) / TX, tiles_y = (a.h + TY - 1) / TY;
    int grid = a.B * tiles_x * tiles_y;
    k_ffn2<E, MT, TY, TX><<<grid, 256, lds, s>>>(a, tiles_x, tiles_y);
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
