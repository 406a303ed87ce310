// Backward of feed_forward (k_ffn.hip) for gfx950 -- autograd of reference models/common/LGT.py:91-109 with the
// pre_norm/residual wrappers (LGT.py:45-61).  Data gradients here; the three 1x1-conv weight gradients are
// pixel-reduction GEMMs in k_wgrad.hip fed by the tensors this file materialises (dh3, dh2, dh1, LN2(x)).
//   k_ffn2_bwd : dh3 = (dy W3) * gelu'(h3)
//   k_ffn1_bwd : dh2 = dw^T dh3 ; dh1 = (dh2 W2) * gelu'(h1) ; dx = dy + LN2^T(dh1 W1) ; dw3x3 / LN2 param grads
// GEMMs on v_mfma_f32_16x16x4_f32 through wave_gemm (weights pre-transposed once per step).
#include "kernels.h"
#include "bwd_kernels.h"
#include "mfma.h"

__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    long n = (long)rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        int r = (int)(i / cols), c = (int)(i - (long)r * cols);
        dst[(long)c * rows + r] = src[i];
    }
}
int launch_transpose(const float* src, float* dst, int rows, int cols, hipStream_t s) {
    long n = (long)rows * cols;
    int grid = (int)((n + 255) / 256);
    if (grid > 1024) grid = 1024;
    k_transpose<<<grid, 256, 0, s>>>(src, dst, rows, cols);
    LG_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
template <int E, int MT>
__global__ __launch_bounds__(256) void k_ffn2_bwd(Ffn2BwdArgs a) {
    constexpr int N1 = 4 * E, MW = 16 * MT, LDA = E + 4;
    extern __shared__ float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    float* bufA = smem + wave * (MW * LDA);
    const long p0 = ((long)blockIdx.x * 4 + wave) * MW;
    for (int i = lane; i < MW * (E / 4); i += 64) {
        const int m = i / (E / 4), k4 = i - m * (E / 4);
        long p = p0 + m;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < a.P) v = *reinterpret_cast<const float4*>(a.dy + p * E + 4 * k4);
        *reinterpret_cast<float4*>(bufA + m * LDA + 4 * k4) = v;
    }
    __syncthreads();
    for (int nc = 0; nc < N1; nc += 64) {
        f32x4 acc[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        wave_gemm<MT, 4, E>(acc, bufA, LDA, a.w3t + (size_t)nc * E);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = nc + nt * 16 + r;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const long p = p0 + mt * 16 + 4 * g + v;
                    if (p < a.P) a.dh3[p * N1 + col] = acc[mt][nt][v] * gelu_grad_f(a.h3[p * N1 + col]);
                }
            }
    }
}

template <int E, int MT>
static int launch_ffn2_bwd_t(const Ffn2BwdArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_FFN2_BWD, s);
    constexpr int MW = 16 * MT;
    size_t lds = (size_t)4 * MW * (E + 4) * sizeof(float);
    long per_wg = 4L * MW;
    int grid = (int)((a.P + per_wg - 1) / per_wg);
    k_ffn2_bwd<E, MT><<<grid, 256, lds, s>>>(a);
    LG_CHECK_LAUNCH();
    return 0;
}
int launch_ffn2_bwd(int e, const Ffn2BwdArgs& a, hipStream_t s) {
    if (e == 16) return launch_ffn2_bwd_t<16, 2>(a, s);
    if (e == 32) return launch_ffn2_bwd_t<32, 2>(a, s);
    if (e == 64) return launch_ffn2_bwd_t<64, 1>(a, s);
    lg_set_error("ffn2_bwd: e=%d unsupported", e);
    return -1;
}

// ------------------------------------------------------------------------------------------------
template <int E, int MT, int TY, int TX>
__global__ __launch_bounds__(256) void k_ffn1_bwd(Ffn1BwdArgs a, int tiles_x, int tiles_y) {
    constexpr int N1 = 4 * E, M = TY * TX, LDH = N1 + 4, LDO = E + 1, CQ = N1 / 4, NTE = E / 16;
    static_assert(M == 64 * MT, "tile");
    extern __shared__ float smem[];
    float* bufH = smem;                  // [M][LDH] dh2
    float* bufH2 = smem + M * LDH;       // [M][LDH] dh1
    float* bufO = bufH2 + M * LDH;       // [M][LDO] d(LN2 output)
    float* sDw = bufO + M * LDO;         // [N1*10] dw3x3 weight/bias grad partials; later [4*2E] reduction scratch
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    int t = blockIdx.x;
    const int tx_i = t % tiles_x;
    t /= tiles_x;
    const int ty_i = t % tiles_y;
    const long b = t / tiles_y;
    const int y0 = ty_i * TY, x0 = tx_i * TX;
    for (int i = threadIdx.x; i < N1 * 10; i += 256) sDw[i] = 0.f;
    __syncthreads();
    // ---- stage 1: dh2 = dw^T dh3 ; depthwise weight/bias gradient partials
    {
        const int q = threadIdx.x % CQ;
        float wq[4][9];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 9; ++k) wq[u][k] = a.dww[(4 * q + u) * 9 + k];
        float pw[4][10];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 10; ++k) pw[u][k] = 0.f;
        for (int m = threadIdx.x / CQ; m < M; m += 256 / CQ) {
            const int y = y0 + m / TX, x = x0 + m % TX;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y < a.h && x < a.w) {
                const long pc = (b * a.h + y) * (long)a.w + x;
                const float4 gc = *reinterpret_cast<const float4*>(a.dh3 + pc * N1 + 4 * q);
                pw[0][9] += gc.x; pw[1][9] += gc.y; pw[2][9] += gc.z; pw[3][9] += gc.w;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        // forward: h3(y',x') += w[dy][dx] * h2(y'+dy-1, x'+dx-1)
                        const int ys = y - dy + 1, xs = x - dx + 1;   // output position fed by this h2 pixel through tap (dy,dx)
                        if (ys >= 0 && ys < a.h && xs >= 0 && xs < a.w) {
                            const float4 gv = *reinterpret_cast<const float4*>(a.dh3 + ((b * a.h + ys) * (long)a.w + xs) * N1 + 4 * q);
                            acc.x += wq[0][dy * 3 + dx] * gv.x; acc.y += wq[1][dy * 3 + dx] * gv.y;
                            acc.z += wq[2][dy * 3 + dx] * gv.z; acc.w += wq[3][dy * 3 + dx] * gv.w;
                        }
                        const int yi = y + dy - 1, xi = x + dx - 1;   // h2 pixel this output reads through tap (dy,dx)
                        if (yi >= 0 && yi < a.h && xi >= 0 && xi < a.w) {
                            const float4 hv = *reinterpret_cast<const float4*>(a.h2 + ((b * a.h + yi) * (long)a.w + xi) * N1 + 4 * q);
                            pw[0][dy * 3 + dx] += gc.x * hv.x; pw[1][dy * 3 + dx] += gc.y * hv.y;
                            pw[2][dy * 3 + dx] += gc.z * hv.z; pw[3][dy * 3 + dx] += gc.w * hv.w;
                        }
                    }
                }
                *reinterpret_cast<float4*>(a.dh2 + pc * N1 + 4 * q) = acc;
            }
            *reinterpret_cast<float4*>(bufH + m * LDH + 4 * q) = acc;
        }
        // threads with equal q inside a wave: lanes q, q+CQ, ...
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                float v = pw[u][k];
#pragma unroll
                for (int off = CQ; off < 64; off <<= 1) v += __shfl_xor(v, off);
                if (lane < CQ) atomicAdd(&sDw[(4 * q + u) * 10 + k], v);
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N1 * 10; i += 256) {
        const int c = i / 10, k = i - c * 10;
        if (k < 9) atomicAdd(a.d_dww + c * 9 + k, sDw[i]); else atomicAdd(a.d_dwb + c, sDw[i]);
    }
    // ---- stage 2: dh1 = (dh2 W2) * gelu'(h1): wave owns rows [wave*16*MT, +16*MT)
    const int row0 = wave * 16 * MT;
    for (int nc = 0; nc < N1; nc += 64) {
        f32x4 acc[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        wave_gemm<MT, 4, N1>(acc, bufH + row0 * LDH, LDH, a.w2t + (size_t)nc * N1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = nc + nt * 16 + r;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int m = row0 + mt * 16 + 4 * g + v;
                    const int y = y0 + m / TX, x = x0 + m % TX;
                    float d = 0.f;
                    if (y < a.h && x < a.w) {
                        const long p = (b * a.h + y) * (long)a.w + x;
                        d = acc[mt][nt][v] * gelu_grad_f(a.h1[p * N1 + col]);
                        a.dh1[p * N1 + col] = d;
                    }
                    bufH2[m * LDH + col] = d;
                }
            }
    }
    __syncthreads();
    // ---- stage 3: d(LN2 out) = dh1 W1
    {
        f32x4 acc[MT][NTE];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTE; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        wave_gemm<MT, NTE, N1>(acc, bufH2 + row0 * LDH, LDH, a.w1t);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTE; ++nt)
#pragma unroll
                for (int v = 0; v < 4; ++v) bufO[(row0 + mt * 16 + 4 * g + v) * LDO + nt * 16 + r] = acc[mt][nt][v];
    }
    __syncthreads();
    // ---- stage 4: LayerNorm backward + residual, LN2 param grads
    float pl[2 * E];
#pragma unroll
    for (int i = 0; i < 2 * E; ++i) pl[i] = 0.f;
    if (threadIdx.x < M) {
        const int m = threadIdx.x;
        const int y = y0 + m / TX, x = x0 + m % TX;
        if (y < a.h && x < a.w) {
            const long p = (b * a.h + y) * (long)a.w + x;
            float xv[E];
            const float4* xs = reinterpret_cast<const float4*>(a.x + p * E);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) {
                float4 v = xs[k];
                xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
            }
            float mu, rstd;
            ln_stats<E>(xv, mu, rstd);
            float m1 = 0.f, m2 = 0.f;
            float dxh[E];
#pragma unroll
            for (int c = 0; c < E; ++c) {
                const float xh = (xv[c] - mu) * rstd;
                const float dyl = bufO[m * LDO + c];
                pl[c] = dyl * xh;
                pl[E + c] = dyl;
                dxh[c] = dyl * a.ln2g[c];
                m1 += dxh[c];
                m2 += dxh[c] * xh;
                xv[c] = xh;
            }
            m1 *= (1.0f / E);
            m2 *= (1.0f / E);
            const float4* dys = reinterpret_cast<const float4*>(a.dy + p * E);
            float4* dxo = reinterpret_cast<float4*>(a.dx + p * E);
            float4* y2o = reinterpret_cast<float4*>(a.y2 + p * E);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) {
                float4 dv = dys[k];
                float o[4], yv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = 4 * k + u;
                    o[u] = rstd * (dxh[c] - m1 - xv[c] * m2);
                    yv[u] = xv[c] * a.ln2g[c] + a.ln2b[c];
                }
                dxo[k] = make_float4(dv.x + o[0], dv.y + o[1], dv.z + o[2], dv.w + o[3]);
                y2o[k] = make_float4(yv[0], yv[1], yv[2], yv[3]);
            }
        }
    }
    // block reduction of the LN2 parameter partials (sDw region is free now: global atomics above were issued from it
    // before stage 2's barrier... they read sDw -> make sure they are done)
    __syncthreads();
    {
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < 2 * E; ++i) {
            float s = pl[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (ln == 0) sDw[wv * 2 * E + i] = s;
        }
        __syncthreads();
        if (threadIdx.x < 2 * E) {
            const float s = sDw[threadIdx.x] + sDw[2 * E + threadIdx.x] + sDw[4 * E + threadIdx.x] + sDw[6 * E + threadIdx.x];
            if (threadIdx.x < E) atomicAdd(a.d_ln2g + threadIdx.x, s); else atomicAdd(a.d_ln2b + threadIdx.x - E, s);
        }
    }
}

template <int E, int MT, int TY, int TX>
static int launch_ffn1_bwd_t(const Ffn1BwdArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_FFN1_BWD, s);
    constexpr int N1 = 4 * E, M = TY * TX;
    size_t lds = (size_t)(2 * M * (N1 + 4) + M * (E + 1) + N1 * 10) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn1_bwd<E, MT, TY, TX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { lg_set_error("ffn1_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done = true;
    }
    int tiles_x = (a.w + TX - 1) / TX, tiles_y = (a.h + TY - 1) / TY;
    int grid = a.B * tiles_x * tiles_y;
    k_ffn1_bwd<E, MT, TY, TX><<<grid, 256, lds, s>>>(a, tiles_x, tiles_y);
    LG_CHECK_LAUNCH();
    return 0;
}
int launch_ffn1_bwd(int e, const Ffn1BwdArgs& a, hipStream_t s) {
    if (e == 16) return launch_ffn1_bwd_t<16, 2, 8, 16>(a, s);
    if (e == 32) return launch_ffn1_bwd_t<32, 2, 8, 16>(a, s);
    if (e == 64) return launch_ffn1_bwd_t<64, 1, 8, 8>(a, s);
    lg_set_error("ffn1_bwd: e=%d unsupported", e);
    return -1;
}
