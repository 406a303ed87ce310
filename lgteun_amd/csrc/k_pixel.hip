// Pixelwise / resampling kernels of the LGTEUN hot path for gfx950.
//  - data module D / DT / R / RT and the proximal-gradient update (reference models/unlg_former.py:29-37,58-61)
//  - LGT patch_embed, down, up+fusion, tail (reference models/common/LGT.py:64-88,280-281,294-295,302-303,337-342)
// Activations inside an LGT are NHWC (one pixel = one contiguous channel vector); the data module works on
// NCHW fp32 planes (C is 4 or 8).  One thread per pixel; weights are wave-uniform (scalar loads).
#include "kernels.h"
#include "resample_tile.h"

// ------------------------------------------------------------------------------------------------
// plain bicubic resample of planes (bmu.sampling_, basic_module_unformer_v2.py:21-23)
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_resample(const float* __restrict__ x, float* __restrict__ y, int planes, int hi,
                                                  int wi, int ho, int wo) {
    long total = (long)planes * ho * wo;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        int ox = (int)(i % wo);
        long r = i / wo;
        int oy = (int)(r % ho);
        long p = r / ho;
        y[i] = resample_at<MODE>(x + p * hi * wi, hi, wi, oy, ox);
    }
}

int launch_resample(int mode, const float* x, float* y, int planes, int hi, int wi, hipStream_t s) {
    int ho, wo;
    if (mode == 0) { ho = hi / 2; wo = wi / 2; } else if (mode == 1) { ho = hi * 2; wo = wi * 2; } else { ho = hi * 4; wo = wi * 4; }
    long total = (long)planes * ho * wo;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (mode == 0) k_resample<0><<<grid, 256, 0, s>>>(x, y, planes, hi, wi, ho, wo);
    else if (mode == 1) k_resample<1><<<grid, 256, 0, s>>>(x, y, planes, hi, wi, ho, wo);
    else k_resample<2><<<grid, 256, 0, s>>>(x, y, planes, hi, wi, ho, wo);
    LG_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// bicubic resample followed by depthwise 3x3 (zero padding) -- one stage of D or DT, LDS-tiled:
// a 32x32 output tile needs the 34x34 resampled halo tile (zero outside the image).
// ------------------------------------------------------------------------------------------------
template <int MODE, int EPI>
__global__ __launch_bounds__(256) void k_resample_dw(DwArgs a) {
    __shared__ float U[34][35];
    // x2: separable evaluation through LDS (resample_tile.h); x0.5 reads a 70 x 70 source window per tile, where the direct 16-tap
    // gather measured faster (8.2 vs 8.7 us)
    __shared__ float rs_scratch[MODE == 1 ? RsTile<1>::FLOATS : 1];
    const int plane = blockIdx.z;
    const int c = plane % a.C;
    const int b = plane / a.C;
    const int ty0 = blockIdx.y * 32, tx0 = blockIdx.x * 32;
    const float* __restrict__ in = a.in + (size_t)plane * a.hi * a.wi;
    // EPI 2: the update's pixelwise operands (all C planes of Z, PAN) are requested before the LDS phases, not inside the pixel loop
    float rzp[EPI == 2 ? 4 : 1], zc[EPI == 2 ? 4 : 1];
    if constexpr (EPI == 2) {
        const size_t hw = (size_t)a.ho * a.wo;
        const float* __restrict__ zb = a.z + (size_t)b * a.C * hw;
        // all band planes of the thread's four pixels are requested before the first one is used (a run-time `for (cc < C)` loop kept one load + wait
        // per band: C dependent round trips per pixel); C <= 8 (launcher), bands beyond C repeat the last plane with a zero weight
        constexpr int CMAX = 8;
        float zv[4][CMAX], pn[4], rwv[CMAX];
#pragma unroll
        for (int cc = 0; cc < CMAX; ++cc) rwv[cc] = cc < a.C ? a.rw[cc < a.C ? cc : 0] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k;
            const int oy = ty0 + (i >> 5), ox = tx0 + (i & 31);
            const size_t pix = (oy < a.ho && ox < a.wo) ? (size_t)oy * a.wo + ox : 0;
#pragma unroll
            for (int cc = 0; cc < CMAX; ++cc) zv[k][cc] = zb[(size_t)(cc < a.C ? cc : a.C - 1) * hw + pix];
            pn[k] = a.pan[(size_t)b * hw + pix];
        }
        const float rb0 = a.rb[0];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float r = rb0;
            zc[k] = 0.f;
#pragma unroll
            for (int cc = 0; cc < CMAX; ++cc) {
                if (cc < a.C) r += rwv[cc] * zv[k][cc];   // (same order of additions as the band loop)
                zc[k] = cc == c ? zv[k][cc] : zc[k];
            }
            rzp[k] = r - pn[k];
        }
    }
    if constexpr (MODE == 1) resample_tile34<1>(in, a.hi, a.wi, a.ho, a.wo, ty0, tx0, U, rs_scratch);
    else {
        for (int i = threadIdx.x; i < 34 * 34; i += 256) {
            int uy = i / 34, ux = i - uy * 34;
            int oy = ty0 + uy - 1, ox = tx0 + ux - 1;
            float v = 0.f;
            if (oy >= 0 && oy < a.ho && ox >= 0 && ox < a.wo) v = resample_at<MODE>(in, a.hi, a.wi, oy, ox);
            U[uy][ux] = v;
        }
    }
    __syncthreads();
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = a.w9[c * 9 + k];
    const float bias = a.bias[c];
    float rtw_c = 0.f, rtb_c = 0.f, eta = 0.f;
    if (EPI == 2) { rtw_c = a.rtw[c]; rtb_c = a.rtb[c]; eta = a.eta[0]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + 256 * k;
        int ly = i >> 5, lx = i & 31;
        int oy = ty0 + ly, ox = tx0 + lx;
        if (oy < a.ho && ox < a.wo) {
            float v = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) v += w[dy * 3 + dx] * U[ly + dy][lx + dx];
            v += bias;
            size_t o = ((size_t)plane * a.ho + oy) * a.wo + ox;
            if (EPI == 1) v -= a.sub[o];
            if (EPI == 2) {
                // Z <- Z - eta * (ms_term + RT(R(Z) - pan))      unlg_former.py:59-61
                const float pan_term = rtw_c * rzp[k] + rtb_c;
                v = zc[k] - eta * (v + pan_term);
            }
            a.out[o] = v;
        }
    }
}

int launch_resample_dw(int mode, int epi, const DwArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_DATASTEP, s);
    dim3 grid((a.wo + 31) / 32, (a.ho + 31) / 32, a.planes);
    if (epi == 2 && a.C > 8) { lg_set_error("resample_dw: the fused update holds at most 8 bands per pixel (C=%d)", a.C); return -2; }
#define LG_RDW(M, E) k_resample_dw<M, E><<<grid, 256, 0, s>>>(a)
    if (mode == 0 && epi == 0) LG_RDW(0, 0);
    else if (mode == 0 && epi == 1) LG_RDW(0, 1);
    else if (mode == 1 && epi == 0) LG_RDW(1, 0);
    else if (mode == 1 && epi == 2) LG_RDW(1, 2);
    else { lg_set_error("resample_dw: unsupported mode/epi %d/%d", mode, epi); return -1; }
#undef LG_RDW
    LG_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// helper: emit the LayerNorm-ed global half of a pixel vector as planar [B,E/2,H,W]
// (input of the next block's FFT mixer; LN over all E channels, LGT.py:58,203-206)
// ------------------------------------------------------------------------------------------------
template <int E>
__device__ __forceinline__ void emit_g(const float (&x)[E], const float* __restrict__ n1g, const float* __restrict__ n1b,
                                       float* __restrict__ g, long b, long s, long HW) {
    float mu, rstd;
    ln_stats<E>(x, mu, rstd);
#pragma unroll
    for (int n = E / 2; n < E; ++n) g[(b * (E / 2) + (n - E / 2)) * HW + s] = (x[n] - mu) * rstd * n1g[n] + n1b[n];
}

// ------------------------------------------------------------------------------------------------
// patch_embedding (patch_size 1): dw1x1 -> 1x1 C->E -> LayerNorm(E)        LGT.py:64-88
// ------------------------------------------------------------------------------------------------
// (a lane = (pixel, channel quad) form of this kernel measured SLOWER, 21 vs 17 us: its planar input and planar LN-half output want a
// lane per pixel; only the NHWC store gains)
template <int C, int E>
__global__ __launch_bounds__(256) void k_embed(EmbedArgs a) {
    long p = blockIdx.x * 256L + threadIdx.x;
    if (p >= a.total) return;
    long b = p / a.HW, s = p - b * a.HW;
    float t[C];
#pragma unroll
    for (int c = 0; c < C; ++c) t[c] = a.z[(b * C + c) * a.HW + s] * a.dww[c] + a.dwb[c];
    float e[E];
#pragma unroll
    for (int n = 0; n < E; ++n) {
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) v += a.w[n * C + c] * t[c];
        e[n] = v + a.b[n];
    }
    float mu, rstd;
    ln_stats<E>(e, mu, rstd);
#pragma unroll
    for (int n = 0; n < E; ++n) e[n] = (e[n] - mu) * rstd * a.lng[n] + a.lnb[n];
#ifndef LG_EMBED_TR
#define LG_EMBED_TR 1
#endif
    if (LG_EMBED_TR && a.total % 64 == 0) {
        // the NHWC rows leave COALESCED: the wave's 64 pixel vectors pass through a wave-private LDS image and every store instruction
        // writes 1 KB of consecutive bytes (lane -> (pixel 16 j + lane / Q4, quad lane % Q4)); as one row per lane a store instruction is
        // 64 separate 16-byte pieces at a 64- / 128-byte stride
        constexpr int Q4 = E / 4, LDT = E + 4;
            __shared__ __attribute__((aligned(16))) float tr[4][64 * LDT];
        const int lane = threadIdx.x & 63;
        float* mine = tr[threadIdx.x >> 6];
#pragma unroll
        for (int n = 0; n < Q4; ++n) *reinterpret_cast<float4*>(mine + lane * LDT + 4 * n) = make_float4(e[4 * n], e[4 * n + 1], e[4 * n + 2], e[4 * n + 3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float4* xo = reinterpret_cast<float4*>(a.x + (p - lane) * E);
#pragma unroll
        for (int j = 0; j < Q4; ++j) {
            const int i = j * 64 + lane;                 // float4 index inside the wave's 64 x E block
            xo[i] = *reinterpret_cast<const float4*>(mine + (i / Q4) * LDT + 4 * (i % Q4));
        }
    } else {
        float4* xo = reinterpret_cast<float4*>(a.x + p * E);
#pragma unroll
        for (int n = 0; n < E / 4; ++n) xo[n] = make_float4(e[4 * n], e[4 * n + 1], e[4 * n + 2], e[4 * n + 3]);
    }
    if (a.g) emit_g<E>(e, a.n1g, a.n1b, a.g, b, s, a.HW);
}

int launch_embed(int C, const EmbedArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_EMBED, s);
    int grid = (int)((a.total + 255) / 256);
    if (C == 4) k_embed<4, 16><<<grid, 256, 0, s>>>(a);
    else if (C == 8) k_embed<8, 32><<<grid, 256, 0, s>>>(a);
    else { lg_set_error("embed: C=%d unsupported", C); return -1; }
    LG_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// encoder down-sampling: bicubic x0.5 (per channel) then 1x1 E->2E          LGT.py:280-281,325-326
// ------------------------------------------------------------------------------------------------
// Lane = (output pixel, channel quad): the E/4 lanes of a level-1 pixel each gather ONE float4 per tap, so a wave-load touches whole
// 64-byte (E = 16) / 128-byte pixel vectors instead of 64 separate 16-byte pieces (a lane per pixel ran at 2.4 TB/s); the resampled vector
// is exchanged through LDS inside the wave, lane q computes outputs 8q .. 8q+7 of the 1x1 conv and stores 32 contiguous bytes of the
// pixel's row.  Same tap and channel order as before: same values.
template <int E>
__global__ __launch_bounds__(256) void k_down(DownArgs a) {
    constexpr int LPP = E / 4, PPW = 256 / LPP, NO = 2 * E / LPP, LDU = E + 4;
    static_assert(NO == 8, "eight outputs per lane");
    __shared__ __attribute__((aligned(16))) float sW[2 * E * E];   // [k][n]
    __shared__ float sB[2 * E], sNg[2 * E], sNb[2 * E];
    __shared__ __attribute__((aligned(16))) float ux[PPW * LDU];
    {   // all four arrays requested before the first store (common.h: lds_stage_ld / _st); absent LayerNorm vectors: a valid dummy source
        float vw[(2 * E * E + 255) / 256], vb[1], vg[1], vn[1];
        lds_stage_ld<256, 2 * E * E>(vw, a.w);
        lds_stage_ld<256, 2 * E>(vb, a.b);
        lds_stage_ld<256, 2 * E>(vg, a.g ? a.n1g : a.b);
        lds_stage_ld<256, 2 * E>(vn, a.g ? a.n1b : a.b);
        // the weight goes into LDS TRANSPOSED, sW[k][n]: lane q then reads its eight outputs' weights of one k as 32 consecutive bytes, the
        // LPP lanes of a pixel 32 bytes apart -- conflict-free.  Row-major (sW[n][k], lane q on rows 8q .. 8q+7) the lanes of a pixel were
        // 8 E dwords apart = on the SAME banks: SQ_LDS_BANK_CONFLICT was 3 x the kernel's LDS time at E = 32 (52 of its 68 us)
#pragma unroll
        for (int k = 0; k < (2 * E * E + 255) / 256; ++k) {
            const int i = k * 256 + threadIdx.x;
            if (i < 2 * E * E) sW[(i % E) * (2 * E) + i / E] = vw[k];
        }
        lds_stage_st<256, 2 * E>(sB, vb);
        lds_stage_st<256, 2 * E>(sNg, vg);
        lds_stage_st<256, 2 * E>(sNb, vn);
    }
    __syncthreads();
    const int q = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const int ho = a.H / 2, wo = a.W / 2;
    const long total = (long)a.B * ho * wo;
    const long p = (long)blockIdx.x * PPW + slot;
    const bool pv = p < total;
    const long pc_ = pv ? p : 0;
    const int ox = (int)(pc_ % wo);
    const long r = pc_ / wo;
    const int oy = (int)(r % ho);
    const long b = r / ho;
    int iy0, ix0;
    float wy[4], wx[4];
    resample_plan<0>(oy, iy0, wy);
    resample_plan<0>(ox, ix0, wx);
    float4 u4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* __restrict__ xin = a.x + 4 * q;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta) {
        const int yy = clampi(iy0 - 1 + ta, 0, a.H - 1);
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            const int xx = clampi(ix0 - 1 + tb, 0, a.W - 1);
            const float wgt = wy[ta] * wx[tb];
            const float4 v = *reinterpret_cast<const float4*>(xin + ((b * a.H + yy) * (long)a.W + xx) * E);
            u4.x += wgt * v.x; u4.y += wgt * v.y; u4.z += wgt * v.z; u4.w += wgt * v.w;
        }
    }
    if (a.u_save && pv) *reinterpret_cast<float4*>(a.u_save + p * E + 4 * q) = u4;
    *reinterpret_cast<float4*>(ux + slot * LDU + 4 * q) = u4;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();     // the lanes of a pixel sit in one wave
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float u[E];
#pragma unroll
    for (int k = 0; k < E / 4; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(ux + slot * LDU + 4 * k);
        u[4 * k] = v.x; u[4 * k + 1] = v.y; u[4 * k + 2] = v.z; u[4 * k + 3] = v.w;
    }
    float o[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) o[j] = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const float4 w0 = *reinterpret_cast<const float4*>(sW + k * (2 * E) + NO * q), w1 = *reinterpret_cast<const float4*>(sW + k * (2 * E) + NO * q + 4);
        o[0] += w0.x * u[k]; o[1] += w0.y * u[k]; o[2] += w0.z * u[k]; o[3] += w0.w * u[k];
        o[4] += w1.x * u[k]; o[5] += w1.y * u[k]; o[6] += w1.z * u[k]; o[7] += w1.w * u[k];
    }
#pragma unroll
    for (int j = 0; j < NO; ++j) o[j] += sB[NO * q + j];
    if (pv) {
        float4* yo = reinterpret_cast<float4*>(a.y + p * (2 * E) + NO * q);
        yo[0] = make_float4(o[0], o[1], o[2], o[3]);
        yo[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
    if (a.g) {   // LayerNorm over the 2E outputs of the pixel (its LPP lanes), planar global half = channels E .. 2E-1 = lanes q >= LPP/2
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NO; ++j) s += o[j];
        const float mu = lane_group_sum<LPP>(s) * (1.0f / (2 * E));
        float vs = 0.f;
#pragma unroll
        for (int j = 0; j < NO; ++j) { const float d = o[j] - mu; vs += d * d; }
        const float rstd = __builtin_amdgcn_rsqf(lane_group_sum<LPP>(vs) * (1.0f / (2 * E)) + LG_EPS);
        if (pv && q >= LPP / 2) {
            const long HWo = (long)ho * wo, sp = (long)oy * wo + ox;
#pragma unroll
            for (int j = 0; j < NO; ++j) {
                const int n = NO * q + j;
                a.g[(b * E + (n - E)) * HWo + sp] = (o[j] - mu) * rstd * sNg[n] + sNb[n];
            }
        }
    }
}

int launch_down(int E, const DownArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_DOWN, s);
    const long total = (long)a.B * (a.H / 2) * (a.W / 2);
    const int ppw = 256 / (E / 4);
    int grid = (int)((total + ppw - 1) / ppw);
    if (E == 16) k_down<16><<<grid, 256, 0, s>>>(a);
    else if (E == 32) k_down<32><<<grid, 256, 0, s>>>(a);
    else { lg_set_error("down: E=%d unsupported", E); return -1; }
    LG_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// decoder: bicubic x2 then 1x1 2E->E ; cat with skip ; fusion 1x1 2E->E    LGT.py:294-295,336-338
// ------------------------------------------------------------------------------------------------
// One workgroup = an 8 x 32 output tile.  The level-1 source pixels the tile's bicubic taps touch (8 x 20, border-clamped) are
// staged in LDS with coalesced 128-byte rows; the up-path 1x1 conv (2E -> E) is applied to those 160 source pixels BEFORE the
// resample (both are linear and the taps sum to 1, so conv(resample(x)) = resample(conv(x)) up to fp32 rounding; the bias is
// added after the resample): the conv runs on a quarter of the pixels and the 16-tap gather reads E channels from LDS instead of
// 2E channels from HBM with one address per lane (that gather was 280 scattered load instructions per wave, 126 us per call).
#ifdef LG_UPF_STAMPS   // diagnostic variant (bash tools/mkvariant.sh upf_stamps k_pixel.hip -DLG_UPF_STAMPS; tools/upf_stamps.py): s_memtime at the phase borders, every wave
__device__ unsigned long long g_upf_stamps[4096 * 4 * 12];
#define USTAMP(i) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)); \
                       g_upf_stamps[((blockIdx.x & 4095) * 4 + (threadIdx.x >> 6)) * 12 + (i)] = t__; } while (0)   /* branch-free: a conditional store splits the blocks and spills */
extern "C" __attribute__((visibility("default"))) int lg_debug_upf_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_upf_stamps), sizeof(g_upf_stamps));
}
#else
#define USTAMP(i) do { } while (0)
#endif
template <int E>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(E == 16 ? 4 : 2, E == 16 ? 4 : 2))) void k_upfuse(UpFuseArgs a, int tiles_x, int tiles_y) {   // e = 16: 38 KB of LDS and ~100 registers -- four workgroups per CU, some loading while others compute
    constexpr int TY = 8, TX = 32, SY = TY / 2 + 4, SX = TX / 2 + 4, NS = SY * SX /*160*/, LDV = E + 4, Q = 2 * E / 4;
    USTAMP(0);
    constexpr int QP = Q + 1;                                // padded pixel pitch in 16-byte units: 2E + 4 floats = 4 mod 64, so the 16 rows of an MFMA A-operand read
                                                             // (lane (r, g): pixel r of the tile, channel 4 ks + g) sit on 16 different banks
    __shared__ float4 srcb[NS * QP];                         // [NS][2E (+4)] level-1 pixels
    __shared__ __attribute__((aligned(16))) float vb[NS * LDV];   // [NS][E] up-conv of them (no bias)
    __shared__ float sFw[E * 2 * E], sFb[2 * E];   // fusion weight | up bias, fusion bias (synchronised by the barriers below)
    const int hi = a.H / 2, wi = a.W / 2;
    int t = blockIdx.x;
    const int tx_i = t % tiles_x;
    t /= tiles_x;
    const int ty_i = t % tiles_y;
    const long b = t / tiles_y;
    const int Y0 = ty_i * TY, X0 = tx_i * TX;
    const int sy0 = Y0 / 2 - 2, sx0 = X0 / 2 - 2;
    // The up-path 1x1 conv (2E -> E on the NS source pixels) runs on the matrix cores (round 6): out[px][n] = sum_k src[px][k] Wu[n][k] as
    // v_mfma_f32_16x16x4_f32 products (exact fp32 multiply-adds) with the source tile as the A operand straight out of LDS and the lane's slices of Wu in
    // registers as B fragments.  As 64-deep dot products per thread fed by two-address broadcast reads of the tile (16 ds_read_b128 per item) it was 40 % of
    // the workgroup's life at e = 32 (stamps: 25.7 k of 64.7 k cycles, tools/upf_stamps.py): every such read costs the LDS pipe 8 cycles for 32 useful bytes.
    constexpr int NMU = NS / 16, NTU = E / 16, KSU = 2 * E / 4, NT2 = NMU * NTU;     // pixel tiles, channel tiles, k-steps, (pixel tile, channel tile) units
    static_assert(NS % 16 == 0, "source tile = whole MFMA row tiles");
    const int ul_ = threadIdx.x & 63, uw_ = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), ur_ = ul_ & 15, ug_ = ul_ >> 4;
    float bu[NTU][KSU];
    constexpr int Q4 = E / 4, LDT = E + 4;
    float4 skc[Q4];
    bool cok[Q4];
    {   // every source piece of the thread is requested before the first one is stored: as a `for (i = tid; i < NS * Q; i += 256)` loop the compiler
        // kept one load + wait + store per trip -- ten dependent HBM round trips per workgroup at e = 32 (the kernel ran at 2.5 x its HBM time)
        constexpr int NPC = NS * Q / 256;
        static_assert(NS * Q % 256 == 0, "source tile = whole trips of the workgroup");
        // the skip rows of the tile (coalesced: thread t takes the float4 items t, t + 256, ... of the tile's rows): requested first, consumed behind the up-conv
        // and the resample.  (Measured both ways: requested behind the source tile -- so that the in-order counter lets the tile be waited for alone -- the workgroup lives
        // 51 k cycles, requested first 44.5 k: all 512 resident workgroups load at once, the phase is the chip's bandwidth, not a latency.)
#pragma unroll
        for (int k = 0; k < Q4; ++k) {
            const int i = threadIdx.x + 256 * k;
            const int px = i / Q4, row = px / TX, col = px - row * TX;
            cok[k] = Y0 + row < a.H && X0 + col < a.W;
            skc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cok[k]) skc[k] = reinterpret_cast<const float4*>(a.skip + ((b * a.H + Y0 + row) * (long)a.W + X0 + col) * E)[i - px * Q4];
        }
        float4 sv[NPC];
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            const int i = threadIdx.x + 256 * j;
            const int px = i / Q, k = i - px * Q;
            const int ly = px / SX, lx = px - ly * SX;
            const int yy = clampi(sy0 + ly, 0, hi - 1), xx = clampi(sx0 + lx, 0, wi - 1);
            sv[j] = reinterpret_cast<const float4*>(a.xb + ((b * hi + yy) * (long)wi + xx) * (2 * E))[k];
        }
        // the lane's B fragments of the up-conv weight (B[k = 4 ks + g][column r] = Wu[16 nt + r][4 ks + g]) and the three parameter arrays behind them, in the
        // same round trip (all requested before the first store)
#pragma unroll
        for (int nt = 0; nt < NTU; ++nt)
#pragma unroll
            for (int ks = 0; ks < KSU; ++ks) bu[nt][ks] = a.upw[(16 * nt + ur_) * 2 * E + 4 * ks + ug_];
        float vw[(E * 2 * E + 255) / 256], vu[1], vf[1];
        lds_stage_ld<256, E * 2 * E>(vw, a.fw);
        lds_stage_ld<256, E>(vu, a.upb);
        lds_stage_ld<256, E>(vf, a.fb);
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            const int i = threadIdx.x + 256 * j, px = i / Q;
            srcb[px * QP + (i - px * Q)] = sv[j];
        }
        lds_stage_st<256, E * 2 * E>(sFw, vw);
        lds_stage_st<256, E>(sFb, vu);
        lds_stage_st<256, E>(sFb + E, vf);
    }
    __syncthreads();
    USTAMP(1);
    {
        typedef float f32x4u __attribute__((ext_vector_type(4)));
        const float* srcf = reinterpret_cast<const float*>(srcb);
        // a unit's KSU A values are requested together, the next unit's behind a scheduling fence in front of this unit's MFMAs (left alone the compiler read
        // one value, waited, issued two MFMAs, read the next ...: 7.2 k cycles for 80 MFMAs per wave)
        constexpr int NU = (NT2 + 3) / 4;
        float av[2][KSU];
        auto a_ld = [&](int i, float (&v)[KSU]) {
            const int t2 = min(uw_ + 4 * i, NT2 - 1);        // (clamped: the last round of a wave without a unit re-reads a valid tile and stores nothing)
            const float* ap = srcf + (16 * (t2 / NTU) + ur_) * (4 * QP) + ug_;
#pragma unroll
            for (int ks = 0; ks < KSU; ++ks) v[ks] = ap[4 * ks];
        };
        a_ld(0, av[0]);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int t2 = uw_ + 4 * i;                      // wave-uniform: the units are dealt round-robin to the four waves
            if (i + 1 < NU) a_ld(i + 1, av[(i + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int t2c = min(t2, NT2 - 1), mt = t2c / NTU, nt = t2c - mt * NTU;
            f32x4u acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSU; ++ks) {
                float bsel = bu[0][ks];
#pragma unroll
                for (int q = 1; q < NTU; ++q) bsel = nt == q ? bu[q][ks] : bsel;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i & 1][ks], bsel, acc, 0, 0, 0);
            }
            if (t2 < NT2) {
#pragma unroll
                for (int v = 0; v < 4; ++v) vb[(16 * mt + 4 * ug_ + v) * LDV + 16 * nt + ur_] = acc[v];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    USTAMP(2);
    const int ly = threadIdx.x / TX, lx = threadIdx.x - ly * TX;
    const int oy = Y0 + ly, ox = X0 + lx;
    const bool inimg = oy < a.H && ox < a.W;
    const long p = (b * a.H + min(oy, a.H - 1)) * (long)a.W + min(ox, a.W - 1);
    // NHWC rows move COALESCED: thread t takes the float4 items t, t + 256, ... of the tile's rows (one tile row = TX pixels = TX * E
    // contiguous floats) and the per-pixel vectors are exchanged through LDS (a thread reading / writing its own 64-byte pixel vector
    // touches 64 scattered 16-byte pieces per wave-instruction).  The skip rows are requested here, early: their latency hides under
    // the resample; they go to LDS once the source-pixel buffer (srcb) is dead.
    float* stg = reinterpret_cast<float*>(srcb);          // [256][LDT] exchange buffer (srcb is dead after the up-conv above)
    static_assert(sizeof(srcb) >= 256 * LDT * sizeof(float), "exchange buffer fits the source-pixel buffer");
    int iy0, ix0;
    float wy[4], wx[4];
    resample_plan<1>(min(oy, a.H - 1), iy0, wy);
    resample_plan<1>(min(ox, a.W - 1), ix0, wx);
    float tt[E];
#pragma unroll
    for (int n = 0; n < E; ++n) tt[n] = 0.f;
    {   // the 16 taps, double-buffered by hand: the next tap's E / 4 vectors are requested before this tap's multiply-adds and a scheduling fence keeps them
        // there.  Left to itself the compiler read every vector into ONE register quad -- read, wait, four multiply-adds, read, wait ... : 128 dependent LDS
        // round trips per thread, 16.3 k of the workgroup's 54 k cycles at e = 32 (tools/upf_stamps.py)
        constexpr int Q4t = E / 4;
        float4 tv[2][Q4t];
        const int base_y = iy0 - 1 - sy0, base_x = ix0 - 1 - sx0;   // local row / column of tap (0, 0) (the LDS tile already holds border-clamped pixels)
        auto tap_ld = [&](int tp, float4 (&v)[Q4t]) {
            const float4* vr = reinterpret_cast<const float4*>(vb + ((base_y + (tp >> 2)) * SX + base_x + (tp & 3)) * LDV);
#pragma unroll
            for (int k = 0; k < Q4t; ++k) v[k] = vr[k];
        };
        tap_ld(0, tv[0]);
#pragma unroll
        for (int tp = 0; tp < 16; ++tp) {
            if (tp + 1 < 16) tap_ld(tp + 1, tv[(tp + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const float wgt = wy[tp >> 2] * wx[tp & 3];
#pragma unroll
            for (int k = 0; k < Q4t; ++k) {
                const float4 v = tv[tp & 1][k];
                tt[4 * k] += wgt * v.x; tt[4 * k + 1] += wgt * v.y; tt[4 * k + 2] += wgt * v.z; tt[4 * k + 3] += wgt * v.w;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int n = 0; n < E; ++n) tt[n] += sFb[n];
    USTAMP(3);
    // ---- fusion 1x1 conv (2E -> E) on the matrix cores: out[px][n] = bf[n] + sum_k Wf[n][E + k] skip[px][k] + sum_k Wf[n][k] t[px][k].
    // The pixel rows pass through the exchange buffer anyway (skip comes in coalesced, t / y go out coalesced), so they ARE the A operand
    // of v_mfma_f32_16x16x4_f32 (row pitch E + 4: conflict-free ds_read_b32), the weights sit in registers as B fragments and wave w
    // owns pixels [64 w, 64 w + 64).  As 512 (e = 16) / 2048 (e = 32) FMAs per thread fed by LDS weight broadcasts this was the LDS-bound
    // bulk of the kernel.
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    constexpr int NTL = E / 16, KS = E / 4;
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6, r_ = lane_ & 15, g_ = lane_ >> 4;
    float bw[2][NTL][KS];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bw[hf][nt][ks] = sFw[(nt * 16 + r_) * 2 * E + hf * E + 4 * ks + g_];
    f32x4_ acc[4][NTL];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) { const float bz = sFb[E + nt * 16 + r_]; acc[mt][nt] = (f32x4_){bz, bz, bz, bz}; }
    auto gemm_half = [&](int hf) {   // a row tile's KS A values requested together, the next tile's in front of this tile's MFMAs
        float ga[2][KS];
        auto g_ld = [&](int mt, float (&v)[KS]) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) v[ks] = stg[(64 * wave_ + 16 * mt + r_) * LDT + 4 * ks + g_];
        };
        g_ld(0, ga[0]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (mt + 1 < 4) g_ld(mt + 1, ga[(mt + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[mt & 1][ks], bw[hf][nt][ks], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto rows_out = [&](float* dst) {   // coalesced store of the tile's rows from the exchange buffer
#pragma unroll
        for (int k = 0; k < Q4; ++k) {
            const int i = threadIdx.x + 256 * k;
            const int px = i / Q4, row = px / TX, col = px - row * TX;
            if (cok[k]) reinterpret_cast<float4*>(dst + ((b * a.H + Y0 + row) * (long)a.W + X0 + col) * E)[i - px * Q4] =
                            *reinterpret_cast<const float4*>(stg + px * LDT + 4 * (i - px * Q4));
        }
    };
    // skip rows: registers -> exchange buffer
#pragma unroll
    for (int k = 0; k < Q4; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int px = i / Q4;
        *reinterpret_cast<float4*>(stg + px * LDT + 4 * (i - px * Q4)) = skc[k];
    }
    __syncthreads();
    USTAMP(4);
    gemm_half(1);
    USTAMP(5);
    __syncthreads();
    // t rows (up path): this thread's pixel vector -> exchange buffer
#pragma unroll
    for (int k = 0; k < Q4; ++k) *reinterpret_cast<float4*>(stg + threadIdx.x * LDT + 4 * k) = make_float4(tt[4 * k], tt[4 * k + 1], tt[4 * k + 2], tt[4 * k + 3]);
    __syncthreads();
    USTAMP(6);
    if (a.t_save) rows_out(a.t_save);
    gemm_half(0);
    USTAMP(7);
    __syncthreads();
    // output rows: accumulator layout (lane (r, g): pixels 4g + v, channel r) -> exchange buffer
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int v = 0; v < 4; ++v) stg[(64 * wave_ + 16 * mt + 4 * g_ + v) * LDT + nt * 16 + r_] = acc[mt][nt][v];
    __syncthreads();
    USTAMP(8);
    rows_out(a.y);
    USTAMP(9);
    if (a.g && inimg) {
        float o[E];
#pragma unroll
        for (int k = 0; k < Q4; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(stg + threadIdx.x * LDT + 4 * k);
            o[4 * k] = v.x; o[4 * k + 1] = v.y; o[4 * k + 2] = v.z; o[4 * k + 3] = v.w;
        }
        emit_g<E>(o, a.n1g, a.n1b, a.g, b, (long)oy * a.W + ox, (long)a.H * a.W);
    }
    USTAMP(10);
}

int launch_upfuse(int E, const UpFuseArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_UPFUSE, s);
    if ((a.H & 1) || (a.W & 1)) { lg_set_error("upfuse: H, W must be even"); return -2; }
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + 7) / 8;
    const int grid = a.B * tiles_x * tiles_y;
    if (E == 16) k_upfuse<16><<<grid, 256, 0, s>>>(a, tiles_x, tiles_y);
    else if (E == 32) k_upfuse<32><<<grid, 256, 0, s>>>(a, tiles_x, tiles_y);
    else { lg_set_error("upfuse: E=%d unsupported", E); return -1; }
    LG_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// reconstruction tail: (identity resample) 1x1 E->C, + x                    LGT.py:302-303,342
// ------------------------------------------------------------------------------------------------
template <int C, int E>
__global__ __launch_bounds__(256) void k_tail(TailArgs a) {
    __shared__ float sW[C * E], sB[C];
    // EVERY load of the workgroup's life is requested before its first wait: the pixel's row and its C values of z together with the
    // parameter arrays.  As staging -> barrier -> row -> matvec -> z the kernel was three dependent round trips long (a workgroup lives for
    // one pixel per thread), 12.0 us for 50 MB
    const long pr = blockIdx.x * 256L + threadIdx.x;
    const long p = pr < a.total ? pr : a.total - 1;
    const long b = p / a.HW, s = p - b * a.HW;
    float4 xv[E / 4];
    float zv[C];
    {
        const float4* src = reinterpret_cast<const float4*>(a.x + p * E);
#pragma unroll
        for (int k = 0; k < E / 4; ++k) xv[k] = src[k];
#pragma unroll
        for (int c = 0; c < C; ++c) zv[c] = a.z[(b * C + c) * a.HW + s];
        float vw[(C * E + 255) / 256], vb[1];
        lds_stage_ld<256, C * E>(vw, a.w);
        lds_stage_ld<256, C>(vb, a.b);
        lds_stage_st<256, C * E>(sW, vw);
        lds_stage_st<256, C>(sB, vb);
    }
    __syncthreads();
    if (pr >= a.total) return;
    float x[E];
#pragma unroll
    for (int k = 0; k < E / 4; ++k) { x[4 * k] = xv[k].x; x[4 * k + 1] = xv[k].y; x[4 * k + 2] = xv[k].z; x[4 * k + 3] = xv[k].w; }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < E; ++k) v += sW[c * E + k] * x[k];
        a.out[(b * C + c) * a.HW + s] = v + sB[c] + zv[c];
    }
}

int launch_tail(int C, const TailArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_TAIL, s);
    int grid = (int)((a.total + 255) / 256);
    if (C == 4) k_tail<4, 16><<<grid, 256, 0, s>>>(a);
    else if (C == 8) k_tail<8, 32><<<grid, 256, 0, s>>>(a);
    else { lg_set_error("tail: C=%d unsupported", C); return -1; }
    LG_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// stand-alone LN1 + planar split (per-op test entry only; in the net it is an epilogue of the producer)
// ------------------------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(256) void k_ln_split(const float* __restrict__ x, const float* __restrict__ n1g,
                                                  const float* __restrict__ n1b, float* __restrict__ g, long HW, long total) {
    long p = blockIdx.x * 256L + threadIdx.x;
    if (p >= total) return;
    long b = p / HW, s = p - b * HW;
    float v[E];
    const float4* src = reinterpret_cast<const float4*>(x + p * E);
#pragma unroll
    for (int k = 0; k < E / 4; ++k) {
        float4 t = src[k];
        v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
    }
    emit_g<E>(v, n1g, n1b, g, b, s, HW);
}

int launch_ln_split(int e, const float* x, const float* n1g, const float* n1b, float* g, int B, int HW, hipStream_t s) {
    long total = (long)B * HW;
    int grid = (int)((total + 255) / 256);
    if (e == 16) k_ln_split<16><<<grid, 256, 0, s>>>(x, n1g, n1b, g, HW, total);
    else if (e == 32) k_ln_split<32><<<grid, 256, 0, s>>>(x, n1g, n1b, g, HW, total);
    else if (e == 64) k_ln_split<64><<<grid, 256, 0, s>>>(x, n1g, n1b, g, HW, total);
    else { lg_set_error("ln_split: e=%d unsupported", e); return -1; }
    LG_CHECK_LAUNCH();
    return 0;
}
