// The 34 x 34 halo tile (a 32 x 32 output tile + one ring; 0 outside the plane) of a bicubic-resampled plane, evaluated SEPARABLY through
// LDS: the source window is staged once with clamped coordinates, contracted along x with each column's four taps, then along y.  Same
// arithmetic in the same order as resample_at (common.h: the four row sums first, then their combination), so the values are the same;
// 8 LDS reads per element instead of 16 clamped global gathers with their index arithmetic (the first form of the data-step kernels spent
// most of their time there).  MODE as in resample_plan: 0 = x0.5, 1 = x2.  (bmu.sampling_unit_, basic_module_unformer_v2.py:26-34)
#pragma once
#include "common.h"

template <int MODE>
struct RsTile {
    static constexpr int SR = MODE == 1 ? 20 : 70;   // source rows / columns that feed 34 consecutive outputs
    static constexpr int SP = SR + 1;                // row pitch of the staged window
    static constexpr int FLOATS = SR * SP + SR * 35; // scratch: window | row-contracted [SR][35]
};

// U[uy][ux] = resample<MODE>(in)(ty0 - 1 + uy, tx0 - 1 + ux) for 0 <= uy, ux < 34, 0 outside [0, ho) x [0, wo).  All 256 threads call;
// `scratch` holds RsTile<MODE>::FLOATS floats; U is complete after the caller's next __syncthreads().
template <int MODE>
__device__ __forceinline__ void resample_tile34(const float* __restrict__ in, int hi, int wi, int ho, int wo, int ty0, int tx0,
                                                float (*U)[35], float* scratch) {
    constexpr int SR = RsTile<MODE>::SR, SP = RsTile<MODE>::SP;
    float* S = scratch;
    float* Hh = scratch + SR * SP;
    const int sy0 = MODE == 1 ? (ty0 >> 1) - 2 : 2 * ty0 - 3;   // pre-clamp source coordinate of S[0][0]
    const int sx0 = MODE == 1 ? (tx0 >> 1) - 2 : 2 * tx0 - 3;
    {   // the window in rounds of up to four values per thread, every value of a round requested before its first store (one load + wait + store per
        // trip is one dependent round trip per trip)
        constexpr int NTR = (SR * SR + 255) / 256, RND = NTR < 4 ? NTR : 4;
#pragma unroll 1
        for (int t0 = 0; t0 < NTR; t0 += RND) {
            float v[RND];
#pragma unroll
            for (int j = 0; j < RND; ++j) {
                const int i = min((t0 + j) * 256 + (int)threadIdx.x, SR * SR - 1);
                const int sy = i / SR, sx = i - sy * SR;
                v[j] = in[(size_t)clampi(sy0 + sy, 0, hi - 1) * wi + clampi(sx0 + sx, 0, wi - 1)];
            }
#pragma unroll
            for (int j = 0; j < RND; ++j) asm volatile("" : "+v"(v[j]));   // (or a partial trip's load sinks into its store's `if`)
#pragma unroll
            for (int j = 0; j < RND; ++j) {
                const int i = (t0 + j) * 256 + (int)threadIdx.x;
                const int sy = i / SR, sx = i - sy * SR;
                if (i < SR * SR) S[sy * SP + sx] = v[j];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SR * 34; i += 256) {
        const int sy = i / 34, ux = i - sy * 34;
        int i0;
        float w[4];
        resample_plan<MODE>(tx0 - 1 + ux, i0, w);
        const float* p = S + sy * SP + (i0 - 1 - sx0);
        float r = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) r += w[b] * p[b];
        Hh[sy * 35 + ux] = r;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 34 * 34; i += 256) {
        const int uy = i / 34, ux = i - uy * 34;
        const int oy = ty0 - 1 + uy, ox = tx0 - 1 + ux;
        float acc = 0.f;
        if (oy >= 0 && oy < ho && ox >= 0 && ox < wo) {
            int i0;
            float w[4];
            resample_plan<MODE>(oy, i0, w);
            const float* p = Hh + (i0 - 1 - sy0) * 35 + ux;
#pragma unroll
            for (int a = 0; a < 4; ++a) acc += w[a] * p[a * 35];
        }
        U[uy][ux] = acc;
    }
}

// adjoint coefficients of the 1-D resampler: input index i (size n_in) receives coef[a] * gout[base + a]
template <int MODE>
struct AdjPlan {
    static constexpr int NC = (MODE == 0) ? 3 : 10;
    int base;
    float coef[NC];
    __device__ __forceinline__ void make(int i, int n_in, int n_out) {
        base = (MODE == 0) ? (i / 2 - 1) : (2 * i - 4);
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            const int o = base + a;
            float c = 0.f;
            if (o >= 0 && o < n_out) {
                int i0;
                float w[4];
                resample_plan<MODE>(o, i0, w);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (clampi(i0 - 1 + t, 0, n_in - 1) == i) c += w[t];
            }
            coef[a] = c;
        }
    }
};

