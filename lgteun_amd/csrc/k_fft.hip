// Global (FFT amplitude/phase) mixer for gfx950 -- reference models/common/LGT.py:149-180.
//
// One workgroup owns one (sample, channel) plane n x n and keeps it in LDS as a complex n x n image
// (n = 128: 128 KiB of the CU's 160 KiB).  rfft2 is restated as fft_H(rfft_W(x)) and irfft2 as
// irfft_W(ifft_H(X)): forward radix-2 DIF leaves bins in bit-reversed positions, the
// amplitude/phase edit is pointwise so it does not care, and the inverse radix-2 DIT consumes bit-reversed
// input -- no reordering pass.  Only columns kx <= n/2 go through the column transforms (half spectrum);
// rows are Hermitian-extended before the last (row) inverse, with Im of the kx = 0 and kx = n/2 columns
// dropped exactly as a c2r transform does.  The four purely-real bins get +0.0 imaginary parts so that
// angle() takes the same branch as pocketfft's r2c (+pi for negative DC).  fp32 throughout.
#include "kernels.h"
#include "bwd_kernels.h"

__device__ __forceinline__ float2 cmul(float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
__device__ __forceinline__ float2 cmulc(float2 a, float2 w) { return make_float2(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y); }

// batched in-place radix-2 butterflies over LDS.  `lines` transforms of length n; element i of line l is at
// buf[l * ls + i * es].  Lines for which skip(l) is true are left untouched.
// S fused radix-2 stages in one LDS pass: a thread loads 2^S points, runs the S butterfly levels in registers and writes
// them back (same data flow as S consecutive radix-2 passes -> same bit-reversed positions), so the plane crosses LDS
// 3 times per 1-D transform (7 = 3 + 3 + 1 stages at n = 128) instead of 7.
template <bool INVERSE, bool COLS, int S>
__device__ __forceinline__ void fft_fused(float2* buf, const float2* tw, int n, int lg, int st) {
    constexpr int R = 1 << S;
    const int half = n >> 1;
    // spans of the fused levels: forward (DIF) largest first, inverse (DIT) smallest first; mL = smallest span
    const int lgmL = INVERSE ? st : (lg - st - S);
    const int mL = 1 << lgmL;
    const int per_line = n >> S;                 // items per line
    const int lgpl = lg - S;
    const int items = n << lgpl;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        int line, t;
        if (COLS) { t = it >> lg; line = it & (n - 1); } else { line = it >> lgpl; t = it & (per_line - 1); }
        if (COLS) {
            const int kx = (int)(__brev((unsigned)line) >> (32 - lg));
            if (kx > half) continue;
        }
        const int lo = t & (mL - 1), hi = t >> lgmL;
        const int i_base = (hi << (lgmL + S)) + lo;
        float2 v[R];
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const int i = i_base + (c << lgmL);
            v[c] = buf[COLS ? ((i << lg) + line) : ((line << lg) + i)];
        }
#pragma unroll
        for (int k = 0; k < S; ++k) {
            // level k pairs (c, c + d); span of this level m = mL * d
            const int dsh = INVERSE ? k : (S - 1 - k);
            const int d = 1 << dsh;
            const int twshift = lg - 1 - (lgmL + dsh);
#pragma unroll
            for (int c = 0; c < R; ++c) {
                if (c & d) continue;
                const int j = lo + ((c & (d - 1)) << lgmL);
                const float2 w = tw[j << twshift];
                const float2 a = v[c], b = v[c + d];
                if (!INVERSE) {
                    const float2 df = make_float2(a.x - b.x, a.y - b.y);
                    v[c] = make_float2(a.x + b.x, a.y + b.y);
                    v[c + d] = cmul(df, w);
                } else {
                    const float2 bw = cmulc(b, w);
                    v[c] = make_float2(a.x + bw.x, a.y + bw.y);
                    v[c + d] = make_float2(a.x - bw.x, a.y - bw.y);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const int i = i_base + (c << lgmL);
            buf[COLS ? ((i << lg) + line) : ((line << lg) + i)] = v[c];
        }
    }
    __syncthreads();
}

// full 1-D transform of every line: stages grouped 3 + 3 + ... + remainder
template <bool INVERSE, bool COLS>
__device__ __forceinline__ void fft_pass(float2* buf, const float2* tw, int n, int lg) {
    int st = 0;
    while (lg - st >= 3) { fft_fused<INVERSE, COLS, 3>(buf, tw, n, lg, st); st += 3; }
    if (lg - st == 2) fft_fused<INVERSE, COLS, 2>(buf, tw, n, lg, st);
    else if (lg - st == 1) fft_fused<INVERSE, COLS, 1>(buf, tw, n, lg, st);
}

// rows: drop Im of the kx = 0 and kx = n/2 columns and Hermitian-extend (what a c2r transform assumes)
__device__ __forceinline__ void hermitian_extend(float2* buf, int n, int lg) {
    const int half = n >> 1;
    for (int it = threadIdx.x; it < n * (half + 1); it += blockDim.x) {
        int y = it / (half + 1), kx = it - y * (half + 1);
        int p = (int)(__brev((unsigned)kx) >> (32 - lg));
        if (kx == 0 || kx == half) {
            buf[y * n + p].y = 0.0f;
        } else {
            int pm = (int)(__brev((unsigned)(n - kx)) >> (32 - lg));
            float2 v = buf[y * n + p];
            buf[y * n + pm] = make_float2(v.x, -v.y);
        }
    }
    __syncthreads();
}

__global__ void k_fftmix(FftArgs a, int lg) {
    extern __shared__ float2 smem2[];
    const int n = a.n, half = n >> 1;
    float2* buf = smem2;          // [n][n]
    float2* tw = smem2 + n * n;   // [n/2]  exp(-2 pi i k / n)
    const int plane = blockIdx.x;
    const int ch = plane % a.ch;
    const float* g = a.g + (size_t)plane * n * n;
    for (int k = threadIdx.x; k < half; k += blockDim.x) {
        float ang = 2.0f * (float)k / (float)n;
        tw[k] = make_float2(cospif(ang), 0.0f - sinpif(ang));
    }
    for (int i = threadIdx.x; i < n * n; i += blockDim.x) buf[i] = make_float2(g[i], 0.0f);
    __syncthreads();
    // ---- rfft2: rows then columns
    fft_pass<false, false>(buf, tw, n, lg);
    for (int y = threadIdx.x; y < n; y += blockDim.x) { buf[y * n + 0].y = 0.0f; buf[y * n + 1].y = 0.0f; }  // kx = 0, n/2 are real
    __syncthreads();
    fft_pass<false, true>(buf, tw, n, lg);
    if (threadIdx.x < 4) buf[(threadIdx.x >> 1) * n + (threadIdx.x & 1)].y = 0.0f;  // the four purely-real bins
    __syncthreads();
    // ---- amplitude / phase edit (LGT.py:168-177)
    const float aw = a.ampw[ch], ab = a.ampb[ch], pw = a.phaw[ch], pb = a.phab[ch];
    for (int it = threadIdx.x; it < n * n; it += blockDim.x) {
        int q = it >> lg, p = it & (n - 1);
        int kx = (int)(__brev((unsigned)p) >> (32 - lg));
        if (kx > half) continue;
        float2 f = buf[it];
        float amp = hypotf(f.x, f.y);
        float pha = atan2f(f.y, f.x);
        if (a.amp) {
            int ky = (int)(__brev((unsigned)q) >> (32 - lg));
            size_t o = ((size_t)plane * n + ky) * (half + 1) + kx;
            a.amp[o] = amp;
            a.pha[o] = pha;
        }
        float am = aw * amp + ab;
        float ph = pw * pha + pb;
        float sn, cs;
        sincosf(ph, &sn, &cs);
        float re = (am * cs + 1e-8f) + 1e-8f;
        float im = am * sn + 1e-8f;
        buf[it] = make_float2(re, im);
    }
    __syncthreads();
    // ---- irfft2: columns (complex), Hermitian extension, rows
    fft_pass<true, true>(buf, tw, n, lg);
    hermitian_extend(buf, n, lg);
    fft_pass<true, false>(buf, tw, n, lg);
    const float sc = 1.0f / ((float)n * (float)n);
    float* o = a.o + (size_t)plane * n * n;
    for (int i = threadIdx.x; i < n * n; i += blockDim.x) {
        const float v = buf[i].x * sc;
        o[i] = fabsf(v);
        if (a.sgn) a.sgn[(size_t)plane * n * n + i] = (v > 0.f) ? 1.0f : ((v < 0.f) ? -1.0f : 0.0f);
    }
}

int launch_fftmix(const FftArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_FFT, s);
    int n = a.n, lg = 0;
    while ((1 << lg) < n) ++lg;
    if ((1 << lg) != n || n < 8 || n > 128) { lg_set_error("fftmix: plane size %d unsupported (power of two, 8..128)", n); return -2; }
    size_t lds = ((size_t)n * n + n / 2) * sizeof(float2);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_fftmix, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) { lg_set_error("fftmix: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done = true;
    }
    int threads = n >= 128 ? 1024 : (n >= 64 ? 512 : 256);
    k_fftmix<<<a.planes, threads, lds, s>>>(a, lg);
    LG_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// backward (autograd of LGT.py:149-180):  dt = do * sign(t);  dY = (c_kx / n^2) rfft2(dt)  [adjoint of the c2r irfft2];
// through re/im = amp' (cos,sin)(pha') and amp' = aw*amp + ab, pha' = pw*pha + pb (parameter gradients);
// through amp = |F|, pha = angle(F);  dg = Re sum_{kx<=n/2} dF e^{+i theta}  [adjoint of the r2c rfft2], which is the
// same c2r machinery applied to dF * n^2 / c_kx.
// ------------------------------------------------------------------------------------------------
__global__ void k_fftmix_bwd(FftBwdArgs a, int lg) {
    extern __shared__ float2 smem2[];
    const int n = a.n, half = n >> 1;
    float2* buf = smem2;
    float2* tw = smem2 + n * n;
    float* red = reinterpret_cast<float*>(tw + half);  // [16][4]
    const int plane = blockIdx.x;
    const int ch = plane % a.ch;
    const size_t base = (size_t)plane * n * n;
    for (int k = threadIdx.x; k < half; k += blockDim.x) {
        float ang = 2.0f * (float)k / (float)n;
        tw[k] = make_float2(cospif(ang), 0.0f - sinpif(ang));
    }
    for (int i = threadIdx.x; i < n * n; i += blockDim.x) buf[i] = make_float2(a.do2[base + i] * a.sgn[base + i], 0.0f);
    __syncthreads();
    fft_pass<false, false>(buf, tw, n, lg);
    fft_pass<false, true>(buf, tw, n, lg);
    const float aw = a.ampw[ch], ab = a.ampb[ch], pw = a.phaw[ch], pb = a.phab[ch];
    const float nn = (float)n * (float)n;
    float s_aw = 0.f, s_ab = 0.f, s_pw = 0.f, s_pb = 0.f;
    for (int it = threadIdx.x; it < n * n; it += blockDim.x) {
        int q = it >> lg, p = it & (n - 1);
        int kx = (int)(__brev((unsigned)p) >> (32 - lg));
        if (kx > half) continue;
        int ky = (int)(__brev((unsigned)q) >> (32 - lg));
        const float c = (kx == 0 || kx == half) ? 1.0f : 2.0f;
        const float2 f = buf[it];
        const float dR = f.x * (c / nn), dI = f.y * (c / nn);
        const size_t o = ((size_t)plane * n + ky) * (half + 1) + kx;
        const float A = a.amp[o], PH = a.pha[o];
        const float Am = aw * A + ab, Ph = pw * PH + pb;
        float sn, cs;
        sincosf(Ph, &sn, &cs);
        const float dAm = dR * cs + dI * sn;
        const float dPh = Am * (dI * cs - dR * sn);
        s_aw += dAm * A; s_ab += dAm; s_pw += dPh * PH; s_pb += dPh;
        const float dA = aw * dAm, dPH = pw * dPh;
        float sn0, cs0;
        sincosf(PH, &sn0, &cs0);
        float dFr = 0.f, dFi = 0.f;
        if (A > 0.f) {
            const float ia = 1.0f / A;
            dFr = dA * cs0 - dPH * sn0 * ia;
            dFi = dA * sn0 + dPH * cs0 * ia;
        }
        const float k2 = nn / c;
        buf[it] = make_float2(dFr * k2, dFi * k2);
    }
    __syncthreads();
    fft_pass<true, true>(buf, tw, n, lg);
    hermitian_extend(buf, n, lg);
    fft_pass<true, false>(buf, tw, n, lg);
    const float sc = 1.0f / nn;
    for (int i = threadIdx.x; i < n * n; i += blockDim.x) a.dg[base + i] = buf[i].x * sc;
    // parameter gradient partials
    float v[4] = {s_aw, s_ab, s_pw, s_pb};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_xor(v[i], off);
        if (lane == 0) red[wave * 4 + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += red[w * 4 + threadIdx.x];
        float* dst = threadIdx.x == 0 ? a.d_ampw : (threadIdx.x == 1 ? a.d_ampb : (threadIdx.x == 2 ? a.d_phaw : a.d_phab));
        atomicAdd(dst + ch, s);
    }
}

int launch_fftmix_bwd(const FftBwdArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_FFT_BWD, s);
    int n = a.n, lg = 0;
    while ((1 << lg) < n) ++lg;
    if ((1 << lg) != n || n < 8 || n > 128) { lg_set_error("fftmix_bwd: plane size %d unsupported", n); return -2; }
    size_t lds = ((size_t)n * n + n / 2) * sizeof(float2) + 64 * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_fftmix_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        if (e != hipSuccess) { lg_set_error("fftmix_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done = true;
    }
    int threads = n >= 128 ? 1024 : (n >= 64 ? 512 : 256);
    k_fftmix_bwd<<<a.planes, threads, lds, s>>>(a, lg);
    LG_CHECK_LAUNCH();
    return 0;
}
