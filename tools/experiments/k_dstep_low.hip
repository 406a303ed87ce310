// The low-resolution middle of the data-step backward as ONE launch (autograd of reference models/unlg_former.py:29-33, the D / DT
// chains of bmu.sampling_unit_ + bmu.dep_conv; basic_module_unformer_v2.py:17-18,26-34):
//
//   gs1 (grad wrt s1 = dw_DT1(up(r)))  ->  gu1 = dw_DT1^T gs1  ->  gr = up^T gu1  ->  gd3 = dw_D3^T gr  ->  gt1 = down^T gd3
//                                      ->  gd1 = dw_D1^T gt1   (+ the three depthwise weight / bias gradients)
//
// These were five launches of 6 - 12 us each on tensors of 0.5 - 2 MB -- pure launch latency, 4 stages per step.  Here one workgroup
// owns one (sample, channel) plane and keeps every intermediate in LDS (half-resolution planes with a zero ring for the 3x3 taps,
// quarter-resolution ones likewise); resamplers and their adjoints are evaluated separably (rows, then columns) exactly as the
// stand-alone kernels do.  Planes up to 128 x 128 PAN fit; larger ones keep the five-launch chain (k_bwd.hip).
#include "kernels.h"
#include "bwd_kernels.h"
#include "resample_tile.h"

namespace {

constexpr int NT = 1024;   // threads per plane: the phases are short and serial, so the plane gets the whole CU

// sum v[0..N) over the NT threads of the block; result valid in threads 0..N-1
template <int N>
__device__ __forceinline__ float block_sum_n(float (&v)[N], float* sm /* [NT/64 * N] */) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float s = v[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) sm[wave * N + i] = s;
    }
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x < N) {
#pragma unroll
        for (int wv = 0; wv < NT / 64; ++wv) r += sm[wv * N + threadIdx.x];
    }
    __syncthreads();
    return r;
}

// plan tables in LDS: [n][NC + 1] = base (int bits), coefficients
template <int MODE>
__device__ __forceinline__ void build_adj_plans(float* tab, int n_in, int n_out) {
    constexpr int NC = AdjPlan<MODE>::NC, PL = NC + 1;
    for (int i = threadIdx.x; i < n_in; i += NT) {
        AdjPlan<MODE> pl;
        pl.make(i, n_in, n_out);
        tab[i * PL] = __int_as_float(pl.base);
#pragma unroll
        for (int a = 0; a < NC; ++a) tab[i * PL + 1 + a] = pl.coef[a];
    }
}

// dst interior (pitch pd, ring 1) [no x wo] = resample<MODE>(src [ni x wi], pitch ps, offset so), rows then columns through tmp [ni][wo]
template <int MODE, typename Src>
__device__ __forceinline__ void resample_plane(Src src, int ni, int wi, float* tmp, float* dst, int pd, int no, int wo) {
    // four outputs per thread and trip with all sixteen source reads issued first (from global memory they are latency, not bandwidth)
    for (int i0_ = threadIdx.x; i0_ < ni * wo; i0_ += 4 * NT) {
        float v[4][4], w[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0_ + u * NT, ni * wo - 1);
            const int y = i / wo, ox = i - y * wo;
            int i0;
            resample_plan<MODE>(ox, i0, w[u]);
#pragma unroll
            for (int b = 0; b < 4; ++b) v[u][b] = src(y, clampi(i0 - 1 + b, 0, wi - 1));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float r = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) r += w[u][b] * v[u][b];
            if (i0_ + u * NT < ni * wo) tmp[i0_ + u * NT] = r;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < no * wo; i += NT) {
        const int oy = i / wo, ox = i - oy * wo;
        int i0;
        float w[4];
        resample_plan<MODE>(oy, i0, w);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) acc += w[a] * tmp[clampi(i0 - 1 + a, 0, ni - 1) * wo + ox];
        dst[(oy + 1) * pd + ox + 1] = acc;
    }
    __syncthreads();
}

// dst interior (pitch pd) [ni x wi] = R<MODE>^T src interior (pitch ps) [no x wo]: along x into tmp [no][wi], then along y
template <int MODE>
__device__ __forceinline__ void adjoint_plane(const float* src, int ps, int no, int wo, float* tmp, float* dst, int pd, int ni, int wi,
                                              const float* tabx, const float* taby) {
    constexpr int NC = AdjPlan<MODE>::NC, PL = NC + 1;
    for (int i = threadIdx.x; i < no * wi; i += NT) {
        const int y = i / wi, ix = i - y * wi;
        const float* pp = tabx + ix * PL;
        const int bx = __float_as_int(pp[0]);
        const float* row = src + (y + 1) * ps + 1;
        float acc = 0.f;
#pragma unroll
        for (int b = 0; b < NC; ++b) acc += pp[1 + b] * row[clampi(bx + b, 0, wo - 1)];   // clamped taps carry coefficient 0
        tmp[i] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ni * wi; i += NT) {
        const int iy = i / wi, ix = i - iy * wi;
        const float* pp = taby + iy * PL;
        const int by = __float_as_int(pp[0]);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < NC; ++a) acc += pp[1 + a] * tmp[clampi(by + a, 0, no - 1) * wi + ix];
        dst[(iy + 1) * pd + ix + 1] = acc;
    }
    __syncthreads();
}

// partial sums of one depthwise conv's gradients over the plane: part[tap] = sum gout(y,x) * in(y+dy-1, x+dx-1), part[9] = sum gout
__device__ __forceinline__ void dw_grad_partials(const float* g, const float* u, int p, int n, int w, float (&part)[10]) {
#pragma unroll
    for (int k = 0; k < 10; ++k) part[k] = 0.f;
    for (int i = threadIdx.x; i < n * w; i += NT) {
        const int y = i / w, x = i - y * w;
        const float gc = g[(y + 1) * p + x + 1];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) part[dy * 3 + dx] += gc * u[(y + dy) * p + x + dx];
        part[9] += gc;
    }
}
// (dw^T g)(y, x) for the 3x3 depthwise conv with zero padding: g has a zero ring
__device__ __forceinline__ float dw_transpose_at(const float* g, int p, int y, int x, const float (&w)[9]) {
    float gi = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) gi += w[dy * 3 + dx] * g[(y + 2 - dy) * p + x + 2 - dx];
    return gi;
}

__global__ __launch_bounds__(NT) void k_dstep_low_bwd(DstepLowBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[NT / 64 * 10];
    const int H = a.H, W = a.W, hh = H / 2, wh = W / 2, hq = H / 4, wq = W / 4;
    const int pa = wh + 2, pc = wq + 2;
    float* bufA = sm;                              // [(hh+2)][pa]  gs1, later gt1
    float* bufB = bufA + (hh + 2) * pa;            // [(hh+2)][pa]  up(r), later gu1, later down(Z)
    float* bufC = bufB + (hh + 2) * pa;            // [(hq+2)][pc]  gr
    float* bufD = bufC + (hq + 2) * pc;            // [(hq+2)][pc]  down(t1), later gd3
    float* scr = bufD + (hq + 2) * pc;             // H * wh floats: saved r | row-contracted planes
    float* tx1 = scr + (size_t)H * wh;             // [wq][11] x2-adjoint plans along x, [hq][11] along y
    float* ty1 = tx1 + wq * 11;
    float* tx0 = ty1 + hq * 11;                    // [wh][4] x0.5-adjoint plans along x, [hh][4] along y
    float* ty0 = tx0 + wh * 4;
    const int plane = blockIdx.x, c = plane % a.C;
    const float* __restrict__ gs1 = a.gs1 + (size_t)plane * hh * wh;
    const float* __restrict__ rin = a.r + (size_t)plane * hq * wq;
    const float* __restrict__ t1 = a.t1 + (size_t)plane * hh * wh;
    const float* __restrict__ z = a.z + (size_t)plane * H * W;
    for (int i = threadIdx.x; i < 2 * (hh + 2) * pa + 2 * (hq + 2) * pc; i += NT) sm[i] = 0.f;   // zero rings (and interiors)
    build_adj_plans<1>(tx1, wq, wh);
    build_adj_plans<1>(ty1, hq, hh);
    build_adj_plans<0>(tx0, wh, wq);
    build_adj_plans<0>(ty0, hh, hq);
    float* rs = scr;                               // [hq][wq] saved r
    for (int i = threadIdx.x; i < hq * wq; i += NT) rs[i] = rin[i];
    __syncthreads();
    for (int i = threadIdx.x; i < hh * wh; i += NT) { const int y = i / wh, x = i - y * wh; bufA[(y + 1) * pa + x + 1] = gs1[i]; }
    // ---- DT1: its conv input up(r), weight gradients, gu1 = dw_DT1^T gs1
    resample_plane<1>([&](int y, int x) { return rs[y * wq + x]; }, hq, wq, scr + hq * wq, bufB, pa, hh, wh);
    float part[10], w9[9];
    dw_grad_partials(bufA, bufB, pa, hh, wh, part);
    {
        const float r = block_sum_n<10>(part, red);     // (its barriers also order the reads of up(r) before gu1 overwrites it)
        if (threadIdx.x < 10) a.part[((size_t)0 * a.planes + plane) * 10 + threadIdx.x] = r;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) w9[k] = a.wdt1[c * 9 + k];
    for (int i = threadIdx.x; i < hh * wh; i += NT) { const int y = i / wh, x = i - y * wh; bufB[(y + 1) * pa + x + 1] = dw_transpose_at(bufA, pa, y, x, w9); }
    __syncthreads();
    // ---- gr = up^T gu1
    adjoint_plane<1>(bufB, pa, hh, wh, scr, bufC, pc, hq, wq, tx1, ty1);
    // ---- D3: its conv input down(t1), weight gradients, gd3 = dw_D3^T gr
    resample_plane<0>([&](int y, int x) { return t1[(size_t)y * wh + x]; }, hh, wh, scr, bufD, pc, hq, wq);
    dw_grad_partials(bufC, bufD, pc, hq, wq, part);
    {
        const float r = block_sum_n<10>(part, red);
        if (threadIdx.x < 10) a.part[((size_t)1 * a.planes + plane) * 10 + threadIdx.x] = r;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) w9[k] = a.wd3[c * 9 + k];
    for (int i = threadIdx.x; i < hq * wq; i += NT) { const int y = i / wq, x = i - y * wq; bufD[(y + 1) * pc + x + 1] = dw_transpose_at(bufC, pc, y, x, w9); }
    __syncthreads();
    // ---- gt1 = down^T gd3 (over the dead gs1)
    adjoint_plane<0>(bufD, pc, hq, wq, scr, bufA, pa, hh, wh, tx0, ty0);
    // ---- D1: its conv input down(Z), weight gradients, gd1 = dw_D1^T gt1 -> global
    resample_plane<0>([&](int y, int x) { return z[(size_t)y * W + x]; }, H, W, scr, bufB, pa, hh, wh);
    dw_grad_partials(bufA, bufB, pa, hh, wh, part);
    {
        const float r = block_sum_n<10>(part, red);
        if (threadIdx.x < 10) a.part[((size_t)2 * a.planes + plane) * 10 + threadIdx.x] = r;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) w9[k] = a.wd1[c * 9 + k];
    float* __restrict__ out = a.gd1 + (size_t)plane * hh * wh;
    for (int i = threadIdx.x; i < hh * wh; i += NT) { const int y = i / wh, x = i - y * wh; out[i] = dw_transpose_at(bufA, pa, y, x, w9); }
}

// depthwise 3x3 (zero padding) + bias at (y, x): u has a zero ring; same summation order as k_resample_dw
__device__ __forceinline__ float dw_at(const float* u, int p, int y, int x, const float (&w)[9], float bias) {
    float v = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) v += w[dy * 3 + dx] * u[(y + dy) * p + x + dx];
    return v + bias;
}

// Forward counterpart: t1 = dw_D1(down(Z)), r = dw_D3(down(t1)) - ms, s1 = dw_DT1(up(r))  (unlg_former.py:29-33,58) -- three launches of
// 5 - 8 us on 0.5 - 2 MB tensors -- in one, one workgroup per plane, every intermediate in LDS; t1 / r / s1 still go to HBM (the backward
// and the full-resolution DT.3 stage read them).
__global__ __launch_bounds__(NT) void k_dstep_low_fwd(DstepLowFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = a.H, W = a.W, hh = H / 2, wh = W / 2, hq = H / 4, wq = W / 4;
    const int pa = wh + 2, pc = wq + 2;
    float* bufA = sm;                              // [(hh+2)][pa]  down(Z), later up(r)
    float* bufC = bufA + (hh + 2) * pa;            // [(hq+2)][pc]  down(t1)
    float* t1s = bufC + (hq + 2) * pc;             // [hh][wh]      t1
    float* rs = t1s + hh * wh;                     // [hq][wq]      r
    float* scr = rs + hq * wq;                     // H * wh floats: row-contracted planes
    const int plane = blockIdx.x, c = plane % a.C;
    const float* __restrict__ z = a.z + (size_t)plane * H * W;
    const float* __restrict__ ms = a.ms + (size_t)plane * hq * wq;
    for (int i = threadIdx.x; i < (hh + 2) * pa + (hq + 2) * pc; i += NT) sm[i] = 0.f;   // zero rings
    __syncthreads();
    float w9[9];
    // ---- t1 = dw_D1(down(Z))
    resample_plane<0>([&](int y, int x) { return z[(size_t)y * W + x]; }, H, W, scr, bufA, pa, hh, wh);
#pragma unroll
    for (int k = 0; k < 9; ++k) w9[k] = a.wd1[c * 9 + k];
    {
        const float bias = a.bd1[c];
        float* __restrict__ o = a.t1 + (size_t)plane * hh * wh;
        for (int i = threadIdx.x; i < hh * wh; i += NT) { const int y = i / wh, x = i - y * wh; const float v = dw_at(bufA, pa, y, x, w9, bias); t1s[i] = v; o[i] = v; }
    }
    __syncthreads();
    // ---- r = dw_D3(down(t1)) - ms
    resample_plane<0>([&](int y, int x) { return t1s[y * wh + x]; }, hh, wh, scr, bufC, pc, hq, wq);
#pragma unroll
    for (int k = 0; k < 9; ++k) w9[k] = a.wd3[c * 9 + k];
    {
        const float bias = a.bd3[c];
        float* __restrict__ o = a.r + (size_t)plane * hq * wq;
        for (int i = threadIdx.x; i < hq * wq; i += NT) { const int y = i / wq, x = i - y * wq; const float v = dw_at(bufC, pc, y, x, w9, bias) - ms[i]; rs[i] = v; o[i] = v; }
    }
    __syncthreads();
    // ---- s1 = dw_DT1(up(r))   (up(r) over the dead down(Z); its ring is still zero)
    resample_plane<1>([&](int y, int x) { return rs[y * wq + x]; }, hq, wq, scr, bufA, pa, hh, wh);
#pragma unroll
    for (int k = 0; k < 9; ++k) w9[k] = a.wdt1[c * 9 + k];
    {
        const float bias = a.bdt1[c];
        float* __restrict__ o = a.s1 + (size_t)plane * hh * wh;
        for (int i = threadIdx.x; i < hh * wh; i += NT) { const int y = i / wh, x = i - y * wh; o[i] = dw_at(bufA, pa, y, x, w9, bias); }
    }
}

}   // namespace

size_t dstep_low_fwd_lds_bytes(int H, int W) {
    const int hh = H / 2, wh = W / 2, hq = H / 4, wq = W / 4;
    return ((size_t)(hh + 2) * (wh + 2) + (size_t)(hq + 2) * (wq + 2) + (size_t)hh * wh + (size_t)hq * wq + (size_t)H * wh) * sizeof(float);
}
bool dstep_low_fwd_fits(int H, int W, int planes) { return (H % 4) == 0 && (W % 4) == 0 && planes <= 65535 && dstep_low_fwd_lds_bytes(H, W) <= 150 * 1024; }
int launch_dstep_low_fwd(const DstepLowFwdArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_DATASTEP, s);
    if (!dstep_low_fwd_fits(a.H, a.W, a.planes)) { lg_set_error("dstep_low_fwd: %dx%d planes do not fit LDS", a.H, a.W); return -2; }
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_dstep_low_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) { lg_set_error("dstep_low_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    k_dstep_low_fwd<<<a.planes, NT, dstep_low_fwd_lds_bytes(a.H, a.W), s>>>(a);
    LG_CHECK_LAUNCH();
    return 0;
}

size_t dstep_low_bwd_lds_bytes(int H, int W) {
    const int hh = H / 2, wh = W / 2, hq = H / 4, wq = W / 4;
    return ((size_t)2 * (hh + 2) * (wh + 2) + (size_t)2 * (hq + 2) * (wq + 2) + (size_t)H * wh + (size_t)(wq + hq) * 11 + (size_t)(wh + hh) * 4) * sizeof(float);
}
bool dstep_low_bwd_fits(int H, int W, int planes) { return (H % 4) == 0 && (W % 4) == 0 && planes <= 65535 && dstep_low_bwd_lds_bytes(H, W) <= 150 * 1024; }

int launch_dstep_low_bwd(const DstepLowBwdArgs& a, hipStream_t s) {
    if (!dstep_low_bwd_fits(a.H, a.W, a.planes)) { lg_set_error("dstep_low_bwd: %dx%d planes do not fit LDS", a.H, a.W); return -2; }
    if (!a.part) { lg_set_error("dstep_low_bwd: partial-sum scratch missing"); return -2; }
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_dstep_low_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) { lg_set_error("dstep_low_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    k_dstep_low_bwd<<<a.planes, NT, dstep_low_bwd_lds_bytes(a.H, a.W), s>>>(a);
    LG_CHECK_LAUNCH();
    return 0;
}
