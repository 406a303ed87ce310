// Backward of the pixelwise / resampling kernels (k_pixel.hip) for gfx950: data module (D, DT, R, RT, eta),
// LGT tail / patch_embed / down / up+fusion.  Autograd of reference models/unlg_former.py:29-37,58-61 and
// models/common/LGT.py:64-88,280-281,294-295,302-303.  Small parameter gradients (a few floats per tensor) leave every workgroup as a
// partial row and are summed in a fixed order by the deferred reduce launch (bwd_kernels.h: no float atomics, bitwise reproducible);
// the tail's and patch_embed's own 1x1-conv weight gradients are accumulated in their backward kernels, the other 1x1 convs' go
// through k_wgrad.hip.  Kernels whose traffic is NHWC rows use lane = (pixel, channel quad); planar ones one lane per pixel.
#include <string.h>

#include "kernels.h"
#include "bwd_kernels.h"
#include "resample_tile.h"

// sum v[0..N) over the 256 threads of the block; result valid in threads 0..N-1 (returned value for index tid)
template <int N>
__device__ __forceinline__ float block_sum(float (&v)[N], float* sm /* [4*N] */) {
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
    if (threadIdx.x < N) r = sm[threadIdx.x] + sm[N + threadIdx.x] + sm[2 * N + threadIdx.x] + sm[3 * N + threadIdx.x];
    __syncthreads();
    return r;
}

// Adjoint of the separable clamped polyphase resampler, itself evaluated separably through LDS.  A workgroup owns RT input
// rows of one plane: the gout rows they touch are staged once (G), contracted along x into T[gout row][ix] with the per-column
// plans, then along y with the per-row plans -- 2 NC LDS reads per element instead of the NC x NC global gathers (100 for the
// x2 adjoint) of the direct form, and the plans (a few hundred ALU ops each) are built once per workgroup, not per element.
template <int MODE>
__global__ __launch_bounds__(256) void k_resample_adj(const float* __restrict__ gout, float* __restrict__ gin, int hi, int wi, int ho, int wo,
                                                      int accumulate, int rt, int grows) {
    constexpr int NC = AdjPlan<MODE>::NC, PL = NC + 1;
    extern __shared__ float sm[];
    float* G = sm;                                  // [grows][wo]
    float* T = G + (size_t)grows * wo;              // [grows][wi]
    float* spx = T + (size_t)grows * wi;            // [wi][PL]: base (int bits), NC coefficients
    float* spy = spx + (size_t)wi * PL;             // [rt][PL]
    const int ry0 = blockIdx.x * rt;
    const int nrow = min(rt, hi - ry0);
    const float* g = gout + (size_t)blockIdx.y * ho * wo;
    for (int i = threadIdx.x; i < wi + nrow; i += 256) {
        AdjPlan<MODE> pl;
        float* dst;
        if (i < wi) { pl.make(i, wi, wo); dst = spx + i * PL; }
        else { pl.make(ry0 + i - wi, hi, ho); dst = spy + (i - wi) * PL; }
        dst[0] = __int_as_float(pl.base);
#pragma unroll
        for (int a = 0; a < NC; ++a) dst[1 + a] = pl.coef[a];
    }
    // gout rows [gy0, gy0 + grows): the windows of rows ry0 .. ry0 + nrow - 1 (rows outside the image are never weighted)
    const int gy0 = (MODE == 0) ? (ry0 / 2 - 1) : (2 * ry0 - 4);
    for (int i = threadIdx.x; i < grows * wo; i += 256) {
        const int ly = (int)((unsigned)i / (unsigned)wo), x = i - ly * wo;
        const int gy = gy0 + ly;
        G[i] = (gy >= 0 && gy < ho) ? g[(size_t)gy * wo + x] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < grows * wi; i += 256) {
        const int ly = (int)((unsigned)i / (unsigned)wi), ix = i - ly * wi;
        const float* pp = spx + ix * PL;
        const int bx = __float_as_int(pp[0]);
        const float* grow = G + ly * wo;
        float acc = 0.f;
#pragma unroll
        for (int b = 0; b < NC; ++b) acc += pp[1 + b] * grow[clampi(bx + b, 0, wo - 1)];   // clamped taps carry coefficient 0
        T[i] = acc;
    }
    __syncthreads();
    float* out = gin + (size_t)blockIdx.y * hi * wi + (size_t)ry0 * wi;
    for (int i = threadIdx.x; i < nrow * wi; i += 256) {
        const int ly = (int)((unsigned)i / (unsigned)wi), ix = i - ly * wi;
        const float* pp = spy + ly * PL;
        const int by = __float_as_int(pp[0]) - gy0;     // first T row of this input row's window
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < NC; ++a) acc += pp[1 + a] * T[clampi(by + a, 0, grows - 1) * wi + ix];
        if (accumulate) out[i] += acc; else out[i] = acc;
    }
}

int launch_resample_adj(int mode, const float* gout, float* gin, int planes, int hi, int wi, int accumulate, hipStream_t s) {
    const int ho = mode == 0 ? hi / 2 : hi * 2, wo = mode == 0 ? wi / 2 : wi * 2;
    if (planes > 65535) { lg_set_error("resample_adj: %d planes exceed the grid limit", planes); return -2; }
    const int NC = mode == 0 ? 3 : 10;
    // rows per workgroup: as many as keep G + T under ~48 KB (three workgroups per CU), at least one
    int rt = 16;
    auto grows_of = [&](int r) { return mode == 0 ? ((r - 1) / 2 + 3) : (2 * r + 8); };
    while (rt > 1 && (size_t)grows_of(rt) * (wo + wi) * sizeof(float) > 48 * 1024) rt >>= 1;
    const int grows = grows_of(rt);
    const size_t lds = ((size_t)grows * (wo + wi) + (size_t)(wi + rt) * (NC + 1)) * sizeof(float);
    if (lds > 150 * 1024) { lg_set_error("resample_adj: plane width %d too large", wi); return -2; }
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_resample_adj<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_resample_adj<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) { lg_set_error("resample_adj: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    dim3 grid((hi + rt - 1) / rt, planes);
    if (mode == 0) k_resample_adj<0><<<grid, 256, lds, s>>>(gout, gin, hi, wi, ho, wo, accumulate, rt, grows);
    else k_resample_adj<1><<<grid, 256, lds, s>>>(gout, gin, hi, wi, ho, wo, accumulate, rt, grows);
    LG_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// backward of "resample -> depthwise 3x3": gin = dw^T gout ; dW, dbias
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_dw_bwd(DwBwdArgs a) {
    __shared__ float Gt[34][35];
    __shared__ float Ut[34][35];
    __shared__ float red[4 * 10];
    __shared__ float rs_scratch[MODE == 1 ? RsTile<1>::FLOATS : 1];   // x2 only (k_resample_dw)
    const int plane = blockIdx.z;
    const int c = plane % a.C;
    const int ty0 = blockIdx.y * 32, tx0 = blockIdx.x * 32;
    const float* __restrict__ in = a.in + (size_t)plane * a.hi * a.wi;
    const float* __restrict__ go = a.gout + (size_t)plane * a.n_h * a.n_w;
    for (int i = threadIdx.x; i < 34 * 34; i += 256) {
        int uy = i / 34, ux = i - uy * 34;
        int oy = ty0 + uy - 1, ox = tx0 + ux - 1;
        float u = 0.f, g = 0.f;
        if (oy >= 0 && oy < a.n_h && ox >= 0 && ox < a.n_w) {
            if (MODE == 0) u = resample_at<MODE>(in, a.hi, a.wi, oy, ox);
            g = go[oy * a.n_w + ox];
        }
        if (MODE == 0) Ut[uy][ux] = u;
        Gt[uy][ux] = g;
    }
    if constexpr (MODE == 1) resample_tile34<1>(in, a.hi, a.wi, a.n_h, a.n_w, ty0, tx0, Ut, rs_scratch);
    __syncthreads();
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = a.w9[c * 9 + k];
    float part[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) part[k] = 0.f;
    for (int i = threadIdx.x; i < 1024; i += 256) {
        int ly = i >> 5, lx = i & 31;
        int oy = ty0 + ly, ox = tx0 + lx;
        if (oy < a.n_h && ox < a.n_w) {
            float gi = 0.f;
            const float gc = Gt[ly + 1][lx + 1];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    gi += w[dy * 3 + dx] * Gt[ly + 2 - dy][lx + 2 - dx];
                    part[dy * 3 + dx] += gc * Ut[ly + dy][lx + dx];
                }
            part[9] += gc;
            a.gin[((size_t)plane * a.n_h + oy) * a.n_w + ox] = gi;
        }
    }
    float r = block_sum<10>(part, red);
    // partial row of this workgroup: slice = (sample, tile), then channel
    const int tiles = gridDim.x * gridDim.y, tile = blockIdx.y * gridDim.x + blockIdx.x;
    const size_t slice = (size_t)(plane / a.C) * tiles + tile;
    if (threadIdx.x < 10) a.part[(slice * a.C + c) * 10 + threadIdx.x] = r;
}

// one wave per (channel, k): fixed summation order
__global__ __launch_bounds__(64) void k_reduce_chan(const float* __restrict__ part, ChanReduce m) {
    const int c = blockIdx.x / m.NK, k = blockIdx.x - c * m.NK, lane = threadIdx.x;
    const bool allc = (m.allc_mask >> k) & 1u;
    if (allc && c != 0) return;
    const int c_end = allc ? m.C : c + 1;
    float sv[4] = {0.f, 0.f, 0.f, 0.f};
    for (int cc = c; cc < c_end; ++cc) {
        int i = lane;
        for (; i + 192 < m.nslices; i += 256) {
#pragma unroll
            for (int u = 0; u < 4; ++u) sv[u] += part[((size_t)(i + 64 * u) * m.C + cc) * m.NK + k];
        }
        for (; i < m.nslices; i += 64) sv[0] += part[((size_t)i * m.C + cc) * m.NK + k];
    }
    float t = (sv[0] + sv[1]) + (sv[2] + sv[3]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
    if (lane == 0) {
        const int o = allc ? 0 : c * m.stride[k];
        m.dst[k][o] += t;
        if (m.dst2[k]) m.dst2[k][o] += t;
    }
}
int launch_reduce_chan(const float* part, const ChanReduce& m, hipStream_t s) {
    if (m.NK < 1 || m.NK > 14 || m.C < 1 || m.nslices < 1) { lg_set_error("reduce_chan: bad shape"); return -2; }
    int qrc = 0;
    if (reduce_chan_enqueue(part, m, &qrc)) return qrc;
    k_reduce_chan<<<m.C * m.NK, 64, 0, s>>>(part, m);
    LG_CHECK_LAUNCH();
    return 0;
}
size_t chan_partial_floats(int C, int B, int H, int W) { return (size_t)((H + 31) / 32) * ((W + 31) / 32) * B * C * 14; }

int launch_dw_bwd(int mode, const DwBwdArgs& a, hipStream_t s) {
    if (!a.part) { lg_set_error("dw_bwd: partial-sum scratch missing"); return -2; }
    dim3 grid((a.n_w + 31) / 32, (a.n_h + 31) / 32, a.planes);
    if (mode == 0) k_dw_bwd<0><<<grid, 256, 0, s>>>(a);
    else k_dw_bwd<1><<<grid, 256, 0, s>>>(a);
    LG_CHECK_LAUNCH();
    ChanReduce m;
    memset(&m, 0, sizeof(m));
    for (int k = 0; k < 9; ++k) { m.dst[k] = a.dw9 + k; m.stride[k] = 9; }
    m.dst[9] = a.dbias; m.stride[9] = 1;
    m.NK = 10; m.C = a.C; m.nslices = (int)(grid.x * grid.y) * (a.planes / a.C); m.allc_mask = 0;
    return launch_reduce_chan(a.part, m, s);
}

// ------------------------------------------------------------------------------------------------
// top of the data-step backward: through Z' = Z - eta*(dw(up(s1)) + RT(R(Z)-pan))   (unlg_former.py:59-61)
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void k_dstep_top_bwd(DstepTopArgs a) {
    __shared__ float Gm[34][35];
    __shared__ float Ut[34][35];
    __shared__ float red[4 * 14];
    __shared__ float rs_scratch[RsTile<1>::FLOATS];
    const int plane = blockIdx.z;
    const int c = plane % C;
    const int b = plane / C;
    const int ty0 = blockIdx.y * 32, tx0 = blockIdx.x * 32;
    const size_t hw = (size_t)a.H * a.W;
    const float* __restrict__ s1 = a.s1 + (size_t)plane * (a.H / 2) * (a.W / 2);
    const float* __restrict__ g = a.g + (size_t)plane * hw;
    const float* __restrict__ gb = a.g + (size_t)b * C * hw;
    const float* __restrict__ zb = a.z + (size_t)b * C * hw;
    const float* __restrict__ panb = a.pan + (size_t)b * hw;
    const float eta = a.eta[0];
    // the pixelwise operands of the thread's four pixels (all C planes of Z and of the incoming gradient, PAN) are requested FIRST: their
    // round trip overlaps the LDS phases below instead of sitting, one dependent load after the other, inside the per-pixel loop
    float rz[4], dpr[4], zc[4], gv[4], pn[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int oy = ty0 + (i >> 5), ox = tx0 + (i & 31);
        ok[k] = oy < a.H && ox < a.W;
        const size_t pix = ok[k] ? (size_t)oy * a.W + ox : 0;
        float r = a.rb[0], d = 0.f;
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            const float zv = zb[cc * hw + pix], gg = gb[cc * hw + pix];
            r += a.rw[cc] * zv;
            d += a.rtw[cc] * gg;
            if (cc == c) { zc[k] = zv; gv[k] = gg; }
        }
        rz[k] = r; dpr[k] = -eta * d; pn[k] = panb[pix];
    }
    for (int i = threadIdx.x; i < 34 * 34; i += 256) {
        int uy = i / 34, ux = i - uy * 34;
        int oy = ty0 + uy - 1, ox = tx0 + ux - 1;
        float gm = 0.f;
        if (oy >= 0 && oy < a.H && ox >= 0 && ox < a.W) gm = -eta * g[(size_t)oy * a.W + ox];
        Gm[uy][ux] = gm;
    }
    resample_tile34<1>(s1, a.H / 2, a.W / 2, a.H, a.W, ty0, tx0, Ut, rs_scratch);
    __syncthreads();
    float w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = a.w9[c * 9 + k];
    const float bias3 = a.b9[c], rtw_c = a.rtw[c], rtb_c = a.rtb[c], rw_c = a.rw[c];
    float part[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) part[k] = 0.f;
    float* __restrict__ gu_o = a.gu + (size_t)plane * hw;
    float* __restrict__ dz_o = a.dz + (size_t)plane * hw;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int i = threadIdx.x + 256 * kk;
        const int ly = i >> 5, lx = i & 31;
        if (ok[kk]) {
            const size_t pix = (size_t)(ty0 + ly) * a.W + tx0 + lx;
            const float gm = Gm[ly + 1][lx + 1];
            float mt = bias3, gu = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float u = Ut[ly + dy][lx + dx];
                    mt += w[dy * 3 + dx] * u;
                    gu += w[dy * 3 + dx] * Gm[ly + 2 - dy][lx + 2 - dx];
                    part[dy * 3 + dx] += gm * u;
                }
            const float pr = rz[kk] - pn[kk];
            const float pt = rtw_c * pr + rtb_c;
            part[9] += gm;                                   // d bias(DT.3) and d RT.bias
            part[10] += -gv[kk] * (mt + pt);                 // d eta
            part[11] += gm * pr;                             // d RT.weight[c]
            part[12] += dpr[kk] * zc[kk];                    // d R.weight[c]
            part[13] += (c == 0) ? dpr[kk] : 0.f;            // d R.bias
            gu_o[pix] = gu;
            dz_o[pix] = gv[kk] + rw_c * dpr[kk];
        }
    }
    float r = block_sum<14>(part, red);
    const int tiles = gridDim.x * gridDim.y, tile = blockIdx.y * gridDim.x + blockIdx.x;
    const size_t slice = (size_t)b * tiles + tile;
    if (threadIdx.x < 14) a.part[(slice * C + c) * 14 + threadIdx.x] = r;
}

int launch_dstep_top_bwd(const DstepTopArgs& a, hipStream_t s) {
    if (!a.part) { lg_set_error("dstep_top_bwd: partial-sum scratch missing"); return -2; }
    dim3 grid((a.W + 31) / 32, (a.H + 31) / 32, a.B * a.C);
    if (a.C == 4) k_dstep_top_bwd<4><<<grid, 256, 0, s>>>(a);
    else if (a.C == 8) k_dstep_top_bwd<8><<<grid, 256, 0, s>>>(a);
    else { lg_set_error("dstep_top_bwd: C=%d unsupported", a.C); return -1; }
    LG_CHECK_LAUNCH();
    ChanReduce m;
    memset(&m, 0, sizeof(m));
    for (int k = 0; k < 9; ++k) { m.dst[k] = a.dw9 + k; m.stride[k] = 9; }
    m.dst[9] = a.dbias; m.dst2[9] = a.drtb; m.stride[9] = 1;   // d bias(DT.3) and d RT.bias
    m.dst[10] = a.deta; m.stride[10] = 0;                        // summed over channels
    m.dst[11] = a.drtw; m.stride[11] = 1;
    m.dst[12] = a.drw; m.stride[12] = 1;
    m.dst[13] = a.drb; m.stride[13] = 0;                         // summed over channels
    m.NK = 14; m.C = a.C; m.nslices = (int)(grid.x * grid.y) * a.B; m.allc_mask = (1u << 10) | (1u << 13);
    return launch_reduce_chan(a.part, m, s);
}

// ------------------------------------------------------------------------------------------------
// tail backward: out = Wt x + bt + z
// ------------------------------------------------------------------------------------------------
// Same lane = (pixel, channel quad) layout as k_embed_bwd below: dx leaves as one contiguous 1 KB row per wave-store, lane q copies plane
// q of the residual path, and the conv's own weight gradient dWt[c][k] = sum_p dout[p][c] x[p][k] (C x E values) and bias gradient are
// accumulated here from the block output x the kernel reads anyway -- no padded pixel-major copy of dout, no separate launch.
#define TAIL_BWD_WGS 1024
template <int C, int E>
__global__ __launch_bounds__(256) void k_tail_bwd(TailBwdArgs a) {
    static_assert(E == 4 * C, "one lane per channel quad");
    constexpr int PPW = 256 / C, NACC = 4 * C + 1;   // dWt[c][4q + u] | db[q]
    __shared__ float red[4 * C * NACC];
    const int q = threadIdx.x % C, slot = threadIdx.x / C, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float wq[C][4];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int u = 0; u < 4; ++u) wq[c][u] = a.w[c * E + 4 * q + u];
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    const float* __restrict__ dout = a.dout;
    const float* __restrict__ xp = a.x;
    for (long p0 = (long)blockIdx.x * PPW; p0 < a.total; p0 += (long)gridDim.x * PPW) {
        const long p = p0 + slot;
        const bool pv = p < a.total;
        const long pc_ = pv ? p : 0;
        const long b = pc_ / a.HW, s = pc_ - b * a.HW;
        const float m_ = pv ? 1.f : 0.f;
        float d[C];
#pragma unroll
        for (int c = 0; c < C; ++c) d[c] = dout[(b * C + c) * a.HW + s] * m_;
        const float4 xv = *reinterpret_cast<const float4*>(xp + pc_ * E + 4 * q);
        const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
        float o[4] = {0.f, 0.f, 0.f, 0.f}, dq = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                o[u] += wq[c][u] * d[c];
                acc[c * 4 + u] += d[c] * xx[u];
            }
            if (q == c) dq = d[c];
        }
        acc[4 * C] += dq;
        if (pv) {
            *reinterpret_cast<float4*>(a.dx + p * E + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
            a.dz[(b * C + q) * a.HW + s] = dq;
        }
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = C; off < 64; off <<= 1) v += __shfl_xor(v, off);
        if (lane < C) red[(wave * C + q) * NACC + i] = v;
    }
    __syncthreads();
    // partial rows of this workgroup: [grid][C*E] dWt | [grid][C] db
    const size_t nwg = gridDim.x, wg = blockIdx.x;
    for (int i = threadIdx.x; i < C * NACC; i += 256) {
        const float v = (red[i] + red[C * NACC + i]) + (red[2 * C * NACC + i] + red[3 * C * NACC + i]);
        const int qq = i / NACC, k = i - qq * NACC;
        if (k < 4 * C) a.part[wg * (C * E) + (k / 4) * E + 4 * qq + (k % 4)] = v;
        else a.part[nwg * (C * E) + wg * C + qq] = v;
    }
}

size_t tail_bwd_part_floats(int C) { return (size_t)TAIL_BWD_WGS * ((size_t)4 * C * C + C); }
int launch_tail_bwd(int C, const TailBwdArgs& a, hipStream_t s) {
    if (!a.part || !a.x || !a.d_w || !a.d_b) { lg_set_error("tail_bwd: block output / partial-sum scratch / destinations missing"); return -2; }
    const int E = 4 * C, ppw = 256 / C;
    long nb = (a.total + ppw - 1) / ppw;
    const int grid = (int)(nb < TAIL_BWD_WGS ? nb : TAIL_BWD_WGS);
    if (C == 4) k_tail_bwd<4, 16><<<grid, 256, 0, s>>>(a);
    else if (C == 8) k_tail_bwd<8, 32><<<grid, 256, 0, s>>>(a);
    else { lg_set_error("tail_bwd: C=%d unsupported", C); return -1; }
    LG_CHECK_LAUNCH();
    return launch_reduce_slab_wb(a.part, a.part + (size_t)grid * C * E, grid, C, E, a.d_w, E, a.d_b, s);
}

// ------------------------------------------------------------------------------------------------
// patch_embed backward: x = LN(W (z*dww + dwb) + b)
// ------------------------------------------------------------------------------------------------
// Lane = (pixel, channel quad): the C = E/4 lanes of a pixel each own four of its E channels, so the [P, E] gradient is read as ONE
// contiguous 1 KB row per wave-load (a lane per pixel fetched 64 scattered 16-byte pieces per instruction: 1.4 TB/s), LayerNorm sums
// run across the C lanes with DPP / shuffles, and lane q finishes channel q of the C-channel side (dz, depthwise-1x1 gradients).
// The 1x1 conv's weight gradient dW[n][c] = sum_p de[p][n] t[p][c] (E x C values) and its bias gradient are accumulated right here
// in registers: de and the conv input t never reach HBM and the separate weight-gradient launch is gone.
#define EMBED_BWD_WGS 1024
template <int C, int E>
__global__ __launch_bounds__(256) void k_embed_bwd(EmbedBwdArgs a) {
    static_assert(E == 4 * C, "one lane per channel quad");
    constexpr int PPW = 256 / C;               // pixels per workgroup pass
    constexpr int NACC = 4 * C + 4 + 8 + 2;    // dW rows of the lane's quad | db | d gamma | d beta | d dww, d dwb of channel q
    __shared__ float red[4 * C * NACC];
    const int q = threadIdx.x % C, slot = threadIdx.x / C, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float wq[4][C], bq[4], gq[4], dwwv[C], dwbv[C];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int c = 0; c < C; ++c) wq[u][c] = a.w[(4 * q + u) * C + c];
        bq[u] = a.b[4 * q + u];
        gq[u] = a.lng[4 * q + u];
    }
#pragma unroll
    for (int c = 0; c < C; ++c) { dwwv[c] = a.dww[c]; dwbv[c] = a.dwb[c]; }
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    const float* __restrict__ zp = a.z;
    const float* __restrict__ dxp = a.dx;
    for (long p0 = (long)blockIdx.x * PPW; p0 < a.total; p0 += (long)gridDim.x * PPW) {
        const long p = p0 + slot;
        const bool pv = p < a.total;
        const long pc_ = pv ? p : 0;
        const long b = pc_ / a.HW, s = pc_ - b * a.HW;
        float zc[C], t[C];
#pragma unroll
        for (int c = 0; c < C; ++c) zc[c] = zp[(b * C + c) * a.HW + s];
        const float4 dv = *reinterpret_cast<const float4*>(dxp + pc_ * E + 4 * q);
        const float m_ = pv ? 1.f : 0.f;
        float dxh[4] = {dv.x * m_, dv.y * m_, dv.z * m_, dv.w * m_};
#pragma unroll
        for (int c = 0; c < C; ++c) t[c] = zc[c] * dwwv[c] + dwbv[c];
        float e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) v += wq[u][c] * t[c];
            e[u] = v + bq[u];
        }
        const float mu = lane_group_sum<C>((e[0] + e[1]) + (e[2] + e[3])) * (1.0f / E);
        float vs = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) { e[u] -= mu; vs += e[u] * e[u]; }
        const float rstd = __builtin_amdgcn_rsqf(lane_group_sum<C>(vs) * (1.0f / E) + LG_EPS);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            e[u] *= rstd;                              // x hat
            acc[4 * C + 4 + u] += dxh[u] * e[u];       // d gamma
            acc[4 * C + 8 + u] += dxh[u];              // d beta
            dxh[u] *= gq[u];
            m1 += dxh[u];
            m2 += dxh[u] * e[u];
        }
        m1 = lane_group_sum<C>(m1) * (1.0f / E);
        m2 = lane_group_sum<C>(m2) * (1.0f / E);
        float dt[C];
#pragma unroll
        for (int c = 0; c < C; ++c) dt[c] = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float de = rstd * (dxh[u] - m1 - e[u] * m2);
            acc[4 * C + u] += de;                      // db
#pragma unroll
            for (int c = 0; c < C; ++c) {
                acc[u * C + c] += de * t[c];           // dW[4q + u][c]
                dt[c] += wq[u][c] * de;
            }
        }
        float dtq = 0.f, zq = 0.f, dwq = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float full = lane_group_sum<C>(dt[c]);
            if (q == c) { dtq = full; zq = zc[c]; dwq = dwwv[c]; }
        }
        acc[4 * C + 12] += dtq * zq;                   // d dww[q]
        acc[4 * C + 13] += dtq;                        // d dwb[q]
        if (pv) a.dz[(b * C + q) * a.HW + s] += dtq * dwq;
    }
    // lanes with equal q hold partials of the same outputs: across the wave, then the 4 waves in LDS (fixed order)
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = C; off < 64; off <<= 1) v += __shfl_xor(v, off);
        if (lane < C) red[(wave * C + q) * NACC + i] = v;
    }
    __syncthreads();
    // partial rows of this workgroup: [grid][E*C] dW | [grid][E] db | [grid][E] d gamma | [grid][E] d beta | [grid][C] d dww | [grid][C] d dwb
    const size_t nwg = gridDim.x, wg = blockIdx.x;
    for (int i = threadIdx.x; i < C * NACC; i += 256) {
        const float v = (red[i] + red[C * NACC + i]) + (red[2 * C * NACC + i] + red[3 * C * NACC + i]);
        const int qq = i / NACC, k = i - qq * NACC;
        float* base = a.part;
        if (k < 4 * C) { base[wg * (E * C) + (4 * qq + k / C) * C + (k % C)] = v; continue; }
        base += nwg * (E * C);
        if (k < 4 * C + 4) { base[wg * E + 4 * qq + (k - 4 * C)] = v; continue; }
        base += nwg * E;
        if (k < 4 * C + 8) { base[wg * E + 4 * qq + (k - 4 * C - 4)] = v; continue; }
        base += nwg * E;
        if (k < 4 * C + 12) { base[wg * E + 4 * qq + (k - 4 * C - 8)] = v; continue; }
        base += nwg * E;
        if (k == 4 * C + 12) base[wg * C + qq] = v;
        else base[nwg * C + wg * C + qq] = v;
    }
}

size_t embed_bwd_part_floats(int C) { return (size_t)EMBED_BWD_WGS * ((size_t)4 * C * C + 12 * C + 2 * C); }
int launch_embed_bwd(int C, const EmbedBwdArgs& a, hipStream_t s) {
    if (!a.part || !a.d_w || !a.d_b) { lg_set_error("embed_bwd: partial-sum scratch / weight-gradient destinations missing"); return -2; }
    const int E = 4 * C, ppw = 256 / C;
    long nb = (a.total + ppw - 1) / ppw;
    const int grid = (int)(nb < EMBED_BWD_WGS ? nb : EMBED_BWD_WGS);
    if (C == 4) k_embed_bwd<4, 16><<<grid, 256, 0, s>>>(a);
    else if (C == 8) k_embed_bwd<8, 32><<<grid, 256, 0, s>>>(a);
    else { lg_set_error("embed_bwd: C=%d unsupported", C); return -1; }
    LG_CHECK_LAUNCH();
    const size_t g = (size_t)grid;
    const float* pw = a.part;
    const float* pb = pw + g * E * C;
    const float* pg = pb + g * E;
    const float* pbt = pg + g * E;
    const float* pd = pbt + g * E;
    int rc = launch_reduce_slab_wb(pw, pb, grid, E, C, a.d_w, C, a.d_b, s);
    if (rc) return rc;
    rc = launch_reduce_slab_pair(pg, pbt, grid, E, a.d_lng, a.d_lnb, s);
    if (rc) return rc;
    return launch_reduce_slab_pair(pd, pd + g * C, grid, C, a.d_dww, a.d_dwb, s);
}

// ------------------------------------------------------------------------------------------------
// down backward: y = Wd * down2(x) + bd
// ------------------------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(256) void k_down_bwd_a(DownBwdArgs a) {
    __shared__ float sW[2 * E * E];
    lds_stage(sW, a.w, 2 * E * E);
    __syncthreads();
    long total = (long)a.B * (a.H / 2) * (a.W / 2);
    long p = blockIdx.x * 256L + threadIdx.x;
    if (p >= total) return;
    float dy[2 * E];
    const float4* src = reinterpret_cast<const float4*>(a.dy + p * 2 * E);
#pragma unroll
    for (int k = 0; k < 2 * E / 4; ++k) {
        float4 v = src[k];
        dy[4 * k] = v.x; dy[4 * k + 1] = v.y; dy[4 * k + 2] = v.z; dy[4 * k + 3] = v.w;
    }
    float4* duo = reinterpret_cast<float4*>(a.du + p * E);
#pragma unroll
    for (int k4 = 0; k4 < E / 4; ++k4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float acc = 0.f;
#pragma unroll
            for (int n = 0; n < 2 * E; ++n) acc += sW[n * E + k4 * 4 + u] * dy[n];
            v[u] = acc;
        }
        duo[k4] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// lane = (pixel, channel quad): one float4 of the skip gradient and of each adjoint tap per lane
template <int E>
__global__ __launch_bounds__(256) void k_down_bwd_b(DownBwdArgs a) {
    constexpr int LPP = E / 4, PPW = 256 / LPP;
    const int q = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const long total = (long)a.B * a.H * a.W;
    const long p = (long)blockIdx.x * PPW + slot;
    if (p >= total) return;
    const int ix = (int)(p % a.W);
    const long r = p / a.W;
    const int iy = (int)(r % a.H);
    const long b = r / a.H;
    const int ho = a.H / 2, wo = a.W / 2;
    AdjPlan<0> py, px;
    py.make(iy, a.H, ho);
    px.make(ix, a.W, wo);
    float4 acc = *reinterpret_cast<const float4*>(a.dskip + p * E + 4 * q);
#pragma unroll
    for (int ya = 0; ya < 3; ++ya) {
        const int yy = clampi(py.base + ya, 0, ho - 1);       // taps outside carry coefficient 0
#pragma unroll
        for (int xb = 0; xb < 3; ++xb) {
            const int xx = clampi(px.base + xb, 0, wo - 1);
            const float wgt = py.coef[ya] * px.coef[xb];
            const float4 v = *reinterpret_cast<const float4*>(a.du + ((b * ho + yy) * (long)wo + xx) * E + 4 * q);
            acc.x += wgt * v.x; acc.y += wgt * v.y; acc.z += wgt * v.z; acc.w += wgt * v.w;
        }
    }
    *reinterpret_cast<float4*>(a.dx + p * E + 4 * q) = acc;
}

int launch_down_bwd_a(int E, const DownBwdArgs& a, hipStream_t s) {
    long total = (long)a.B * (a.H / 2) * (a.W / 2);
    int grid = (int)((total + 255) / 256);
    if (E == 16) k_down_bwd_a<16><<<grid, 256, 0, s>>>(a);
    else if (E == 32) k_down_bwd_a<32><<<grid, 256, 0, s>>>(a);
    else { lg_set_error("down_bwd: E=%d unsupported", E); return -1; }
    LG_CHECK_LAUNCH();
    return 0;
}
int launch_down_bwd_b(int E, const DownBwdArgs& a, hipStream_t s) {
    long total = (long)a.B * a.H * a.W;
    const int ppw = 256 / (E / 4);
    int grid = (int)((total + ppw - 1) / ppw);
    if (E == 16) k_down_bwd_b<16><<<grid, 256, 0, s>>>(a);
    else if (E == 32) k_down_bwd_b<32><<<grid, 256, 0, s>>>(a);
    else { lg_set_error("down_bwd: E=%d unsupported", E); return -1; }
    LG_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// up + fusion backward: y = Wf [t ; skip] + bf,  t = Wu up2(xb) + bu  (1x1 conv and resampler commute)
// ------------------------------------------------------------------------------------------------
// lane = (pixel, channel quad): dy is exchanged inside the wave through LDS, lane q forms columns 4q .. 4q+3 of both halves of Wf^T dy
template <int E>
__global__ __launch_bounds__(256) void k_upfuse_bwd_a(UpFuseBwdArgs a) {
    constexpr int LPP = E / 4, PPW = 256 / LPP, LDU = E + 4;
    __shared__ float sFw[E * 2 * E];
    __shared__ __attribute__((aligned(16))) float dx_[PPW * LDU];
    lds_stage(sFw, a.fw, E * 2 * E);
    __syncthreads();
    const int q = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const long total = (long)a.B * a.H * a.W;
    for (long p = (long)blockIdx.x * PPW + slot; p < total; p += (long)gridDim.x * PPW) {   // whole pixels (total is a multiple of PPW)
        *reinterpret_cast<float4*>(dx_ + slot * LDU + 4 * q) = *reinterpret_cast<const float4*>(a.dy + p * E + 4 * q);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();     // the lanes of a pixel sit in one wave
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float dy[E];
#pragma unroll
        for (int k = 0; k < E / 4; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(dx_ + slot * LDU + 4 * k);
            dy[4 * k] = v.x; dy[4 * k + 1] = v.y; dy[4 * k + 2] = v.z; dy[4 * k + 3] = v.w;
        }
        __builtin_amdgcn_wave_barrier();     // dx_ is rewritten by the next pixel group
        float v[4] = {0.f, 0.f, 0.f, 0.f}, w[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < E; ++n) {
            const float4 f1 = *reinterpret_cast<const float4*>(sFw + n * 2 * E + 4 * q);
            const float4 f2 = *reinterpret_cast<const float4*>(sFw + n * 2 * E + E + 4 * q);
            v[0] += f1.x * dy[n]; v[1] += f1.y * dy[n]; v[2] += f1.z * dy[n]; v[3] += f1.w * dy[n];
            w[0] += f2.x * dy[n]; w[1] += f2.y * dy[n]; w[2] += f2.z * dy[n]; w[3] += f2.w * dy[n];
        }
        *reinterpret_cast<float4*>(a.dt + p * E + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(a.dskip + p * E + 4 * q) = make_float4(w[0], w[1], w[2], w[3]);
    }
}

// Second half, SEPARABLY and in the lane = (pixel, channel quad) layout: k_upadj_h contracts the x2 resampler's adjoint along x
// (dt [B,H,W,E] -> tmp [B,H,W/2,E]), k_upfuse_bwd_b along y and applies Wu^T.  10 + 10 coalesced float4 taps per lane instead of the
// 10 x 10 window of 64-byte gathers per level-1 pixel of the direct form (92 us -> two launches of ~12 us).
template <int E>
__global__ __launch_bounds__(256) void k_upadj_h(const float* __restrict__ dt, float* __restrict__ tmp, int rows, int W) {
    constexpr int LPP = E / 4, PPW = 256 / LPP;
    const int q = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const int wi = W / 2;
    const int ix = blockIdx.x * PPW + slot;
    if (ix >= wi) return;
    AdjPlan<1> px;
    px.make(ix, wi, W);
    for (long row = blockIdx.y; row < rows; row += gridDim.y) {   // row = b * H + y
        const float* __restrict__ src = dt + (row * W) * E + 4 * q;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int x = clampi(px.base + k, 0, W - 1);     // clamped taps carry coefficient 0
            const float4 v = *reinterpret_cast<const float4*>(src + (long)x * E);
            const float c = px.coef[k];
            acc.x += c * v.x; acc.y += c * v.y; acc.z += c * v.z; acc.w += c * v.w;
        }
        *reinterpret_cast<float4*>(tmp + (row * wi + ix) * E + 4 * q) = acc;
    }
}

template <int E>
__global__ __launch_bounds__(256) void k_upfuse_bwd_b(UpFuseBwdArgs a) {
    constexpr int LPP = E / 4, PPW = 256 / LPP, NO = 2 * E / LPP, LDU = E + 4;
    static_assert(NO == 8, "eight outputs per lane");
    __shared__ float sUw[E * 2 * E];
    __shared__ __attribute__((aligned(16))) float vx[PPW * LDU];
    lds_stage(sUw, a.upw, E * 2 * E);
    __syncthreads();
    const int q = threadIdx.x % LPP, slot = threadIdx.x / LPP;
    const int hi = a.H / 2, wi = a.W / 2;
    const long total = (long)a.B * hi * wi;
    const long p = (long)blockIdx.x * PPW + slot;
    const bool pv = p < total;
    const long pc_ = pv ? p : 0;
    const int ix = (int)(pc_ % wi);
    const long r = pc_ / wi;
    const int iy = (int)(r % hi);
    const long b = r / hi;
    AdjPlan<1> py;
    py.make(iy, hi, a.H);
    const float* __restrict__ src = a.tmp + ((b * a.H) * (long)wi + ix) * E + 4 * q;
    float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const int y = clampi(py.base + k, 0, a.H - 1);
        const float4 t = *reinterpret_cast<const float4*>(src + (long)y * wi * E);
        const float c = py.coef[k];
        v4.x += c * t.x; v4.y += c * t.y; v4.z += c * t.z; v4.w += c * t.w;
    }
    if (pv) *reinterpret_cast<float4*>(a.v + p * E + 4 * q) = v4;
    *reinterpret_cast<float4*>(vx + slot * LDU + 4 * q) = v4;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();     // the lanes of a pixel sit in one wave
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float v[E];
#pragma unroll
    for (int k = 0; k < E / 4; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(vx + slot * LDU + 4 * k);
        v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
    }
    float o[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int n = 0; n < E; ++n) acc += sUw[n * 2 * E + NO * q + j] * v[n];
        o[j] = acc;
    }
    if (pv) {
        float4* dxo = reinterpret_cast<float4*>(a.dxb + p * 2 * E + NO * q);
        dxo[0] = make_float4(o[0], o[1], o[2], o[3]);
        dxo[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
}

int launch_upfuse_bwd_a(int E, const UpFuseBwdArgs& a, hipStream_t s) {
    if (E != 16 && E != 32) { lg_set_error("upfuse_bwd: E=%d unsupported", E); return -1; }
    const long total = (long)a.B * a.H * a.W;
    const int ppw = 256 / (E / 4);
    if (total % ppw) { lg_set_error("upfuse_bwd: %ld pixels are not a multiple of %d", total, ppw); return -2; }
    const long ng = total / ppw;
    const int grid = (int)(ng < 2048 ? ng : 2048);
    if (E == 16) k_upfuse_bwd_a<16><<<grid, 256, 0, s>>>(a);
    else k_upfuse_bwd_a<32><<<grid, 256, 0, s>>>(a);
    LG_CHECK_LAUNCH();
    return 0;
}
int launch_upfuse_bwd_b(int E, const UpFuseBwdArgs& a, hipStream_t s) {
    if (!a.tmp) { lg_set_error("upfuse_bwd: row-contracted scratch missing"); return -2; }
    if (E != 16 && E != 32) { lg_set_error("upfuse_bwd: E=%d unsupported", E); return -1; }
    const int ppw = 256 / (E / 4), wi = a.W / 2;
    const int rows = a.B * a.H;
    dim3 gh((wi + ppw - 1) / ppw, rows < 1024 ? rows : 1024);
    if (E == 16) k_upadj_h<16><<<gh, 256, 0, s>>>(a.dt, a.tmp, rows, a.W);
    else k_upadj_h<32><<<gh, 256, 0, s>>>(a.dt, a.tmp, rows, a.W);
    LG_CHECK_LAUNCH();
    const long total = (long)a.B * (a.H / 2) * wi;
    const int grid = (int)((total + ppw - 1) / ppw);
    if (E == 16) k_upfuse_bwd_b<16><<<grid, 256, 0, s>>>(a);
    else k_upfuse_bwd_b<32><<<grid, 256, 0, s>>>(a);
    LG_CHECK_LAUNCH();
    return 0;
}
