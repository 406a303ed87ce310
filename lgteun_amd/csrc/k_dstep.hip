// The data module's proximal-gradient step as ONE launch per direction for planes that fit a CU's LDS (gfx950: 160 KB).
// Reference: models/unlg_former.py:29-37 (D, DT, R, RT), 58-61 (the update); autograd of the same for the backward.
//
// Everything in D and DT is per (sample, channel) plane -- bicubic x0.5 / x2 resampling (bmu.sampling_unit_) and depthwise 3x3 convolutions
// (bmu.dep_conv) -- and a 128 x 128 fp32 plane is 64 KB: one 1024-thread workgroup owns a plane and walks the whole chain
//     Z -> x0.5 -> dw3 (t1) -> x0.5 -> dw3 - ms (r) -> x2 -> dw3 (s1) -> x2 -> dw3 -> Z - eta (. + RT(R Z - pan))
// through LDS, where the tile kernels of k_pixel.hip / k_bwd_pixel.hip (the general path: any plane size) take four launches forward and
// nine backward, each a few dependent L2 round trips long on tensors of 0.5 ... 8 MB (round 4 trace: 34 + 101 us per stage at C = 4,
// 8 % of the train step).  Only R couples the channels: its per-pixel sum over the C planes is read straight from global memory (L2) by
// every plane's workgroup.  The arithmetic is that of the tile kernels in the same order (row sums of the four taps first, then their
// combination; the nine conv taps in raster order), so the forward results are bitwise those of the general path (tested).
#include "kernels.h"
#include "bwd_kernels.h"
#include "resample_tile.h"
#include <string.h>
#include <type_traits>

// In-kernel phase stamps (diagnostic build only, -DLG_STAMPS: tools/build_stamps.sh + tools/dstep_stamps.py): every wave of every workgroup
// stores s_memtime at the phase boundaries (branch-free: round 4's lesson on conditional stamp stores)
#ifdef LG_STAMPS
__device__ unsigned long long g_ds_stamps[512 * 16 * 32];   // [workgroup][wave][stamp]
// stamps go to LDS and leave in one burst at the kernel's end: a global store in front of the parameter loads would turn those (uniform,
// otherwise scalar) loads into vector loads with a full wait each -- the probe would measure its own disturbance
#define DSTAMP_DECL __shared__ unsigned long long ds_stamp_lds[16 * 32]; int stamp_i = 0; (void)stamp_i; \
    if (threadIdx.x < 16 * 32) ds_stamp_lds[threadIdx.x] = 0; __syncthreads();
#define DSTAMP(i) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)); \
                       ds_stamp_lds[(threadIdx.x >> 6) * 32 + (i)] = t__; } while (0)
#define DSTAMP_FLUSH() do { __syncthreads(); if (threadIdx.x < 16 * 32) g_ds_stamps[(blockIdx.x & 511) * 16 * 32 + threadIdx.x] = ds_stamp_lds[threadIdx.x]; } while (0)
extern "C" __attribute__((visibility("default"))) int lg_debug_ds_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ds_stamps), sizeof(g_ds_stamps));
}
#else
#define DSTAMP_DECL int stamp_i = 0; (void)stamp_i;
#define DSTAMP(i) do { } while (0)
#define DSTAMP_FLUSH() do { } while (0)
#endif

namespace {
constexpr int DS_NT = 1024;

// ---- LDS images.  A plane of n x n values lives at pitch n + 8 with its data at column offset 4 (16-byte aligned quads).  Every phase works on
// QUADS of four consecutive outputs per thread and iteration: the taps of a quad come in as 16- / 8-byte LDS reads, and the one value a quad
// needs from each NEIGHBOUR quad comes from the neighbour lane's registers (DPP wave shift) -- as 4-byte LDS reads at a 16- or 32-byte lane
// stride those were 4- and 8-way bank conflicts and half of the phases' time (round 4 stamps).  Image borders: x0.5 sources clamp in
// registers, x2 sources carry two REPLICATED columns each side (the clamped taps of F.interpolate become plain reads), conv sources a ZERO
// row above and below (the columns beside a row are zeroed in registers).  Pointers passed around point at the data origin (row 0, column 0).
template <int N>
struct DsL {
    static constexpr int H = N / 2, Q = N / 4;
    static constexpr int P0 = N + 8, PH = H + 8, PQ = Q + 8;     // pitches of the padded images
    static constexpr int R0 = (N + 2) * P0;                      // Z rows (resampler source), later the last x2 image with its zero rows
    static constexpr int HX1 = N * (H + 4), HX4 = H * (N + 4);   // row-contracted images: pitch = columns + 4
    static constexpr int R1 = HX1 > HX4 ? HX1 : HX4;
    static constexpr int R2 = (H + 2) * PH;                      // half-resolution conv sources
    static constexpr int R3 = H * PH;                            // t1 / s1 as resampler sources
    // inside R1 while it is free (between the first and the last resampler): the quarter-resolution images
    static constexpr int HX2 = H * (Q + 4), HX3 = Q * (H + 4);
    static constexpr int O4 = 0, O5 = (HX2 > HX3 ? HX2 : HX3), O6 = O5 + (Q + 2) * PQ;
    static_assert(O6 + Q * PQ <= R1, "quarter-resolution images fit the free row-contracted buffer");
    static constexpr int FLOATS = R0 + R1 + R2 + R3;
};

// lane l receives v of lane l - 1 (shr) / l + 1 (shl); lanes without a source keep `edge`.  All lanes of the wave must be active.
__device__ __forceinline__ float ds_shr1(float edge, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float ds_shl1(float edge, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}

// x0.5 along x: Hx[row][ox] = sum_b w[b] S[row][clamp(2 ox - 1 + b)]  (resample_at's row sums in its order); quad m reads inputs 8m .. 8m+7 and
// takes 8m - 1 / 8m + 8 from its neighbour lanes (the row's first / last quad: its own border value = the clamp)
template <int ROWS, int NIC, int SP, int HP>
__device__ __forceinline__ void ds_xhalf(const float* __restrict__ S, float* __restrict__ Hx) {
    constexpr int NQ = NIC / 8;
    static_assert((ROWS * NQ) % 64 == 0 && 64 % NQ == 0, "whole waves, whole rows per wave");
    float w[4];
    cubic_w(0.5f, w);
    for (int i = threadIdx.x; i < ROWS * NQ; i += DS_NT) {
        const int row = i / NQ, m = i % NQ;
        const float* p = S + row * SP + 8 * m;
        const float4 A = *reinterpret_cast<const float4*>(p), B = *reinterpret_cast<const float4*>(p + 4);
        const float lo = ds_shr1(A.x, B.w), hi = ds_shl1(B.w, A.x);
        const float in[10] = {m == 0 ? A.x : lo, A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w, m == NQ - 1 ? B.w : hi};
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float r = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) r += w[b] * in[2 * u + b];
            o[u] = r;
        }
        *reinterpret_cast<float4*>(Hx + row * HP + 4 * m) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
// x2 along x: outputs 4m .. 4m+3 read inputs 2m-2 .. 2m+3 (S carries two replicated columns each side); resample_plan<1>: even o = 2k:
// i0 = k - 1, t = 0.75; odd o = 2k + 1: i0 = k, t = 0.25
template <int ROWS, int NIC, int SP, int HP>
__device__ __forceinline__ void ds_xdouble(const float* __restrict__ S, float* __restrict__ Hx) {
    constexpr int NQ = NIC / 2;
    float we[4], wo[4];
    cubic_w(0.75f, we);
    cubic_w(0.25f, wo);
    for (int i = threadIdx.x; i < ROWS * NQ; i += DS_NT) {
        const int row = i / NQ, m = i % NQ;
        const float* p = S + row * SP + 2 * m - 2;
        const float2 A = *reinterpret_cast<const float2*>(p), B = *reinterpret_cast<const float2*>(p + 2), Cc = *reinterpret_cast<const float2*>(p + 4);
        const float in[6] = {A.x, A.y, B.x, B.y, Cc.x, Cc.y};
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int off = (u + 1) >> 1;   // 0, 1, 1, 2
            float r = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) r += ((u & 1) ? wo[b] : we[b]) * in[off + b];
            o[u] = r;
        }
        *reinterpret_cast<float4*>(Hx + row * HP + 4 * m) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
// along y: U[oy][ox] = sum_a w[a] Hx[clamp(i0 - 1 + a)][ox]; U = data origin of a padded image (16-byte aligned quads)
template <int MODE, int NIR, int NOC, int HP, int UP>
__device__ __forceinline__ void ds_ycon(const float* __restrict__ Hx, float* __restrict__ U) {
    constexpr int NOR = MODE == 0 ? NIR / 2 : NIR * 2, NQ = NOC / 4;
    for (int i = threadIdx.x; i < NOR * NQ; i += DS_NT) {
        const int oy = i / NQ, m = i % NQ;
        int i0;
        float w[4];
        resample_plan<MODE>(oy, i0, w);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float4 v = *reinterpret_cast<const float4*>(Hx + clampi(i0 - 1 + a, 0, NIR - 1) * HP + 4 * m);
            acc.x += w[a] * v.x; acc.y += w[a] * v.y; acc.z += w[a] * v.z; acc.w += w[a] * v.w;
        }
        *reinterpret_cast<float4*>(U + oy * UP + 4 * m) = acc;
    }
}
// depthwise 3x3 of the quad (oy, 4m .. 4m+3) of a conv source with NQ quads per row: the nine taps in raster order, as k_resample_dw
// (k_pixel.hip).  All lanes of the wave call (neighbour columns through DPP), whole rows per wave.
template <int UP, int NQ>
__device__ __forceinline__ void ds_dwq(const float* __restrict__ U, int oy, int m, const float (&w)[9], float (&o)[4]) {
    static_assert(64 % NQ == 0, "whole rows per wave");
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const float4 c = *reinterpret_cast<const float4*>(U + (oy + dy - 1) * UP + 4 * m);
        const float lo = ds_shr1(0.f, c.w), hi = ds_shl1(0.f, c.x);
        const float in[6] = {m == 0 ? 0.f : lo, c.x, c.y, c.z, c.w, m == NQ - 1 ? 0.f : hi};
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] += w[dy * 3 + dx] * in[u + dx];
    }
}
// store a quad of a resampler source (PAD replicated columns each side)
template <int NCOL, int PAD>
__device__ __forceinline__ void ds_put_src(float* __restrict__ row, int m, const float (&o)[4]) {
    *reinterpret_cast<float4*>(row + 4 * m) = make_float4(o[0], o[1], o[2], o[3]);
    if (PAD > 0 && m == 0) {
#pragma unroll
        for (int k = 1; k <= PAD; ++k) row[-k] = o[0];
    }
    if (PAD > 0 && m == NCOL / 4 - 1) {
#pragma unroll
        for (int k = 0; k < PAD; ++k) row[NCOL + k] = o[3];
    }
}
// zero rows -1 and n of an n x n conv source (data origin U, pitch UP)
template <int NN, int UP>
__device__ __forceinline__ void ds_zero_rows(float* __restrict__ U) {
    for (int i = threadIdx.x; i < 2 * NN; i += DS_NT) U[(i < NN ? -UP : NN * UP - NN) + i] = 0.f;
}

template <int N, int C>
__global__ __launch_bounds__(DS_NT) void k_dstep_fwd(DstepFwdArgs a) {
    using L = DsL<N>;
    constexpr int H = L::H, Q = L::Q, P0 = L::P0, PH = L::PH, PQ = L::PQ;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Zs = sm + 4;                               // resampler source, data origin
    float* U4 = sm + P0 + 4;                          // the last conv source (same buffer, one zero row above)
    float* R1 = sm + L::R0;
    float* U1 = R1 + L::R1 + PH + 4;                  // half-resolution conv sources (U1, then U3)
    float* V1 = R1 + L::R1 + L::R2 + 4;               // t1, then s1, as resampler sources
    float* Hq = R1 + L::O4;                           // row-contracted quarter-resolution images
    float* U2 = R1 + L::O5 + PQ + 4;
    float* Rq = R1 + L::O6 + 4;                       // r as resampler source
    DSTAMP_DECL
    DSTAMP(stamp_i++);
    const int plane = blockIdx.x, c = plane % C, b = plane / C;
    const int tid = threadIdx.x;
    // the thread's pixels of the full-resolution plane: NV quads
    constexpr int NV = (N * N / 4 + DS_NT - 1) / DS_NT, F4 = N * N / 4;
    static_assert(F4 % DS_NT == 0 || F4 < DS_NT, "every thread owns NV quads");
    // R Z - pan of the sample (k_dstep_pre: one pixelwise launch in front; as C + 1 planes read by each of the sample's C workgroups it was
    // 320 KB per CU at ~15 bytes per cycle and CU = half of this kernel's time) and the plane itself: every load requested here, the R term
    // consumed by the update at the very end
    float4 zc[NV], rzp[NV];
    const float4* __restrict__ zp = reinterpret_cast<const float4*>(a.z + (size_t)plane * N * N);
    const float4* __restrict__ prp = reinterpret_cast<const float4*>(a.pr + (size_t)b * N * N);
#pragma unroll
    for (int k = 0; k < NV; ++k) zc[k] = zp[min(tid + DS_NT * k, F4 - 1)];
#pragma unroll
    for (int k = 0; k < NV; ++k) rzp[k] = prp[min(tid + DS_NT * k, F4 - 1)];
    auto put_z = [&](int k) {
        const int f = tid + DS_NT * k;
        if (f < F4) {
            const float o[4] = {zc[k].x, zc[k].y, zc[k].z, zc[k].w};
            ds_put_src<N, 0>(Zs + (f / (N / 4)) * P0, f % (N / 4), o);
        }
    };
    // every parameter of the plane's channel, once, in front of everything (scalar loads: as four dependent fetches in front of the four convs
    // they were ~1 000 cycles each)
    float wD1[9], wD3[9], wT1[9], wT3[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { wD1[k] = a.d1w[c * 9 + k]; wD3[k] = a.d3w[c * 9 + k]; wT1[k] = a.dt1w[c * 9 + k]; wT3[k] = a.dt3w[c * 9 + k]; }
    const float bD1 = a.d1b[c], bD3 = a.d3b[c], bT1 = a.dt1b[c], bT3 = a.dt3b[c];
    const float rtw_c = a.rtw[c], rtb_c = a.rtb[c], eta = a.eta[0];
    static_assert(Q * Q / 4 <= DS_NT, "one ms quad per thread");
    const float4 msv = reinterpret_cast<const float4*>(a.ms + (size_t)plane * Q * Q)[min(tid, Q * Q / 4 - 1)];
    ds_zero_rows<H, PH>(U1);
#pragma unroll
    for (int k = 0; k < NV; ++k) put_z(k);
    __syncthreads();
    DSTAMP(stamp_i++);
    // ---- D, first half: x0.5 -> dw3 = t1
    ds_xhalf<N, N, P0, H + 4>(Zs, R1);
    __syncthreads();
    DSTAMP(stamp_i++);
    ds_ycon<0, N, H, H + 4, PH>(R1, U1);
    ds_zero_rows<N, P0>(U4);       // Zs is dead
    __syncthreads();
    DSTAMP(stamp_i++);
    ds_zero_rows<Q, PQ>(U2);       // the first row-contracted image is dead: the quarter-resolution images live in its buffer
    {
        float* __restrict__ t1 = a.t1 + (size_t)plane * H * H;
        for (int i = tid; i < H * H / 4; i += DS_NT) {
            const int oy = i / (H / 4), m = i % (H / 4);
            float o[4];
            ds_dwq<PH, H / 4>(U1, oy, m, wD1, o);
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] += bD1;
            ds_put_src<H, 0>(V1 + oy * PH, m, o);
            reinterpret_cast<float4*>(t1)[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    DSTAMP(stamp_i++);
    // ---- D, second half: x0.5 -> dw3 - ms = r
    ds_xhalf<H, H, PH, Q + 4>(V1, Hq);
    __syncthreads();
    DSTAMP(stamp_i++);
    ds_ycon<0, H, Q, Q + 4, PQ>(Hq, U2);
    __syncthreads();
    DSTAMP(stamp_i++);
    static_assert((Q * Q / 4) % 64 == 0, "whole waves in the quarter-resolution conv");
    if (tid < Q * Q / 4) {
        const int oy = tid / (Q / 4), m = tid % (Q / 4);
        float o[4];
        ds_dwq<PQ, Q / 4>(U2, oy, m, wD3, o);
        const float sub[4] = {msv.x, msv.y, msv.z, msv.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) { o[u] += bD3; o[u] -= sub[u]; }
        ds_put_src<Q, 2>(Rq + oy * PQ, m, o);
        reinterpret_cast<float4*>(a.r + (size_t)plane * Q * Q)[tid] = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
    DSTAMP(stamp_i++);
    // ---- DT, first half: x2 -> dw3 = s1
    ds_xdouble<Q, Q, PQ, H + 4>(Rq, Hq);
    __syncthreads();
    DSTAMP(stamp_i++);
    ds_ycon<1, Q, H, H + 4, PH>(Hq, U1);
    __syncthreads();
    DSTAMP(stamp_i++);
    {
        float* __restrict__ s1 = a.s1 + (size_t)plane * H * H;
        for (int i = tid; i < H * H / 4; i += DS_NT) {
            const int oy = i / (H / 4), m = i % (H / 4);
            float o[4];
            ds_dwq<PH, H / 4>(U1, oy, m, wT1, o);
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] += bT1;
            ds_put_src<H, 2>(V1 + oy * PH, m, o);
            reinterpret_cast<float4*>(s1)[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
    DSTAMP(stamp_i++);
    // ---- DT, second half: x2 -> dw3, then the update   Z <- Z - eta (ms_term + RT(R Z - pan))      unlg_former.py:59-61
    ds_xdouble<H, H, PH, N + 4>(V1, R1);
    __syncthreads();
    DSTAMP(stamp_i++);
    ds_ycon<1, H, N, N + 4, P0>(R1, U4);
    __syncthreads();
    DSTAMP(stamp_i++);
    {
        float4* __restrict__ zo = reinterpret_cast<float4*>(a.zout + (size_t)plane * N * N);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int f = min(tid + DS_NT * k, F4 - 1);
            const float zin[4] = {zc[k].x, zc[k].y, zc[k].z, zc[k].w};
            const float rz[4] = {rzp[k].x, rzp[k].y, rzp[k].z, rzp[k].w};
            float o[4];
            ds_dwq<P0, N / 4>(U4, f / (N / 4), f % (N / 4), wT3, o);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v = o[u];
                v += bT3;
                const float pan_term = rtw_c * rz[u] + rtb_c;
                v = zin[u] - eta * (v + pan_term);
                o[u] = v;
            }
            if (tid + DS_NT * k < F4) zo[f] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    DSTAMP(stamp_i++);
    DSTAMP_FLUSH();
}

// ---- pixelwise launches in front of the plane kernels: the only place where the C planes of a sample meet (R / RT are 1x1 convs C -> 1 -> C).
// lane = a quad of pixels, all C channels of it in registers.
template <int C>
__global__ __launch_bounds__(256) void k_dstep_pre_fwd(const float* __restrict__ z, const float* __restrict__ pan, const float* __restrict__ rwp,
                                                       const float* __restrict__ rbp, float* __restrict__ pr, int hw4, long total4) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= total4) return;
    const long b = i / hw4;
    const int f = (int)(i - b * hw4);
    const float4* __restrict__ zb = reinterpret_cast<const float4*>(z) + b * C * hw4 + f;
    float4 zz[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) zz[cc] = zb[(size_t)cc * hw4];
    const float4 pv = reinterpret_cast<const float4*>(pan)[i];
    const float rb = rbp[0];
    float4 r = make_float4(rb, rb, rb, rb);
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const float rw = rwp[cc];
        r.x += rw * zz[cc].x; r.y += rw * zz[cc].y; r.z += rw * zz[cc].z; r.w += rw * zz[cc].w;
    }
    reinterpret_cast<float4*>(pr)[i] = make_float4(r.x - pv.x, r.y - pv.y, r.z - pv.z, r.w - pv.w);
}

// sum over the wave (all 64 lanes active): total in lane 63.  DPP only: two quad_perm steps, two row rotations, two row broadcasts
__device__ __forceinline__ float ds_wave_sum63(float v) {
    auto dpp = [](float x, auto ctrl, auto rmask) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(rmask)::value, 0xF, false));
    };
    using std::integral_constant;
    v += dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xF>{});    // quad_perm [1,0,3,2]
    v += dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xF>{});    // quad_perm [2,3,0,1]
    v += dpp(v, integral_constant<int, 0x124>{}, integral_constant<int, 0xF>{});   // row_ror:4
    v += dpp(v, integral_constant<int, 0x128>{}, integral_constant<int, 0xF>{});   // row_ror:8  -> every lane of a row holds the row's sum
    v += dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xA>{});   // row_bcast:15 into rows 1 and 3
    v += dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xC>{});   // row_bcast:31 into rows 2 and 3
    return v;
}

// Backward, pixelwise part: everything of the step's top that needs no neighbour (autograd of unlg_former.py:59-61 around the DT chain):
//   dpr = -eta sum_c RTw[c] g_c ;  dZ_c (direct part) = g_c + Rw[c] dpr  -> dz (the plane kernel adds the chain's part on top)
//   per channel: sum gm (d bias(DT.3), d RT.bias), sum -g (RTw[c] pr + RTb[c]) (d eta, the part without the chain), sum gm pr (d RT.weight),
//   sum dpr z_c (d R.weight), sum dpr (d R.bias; channel 0's row), with gm = -eta g, pr = R Z - pan.  One partial row [C][5] per workgroup;
// the plane kernel of (sample, channel) folds the sample's rows into its own (no reduce job of 512 x C slices; and two jobs adding into
// d eta would race inside the deferred reduce launch).
template <int C>
__global__ __launch_bounds__(256) void k_dstep_pre_bwd(DstepPreBwdArgs a) {
    __shared__ float red[4][C * 5];   // per channel: [sum gm | d eta part | d RT.weight | d R.weight | d R.bias]
    const long i = blockIdx.x * 256L + threadIdx.x;      // total4 is a multiple of 256 (whole waves: the wave sums need every lane)
    const long b = i / a.hw4;
    const int f = (int)(i - b * a.hw4);
    const float4* __restrict__ zb = reinterpret_cast<const float4*>(a.z) + b * C * a.hw4 + f;
    const float4* __restrict__ gb = reinterpret_cast<const float4*>(a.g) + b * C * a.hw4 + f;
    float4 zz[C], gg[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) { zz[cc] = zb[(size_t)cc * a.hw4]; gg[cc] = gb[(size_t)cc * a.hw4]; }
    const float4 pv = reinterpret_cast<const float4*>(a.pan)[i];
    const float eta = a.eta[0], rb = a.rb[0];
    float r[4] = {rb, rb, rb, rb}, d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const float rw = a.rw[cc], rtw = a.rtw[cc];
        const float zv[4] = {zz[cc].x, zz[cc].y, zz[cc].z, zz[cc].w}, gv[4] = {gg[cc].x, gg[cc].y, gg[cc].z, gg[cc].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) { r[u] += rw * zv[u]; d[u] += rtw * gv[u]; }
    }
    const float pn[4] = {pv.x, pv.y, pv.z, pv.w};
    float pr[4], dpr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { pr[u] = r[u] - pn[u]; dpr[u] = -eta * d[u]; }
    float4* __restrict__ dzb = reinterpret_cast<float4*>(a.dz) + b * C * a.hw4 + f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const float rw = a.rw[cc], rtw = a.rtw[cc], rtb = a.rtb[cc];
        const float zv[4] = {zz[cc].x, zz[cc].y, zz[cc].z, zz[cc].w}, gv[4] = {gg[cc].x, gg[cc].y, gg[cc].z, gg[cc].w};
        float o[4], p[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float gm = -eta * gv[u];
            const float pt = rtw * pr[u] + rtb;
            p[0] += gm;
            p[1] += -gv[u] * pt;
            p[2] += gm * pr[u];
            p[3] += dpr[u] * zv[u];
            p[4] += cc == 0 ? dpr[u] : 0.f;     // d R.bias: once per pixel (channel 0's row; the reduce sums it over channels)
            o[u] = gv[u] + rw * dpr[u];
        }
        dzb[(size_t)cc * a.hw4] = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float t = ds_wave_sum63(p[k]);
            if (lane == 63) red[wave][cc * 5 + k] = t;
        }
    }
    __syncthreads();
    if (threadIdx.x < C * 5) {
        const int t = threadIdx.x;
        a.part[(size_t)blockIdx.x * C * 5 + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    }
}

// ---- backward of the chain, one workgroup per plane (the forward's mirror; its intermediates t1 / r / s1 come from the forward) ----
// 3 x 6 window of the quad (oy, 4m .. 4m+3) of a conv source (zero rows above / below; neighbour columns through DPP, zero at the row ends)
template <int UP, int NQ>
__device__ __forceinline__ void ds_win(const float* __restrict__ U, int oy, int m, float (&in)[3][6]) {
    static_assert(64 % NQ == 0, "whole rows per wave");
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const float4 c = *reinterpret_cast<const float4*>(U + (oy + dy - 1) * UP + 4 * m);
        const float lo = ds_shr1(0.f, c.w), hi = ds_shl1(0.f, c.x);
        in[dy][0] = m == 0 ? 0.f : lo; in[dy][1] = c.x; in[dy][2] = c.y; in[dy][3] = c.z; in[dy][4] = c.w; in[dy][5] = m == NQ - 1 ? 0.f : hi;
    }
}
// one depthwise-conv stage of the backward on a quad: inU = window of the conv's input, inG = window of the gradient wrt its output.
//   part[k] += sum_u g_u U(u + tap k) (k < 9), part[9] += sum_u g_u ; gi_u = (dw^T g)_u
__device__ __forceinline__ void ds_dw_bwd_quad(const float (&inU)[3][6], const float (&inG)[3][6], const float (&w)[9], float (&part)[10],
                                               float (&gi)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float g = inG[1][u + 1];
        float acc = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                acc += w[dy * 3 + dx] * inG[2 - dy][u + 2 - dx];
                part[dy * 3 + dx] += g * inU[dy][u + dx];
            }
        part[9] += g;
        gi[u] = acc;
    }
}
// Adjoints of the resamplers, written against the PADDED input index (F.interpolate clamps its taps: in_p[i] = in[clamp(i)], so the adjoint is
// the plain transpose onto i in [-PAD, n + PAD) followed by folding the pads onto the border elements).  With x0.5 (w = cubic(1/2)):
//   gin_p[i] = i even: w1 G[i/2] + w3 G[i/2 - 1] ; i odd: w0 G[(i+1)/2] + w2 G[(i-1)/2]            i in [-1, n]
// and x2 (we = cubic(3/4) for even outputs, wo = cubic(1/4) for odd ones):
//   gin_p[i] = wo3 G[2i-3] + we3 G[2i-2] + wo2 G[2i-1] + we2 G[2i] + wo1 G[2i+1] + we1 G[2i+2] + wo0 G[2i+3] + we0 G[2i+4]      i in [-2, n + 1]
// (G = 0 outside its range).  Along x the quad's neighbours come through DPP from the lanes beside it; along y from LDS rows.
// x2 adjoint along x: gradient quad g (columns 4m .. 4m+3 of a row 2n wide, NQ quads) -> outputs 2m, 2m+1 of the row n wide
template <int NQ>
__device__ __forceinline__ void ds_adj2_x(const float (&g)[4], int m, float (&o)[2]) {
    static_assert(64 % NQ == 0, "whole rows per wave");
    float we[4], wo[4];
    cubic_w(0.75f, we);
    cubic_w(0.25f, wo);
    float py = ds_shr1(0.f, g[1]), pz = ds_shr1(0.f, g[2]), pw = ds_shr1(0.f, g[3]);
    float nx = ds_shl1(0.f, g[0]), ny = ds_shl1(0.f, g[1]), nz = ds_shl1(0.f, g[2]);
    if (m == 0) { py = 0.f; pz = 0.f; pw = 0.f; }
    if (m == NQ - 1) { nx = 0.f; ny = 0.f; nz = 0.f; }
    // padded output i reads G[2i - 3 .. 2i + 4]; in quad coordinates (G[4m + k] = g[k]): i = 2m: k = -3 .. 4 ; i = 2m + 1: k = -1 .. 6
    o[0] = wo[3] * py + we[3] * pz + wo[2] * pw + we[2] * g[0] + wo[1] * g[1] + we[1] * g[2] + wo[0] * g[3] + we[0] * nx;
    o[1] = wo[3] * pw + we[3] * g[0] + wo[2] * g[1] + we[2] * g[2] + wo[1] * g[3] + we[1] * nx + wo[0] * ny + we[0] * nz;
    if (m == 0) o[0] += we[0] * g[0] + (wo[0] * g[1] + we[1] * g[0] + we[0] * g[2]);                 // pads i = -2 and -1 onto 0
    if (m == NQ - 1) o[1] += (wo[3] * g[1] + we[3] * g[2] + wo[2] * g[3]) + wo[3] * g[3];            // pads i = n and n + 1 onto n - 1
}
// x2 adjoint along y: T [2n rows][pitch TP] -> quad (i, 4m ..) of the n-row image
template <int NN /* n */, int TP>
__device__ __forceinline__ float4 ds_adj2_y(const float* __restrict__ T, int i, int m) {
    float we[4], wo[4];
    cubic_w(0.75f, we);
    cubic_w(0.25f, wo);
    const float ck[8] = {wo[3], we[3], wo[2], we[2], wo[1], we[1], wo[0], we[0]};
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto tap = [&](int ip) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = 2 * ip - 3 + k;
            const float c = (unsigned)row < (unsigned)(2 * NN) ? ck[k] : 0.f;
            const float4 v = *reinterpret_cast<const float4*>(T + clampi(row, 0, 2 * NN - 1) * TP + 4 * m);
            acc.x += c * v.x; acc.y += c * v.y; acc.z += c * v.z; acc.w += c * v.w;
        }
    };
    tap(i);
    if (i == 0) { tap(-2); tap(-1); }
    if (i == NN - 1) { tap(NN); tap(NN + 1); }
    return acc;
}
// x0.5 adjoint along x: gradient quad g (columns 4m .. 4m+3 of a row n/2 wide, NQ quads) -> outputs 8m .. 8m+7 of the row n wide
template <int NQ>
__device__ __forceinline__ void ds_adjh_x(const float (&g)[4], int m, float (&o)[8]) {
    static_assert(64 % NQ == 0, "whole rows per wave");
    float w[4];
    cubic_w(0.5f, w);
    float pw = ds_shr1(0.f, g[3]), nx = ds_shl1(0.f, g[0]);
    if (m == 0) pw = 0.f;
    if (m == NQ - 1) nx = 0.f;
    const float e[6] = {pw, g[0], g[1], g[2], g[3], nx};   // e[k] = G[4m - 1 + k]
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        o[2 * u] = w[1] * e[u + 1] + w[3] * e[u];          // i = 8m + 2u     : w1 G[4m + u] + w3 G[4m + u - 1]
        o[2 * u + 1] = w[0] * e[u + 2] + w[2] * e[u + 1];  // i = 8m + 2u + 1 : w0 G[4m + u + 1] + w2 G[4m + u]
    }
    if (m == 0) o[0] += w[0] * g[0];            // pad i = -1 onto 0
    if (m == NQ - 1) o[7] += w[3] * g[3];       // pad i = n onto n - 1
}
// x0.5 adjoint along y: T [n/2 rows][pitch TP] -> quad (i, 4m ..) of the n-row image
template <int NN /* n */, int TP>
__device__ __forceinline__ float4 ds_adjh_y(const float* __restrict__ T, int i, int m) {
    float w[4];
    cubic_w(0.5f, w);
    const int ra = (i & 1) ? (i + 1) / 2 : i / 2, rb = ra - 1;          // the two gradient rows of padded row i; weights by parity
    float ca = (i & 1) ? w[0] : w[1], cb = (i & 1) ? w[2] : w[3];
    if (ra >= NN / 2) ca = 0.f;
    if (rb < 0) cb = 0.f;
    const float4 va = *reinterpret_cast<const float4*>(T + clampi(ra, 0, NN / 2 - 1) * TP + 4 * m);
    const float4 vb = *reinterpret_cast<const float4*>(T + clampi(rb, 0, NN / 2 - 1) * TP + 4 * m);
    float4 acc = make_float4(ca * va.x + cb * vb.x, ca * va.y + cb * vb.y, ca * va.z + cb * vb.z, ca * va.w + cb * vb.w);
    if (i == 0) { acc.x += w[0] * va.x; acc.y += w[0] * va.y; acc.z += w[0] * va.z; acc.w += w[0] * va.w; }                  // pad -1: w0 T[0]  (ra = 0)
    if (i == NN - 1) { acc.x += w[3] * va.x; acc.y += w[3] * va.y; acc.z += w[3] * va.z; acc.w += w[3] * va.w; }             // pad n : w3 T[n/2 - 1]  (ra = n/2 - 1)
    return acc;
}

template <int N>
struct DsLB {   // LDS carve of the backward kernel (floats): two plane-sized regions + the cross-wave sums
    static constexpr int H = N / 2, Q = N / 4;
    static constexpr int P0 = N + 8, PH = H + 8, PQ = Q + 8;
    static constexpr int RA = (N + 2) * P0, RB = (N + 2) * P0;
    static constexpr int RED = 4 * 16 * 10;
    static constexpr int FLOATS = RA + RB + RED;
    // region B while the half-resolution DT stage runs: G1 | Vr | Hx3 | U3
    static constexpr int B_G1 = 0, B_VR = B_G1 + (H + 2) * PH, B_HX3 = B_VR + Q * PQ, B_U3 = B_HX3 + Q * (H + 4);
    static_assert(B_U3 + (H + 2) * PH <= RB, "DT stage fits region B");
    // region A while the quarter-resolution D stage is prepared: T3 | G2 | Vt | Hx2 | U2
    static constexpr int A_T3 = 0, A_G2 = A_T3 + H * (Q + 4), A_VT = A_G2 + (Q + 2) * PQ, A_HX2 = A_VT + H * PH, A_U2 = A_HX2 + H * (Q + 4);
    static_assert(A_U2 + (Q + 2) * PQ <= RA, "D stage fits region A");
    // region B for the last stage: T2 | G3 | Hx1 ; region A: Zs, then U1 | T1
    static constexpr int B_T2 = 0, B_G3 = B_T2 + Q * (H + 4), B_HX1 = B_G3 + (H + 2) * PH;
    static_assert(B_HX1 + N * (H + 4) <= RB, "last stage fits region B");
    static constexpr int A_U1 = 0, A_T1 = A_U1 + (H + 2) * PH;
    static_assert(A_T1 + H * (N + 4) <= RA, "last stage fits region A");
};

template <int N, int C>
__global__ __launch_bounds__(DS_NT) void k_dstep_bwd(DstepBwdArgs a) {
    using L = DsLB<N>;
    constexpr int H = L::H, Q = L::Q, P0 = L::P0, PH = L::PH, PQ = L::PQ;
    constexpr int NV = (N * N / 4 + DS_NT - 1) / DS_NT, F4 = N * N / 4, NQ0 = N / 4, NQH = H / 4, NQQ = Q / 4;
    static_assert(F4 % DS_NT == 0 && H * H / 4 <= DS_NT && (H * H / 4) % 64 == 0 && (Q * Q / 4) % 64 == 0, "whole waves everywhere");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* RA = sm;
    float* RB = sm + L::RA;
    float* red = RB + L::RB;                          // [4 stages][16 waves][10]
    DSTAMP_DECL
    DSTAMP(stamp_i++);
    const int plane = blockIdx.x, c = plane % C;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // the channel's parameters first (scalar loads: their two dependent fetches -- pointer, then data -- run under the first vector load's round trip)
    const float eta = a.eta[0];
    float wD1[9], wD3[9], wT1[9], wT3[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { wD1[k] = a.d1w[c * 9 + k]; wD3[k] = a.d3w[c * 9 + k]; wT1[k] = a.dt1w[c * 9 + k]; wT3[k] = a.dt3w[c * 9 + k]; }
    const float bT3 = a.dt3b[c];
    // every load of the first stages is requested before anything else, in the order of use (the counter of outstanding loads is in-order)
    const float4 s1q = reinterpret_cast<const float4*>(a.s1 + (size_t)plane * H * H)[min(tid, H * H / 4 - 1)];
    const float4* __restrict__ gp = reinterpret_cast<const float4*>(a.g + (size_t)plane * N * N);
    float4 gq[NV];                                    // parked in registers until region B is free (it holds the row-contracted image first)
#pragma unroll
    for (int k = 0; k < NV; ++k) gq[k] = gp[tid + DS_NT * k];
    const float4 rq = reinterpret_cast<const float4*>(a.r + (size_t)plane * Q * Q)[min(tid, Q * Q / 4 - 1)];
    const float4 t1q = reinterpret_cast<const float4*>(a.t1 + (size_t)plane * H * H)[min(tid, H * H / 4 - 1)];
    // (the parameters pinned HERE, behind the requests: left to itself the compiler sinks their scalar loads behind the first vector load's wait)
#define DS_PIN9(w) asm volatile("" :: "s"(w[0]), "s"(w[1]), "s"(w[2]), "s"(w[3]), "s"(w[4]), "s"(w[5]), "s"(w[6]), "s"(w[7]), "s"(w[8]))
    DS_PIN9(wT3); DS_PIN9(wT1); DS_PIN9(wD3); DS_PIN9(wD1);
    asm volatile("" :: "s"(eta), "s"(bT3));
    DSTAMP(stamp_i++);
    auto put4 = [](float4 v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; };
    // the partial sums of a stage: wave totals into red[stage][wave][k]; summed over the waves (fixed order) by sum_stage after a barrier
    auto wave_part = [&](int stage, const float (&part)[10]) {
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float t = ds_wave_sum63(part[k]);
            if (lane == 63) red[(stage * 16 + wave) * 10 + k] = t;
        }
    };
    auto sum_stage = [&](int stage, int nwaves, float scale_w, float* __restrict__ dst) {   // scale_w: slots 0 .. 8; slot 9 as it is
        if (tid < 10) {
            float t = 0.f;
            for (int wv = 0; wv < nwaves; ++wv) t += red[(stage * 16 + wv) * 10 + tid];
            dst[(size_t)plane * 10 + tid] = tid < 9 ? scale_w * t : t;
        }
    };
    // ---- U4 = x2(s1) with its zero rows (region A), g (region B)
    float* Vs = RA + 4;
    if (tid < H * H / 4) { float o[4]; put4(s1q, o); ds_put_src<H, 2>(Vs + (tid / NQH) * PH, tid % NQH, o); }
    DSTAMP(stamp_i++);
    __syncthreads();
    DSTAMP(stamp_i++);
    ds_xdouble<H, H, PH, N + 4>(Vs, RB);
    __syncthreads();
    DSTAMP(stamp_i++);
    float* U4 = RA + P0 + 4;
    ds_ycon<1, H, N, N + 4, P0>(RB, U4);
    ds_zero_rows<N, P0>(U4);
    __syncthreads();
    DSTAMP(stamp_i++);
    float* Gs = RB + P0 + 4;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int f = tid + DS_NT * k;
        *reinterpret_cast<float4*>(Gs + (f / NQ0) * P0 + 4 * (f % NQ0)) = make_float4(gq[k].x, gq[k].y, gq[k].z, gq[k].w);   // (a struct copy becomes a memcpy through a stack slot)
    }
    ds_zero_rows<N, P0>(Gs);
    __syncthreads();
    DSTAMP(stamp_i++);
    // ---- top: through the DT.3 conv (its weight gradient, the chain's part of d eta) and the x2 adjoint along x.  The row-contracted
    // gradient T [N][H + 4] of iteration k is written over the U4 rows iteration k has finished with (all threads walk the plane in row order)
    {
        float part[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) part[k] = 0.f;
        float* T = RA;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int f = tid + DS_NT * k, y = f / NQ0, m = f % NQ0;
            float inU[3][6], inG[3][6], gi[4], o[2];
            ds_win<P0, NQ0>(U4, y, m, inU);
            ds_win<P0, NQ0>(Gs, y, m, inG);
            ds_dw_bwd_quad(inU, inG, wT3, part, gi);
            ds_adj2_x<NQ0>(gi, m, o);
            __syncthreads();
            *reinterpret_cast<float2*>(T + y * (H + 4) + 2 * m) = make_float2(o[0], o[1]);
        }
        wave_part(0, part);
    }
    __syncthreads();
    DSTAMP(stamp_i++);
    // ---- gs1 = -eta adj_x2(...) (64 x 64, conv source G1 in region B), and U3 = x2(r)
    float* G1 = RB + L::B_G1 + PH + 4;
    float* Vr = RB + L::B_VR + 4;
    float* Hx3 = RB + L::B_HX3;
    float* U3 = RB + L::B_U3 + PH + 4;
    if (tid < H * H / 4) {
        const int i = tid / NQH, m = tid % NQH;
        const float4 t = ds_adj2_y<H, H + 4>(RA, i, m);
        *reinterpret_cast<float4*>(G1 + i * PH + 4 * m) = make_float4(-eta * t.x, -eta * t.y, -eta * t.z, -eta * t.w);
    }
    ds_zero_rows<H, PH>(G1);
    if (tid < Q * Q / 4) { float o[4]; put4(rq, o); ds_put_src<Q, 2>(Vr + (tid / NQQ) * PQ, tid % NQQ, o); }
    __syncthreads();
    DSTAMP(stamp_i++);
    ds_xdouble<Q, Q, PQ, H + 4>(Vr, Hx3);
    __syncthreads();
    DSTAMP(stamp_i++);
    ds_ycon<1, Q, H, H + 4, PH>(Hx3, U3);
    ds_zero_rows<H, PH>(U3);
    __syncthreads();
    DSTAMP(stamp_i++);
    // ---- DT.1 conv stage (64 x 64), x2 adjoint along x -> T3 [H][Q + 4] (region A); t1 as x0.5 source into region A
    float* T3 = RA + L::A_T3;
    float* G2 = RA + L::A_G2 + PQ + 4;
    float* Vt = RA + L::A_VT + 4;
    float* Hx2 = RA + L::A_HX2;
    float* U2 = RA + L::A_U2 + PQ + 4;
    // the plane's Z (source of the last stage's conv input) is requested here and parked in registers until region A is free for it
    const float4* __restrict__ zp = reinterpret_cast<const float4*>(a.z + (size_t)plane * N * N);
    float4 zq[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) zq[k] = zp[tid + DS_NT * k];
    if (tid < H * H / 4) {
        const int i = tid / NQH, m = tid % NQH;
        float part[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) part[k] = 0.f;
        float inU[3][6], inG[3][6], gi[4], o[2];
        ds_win<PH, NQH>(U3, i, m, inU);
        ds_win<PH, NQH>(G1, i, m, inG);
        ds_dw_bwd_quad(inU, inG, wT1, part, gi);
        ds_adj2_x<NQH>(gi, m, o);
        *reinterpret_cast<float2*>(T3 + i * (Q + 4) + 2 * m) = make_float2(o[0], o[1]);
        wave_part(1, part);
        float t[4];
        put4(t1q, t);
        ds_put_src<H, 0>(Vt + i * PH, m, t);
    }
    __syncthreads();
    DSTAMP(stamp_i++);
    sum_stage(1, H * H / 4 / 64, 1.0f, a.part_dt1);
    // ---- gr = adj_x2 along y (32 x 32, conv source G2); U2 = x0.5(t1)
    if (tid < Q * Q / 4) {
        const int i = tid / NQQ, m = tid % NQQ;
        *reinterpret_cast<float4*>(G2 + i * PQ + 4 * m) = ds_adj2_y<Q, Q + 4>(T3, i, m);
    }
    ds_zero_rows<Q, PQ>(G2);
    ds_xhalf<H, H, PH, Q + 4>(Vt, Hx2);
    __syncthreads();
    DSTAMP(stamp_i++);
    ds_ycon<0, H, Q, Q + 4, PQ>(Hx2, U2);
    ds_zero_rows<Q, PQ>(U2);
    __syncthreads();
    DSTAMP(stamp_i++);
    // ---- D.3 conv stage (32 x 32), x0.5 adjoint along x -> T2 [Q][H + 4] (region B)
    float* T2 = RB + L::B_T2;
    float* G3 = RB + L::B_G3 + PH + 4;
    float* Hx1 = RB + L::B_HX1;
    if (tid < Q * Q / 4) {
        const int i = tid / NQQ, m = tid % NQQ;
        float part[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) part[k] = 0.f;
        float inU[3][6], inG[3][6], gi[4], o[8];
        ds_win<PQ, NQQ>(U2, i, m, inU);
        ds_win<PQ, NQQ>(G2, i, m, inG);
        ds_dw_bwd_quad(inU, inG, wD3, part, gi);
        ds_adjh_x<NQQ>(gi, m, o);
        *reinterpret_cast<float4*>(T2 + i * (H + 4) + 8 * m) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(T2 + i * (H + 4) + 8 * m + 4) = make_float4(o[4], o[5], o[6], o[7]);
        wave_part(2, part);
    }
    // region A is free: Z as the x0.5 source of the last conv input
    float* Zs = RA + 4;
    __syncthreads();
    DSTAMP(stamp_i++);
    sum_stage(2, Q * Q / 4 / 64, 1.0f, a.part_d3);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int f = tid + DS_NT * k;
        float o[4];
        put4(zq[k], o);
        ds_put_src<N, 0>(Zs + (f / NQ0) * P0, f % NQ0, o);
    }
    // ---- gt1 = adj_x0.5 along y (64 x 64, conv source G3)
    if (tid < H * H / 4) {
        const int i = tid / NQH, m = tid % NQH;
        *reinterpret_cast<float4*>(G3 + i * PH + 4 * m) = ds_adjh_y<H, H + 4>(T2, i, m);
    }
    ds_zero_rows<H, PH>(G3);
    __syncthreads();
    DSTAMP(stamp_i++);
    ds_xhalf<N, N, P0, H + 4>(Zs, Hx1);
    __syncthreads();
    DSTAMP(stamp_i++);
    float* U1 = RA + L::A_U1 + PH + 4;
    float* T1 = RA + L::A_T1;
    ds_ycon<0, N, H, H + 4, PH>(Hx1, U1);
    ds_zero_rows<H, PH>(U1);
    // the direct part of dZ (the pixelwise kernel's) is requested here, two phases in front of its use
    float4* __restrict__ dzp = reinterpret_cast<float4*>(a.dz + (size_t)plane * N * N);
    float4 dq[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) dq[k] = dzp[tid + DS_NT * k];
    __syncthreads();
    DSTAMP(stamp_i++);
    // ---- D.1 conv stage (64 x 64), x0.5 adjoint along x -> T1 [H][N + 4] (region A)
    if (tid < H * H / 4) {
        const int i = tid / NQH, m = tid % NQH;
        float part[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) part[k] = 0.f;
        float inU[3][6], inG[3][6], gi[4], o[8];
        ds_win<PH, NQH>(U1, i, m, inU);
        ds_win<PH, NQH>(G3, i, m, inG);
        ds_dw_bwd_quad(inU, inG, wD1, part, gi);
        ds_adjh_x<NQH>(gi, m, o);
        *reinterpret_cast<float4*>(T1 + i * (N + 4) + 8 * m) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(T1 + i * (N + 4) + 8 * m + 4) = make_float4(o[4], o[5], o[6], o[7]);
        wave_part(3, part);
    }
    __syncthreads();
    DSTAMP(stamp_i++);
    sum_stage(3, H * H / 4 / 64, 1.0f, a.part_d1);
    // ---- dZ = direct part + adj_x0.5 along y
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int f = tid + DS_NT * k;
        const float4 t = ds_adjh_y<N, N + 4>(T1, f / NQ0, f % NQ0);
        dzp[f] = make_float4(dq[k].x + t.x, dq[k].y + t.y, dq[k].z + t.z, dq[k].w + t.w);
    }
    // ---- the top stage's row, last: its sums have been waiting in `red` since the top loop, and the few threads that combine them with the
    // pixelwise kernel's rows would hold every barrier behind them up (9 000 cycles where this stood).  Row = [d DT.3 weight (9) | d eta |
    // sum gm (d DT.3 bias, d RT.bias) | d RT.weight | d R.weight | d R.bias]:
    //   d eta = sum -g (DT.3(up(s1)) + pt); the chain's part needs no pass of its own,
    //   sum_pix g (w . U-window + b) = sum_k w[k] (sum_pix g U(+tap k)) + b sum_pix g = w . (the conv's raw weight-gradient sums) + b (slot 9)
    if (tid < 14) {
        constexpr int NWG = N * N / 4 / 256;
        float v;
        if (tid < 9) {
            float t = 0.f;
            for (int wv = 0; wv < 16; ++wv) t += red[wv * 10 + tid];
            v = -eta * t;                                  // the conv saw the raw g: its weight-gradient slots carry the -eta of gm = -eta g
        } else {
            const int slot = tid == 9 ? 1 : (tid == 10 ? 0 : tid - 9);   // row of the pixelwise kernel: [sum gm | d eta | d RTw | d Rw | d Rb]
            const float* __restrict__ pe = a.part_pre + ((size_t)(plane / C) * NWG * C + c) * 5 + slot;
            float pv[NWG];
#pragma unroll
            for (int j = 0; j < NWG; ++j) pv[j] = pe[j * C * 5];
            v = 0.f;
#pragma unroll
            for (int j = 0; j < NWG; ++j) v += pv[j];
            if (tid == 9) {
                float ch = 0.f;
                for (int k = 0; k < 10; ++k) {
                    float t = 0.f;
                    for (int wv = 0; wv < 16; ++wv) t += red[wv * 10 + k];
                    ch += (k < 9 ? a.dt3w[c * 9 + k] : bT3) * t;
                }
                v -= ch;
            }
        }
        a.part_top[(size_t)plane * 14 + tid] = v;
    }
    DSTAMP(stamp_i++);
    DSTAMP_FLUSH();
}

template <int N, int C>
int launch_fwd_t(const DstepFwdArgs& a, hipStream_t s) {
    constexpr size_t lds = DsL<N>::FLOATS * sizeof(float);
    static_assert(lds <= 160 * 1024, "plane does not fit");
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_dstep_fwd<N, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { lg_set_error("dstep_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    const long total4 = (long)a.B * N * N / 4;
    k_dstep_pre_fwd<C><<<(int)((total4 + 255) / 256), 256, 0, s>>>(a.z, a.pan, a.rw, a.rb, a.pr, N * N / 4, total4);
    LG_CHECK_LAUNCH();
    k_dstep_fwd<N, C><<<a.B * C, DS_NT, lds, s>>>(a);
    LG_CHECK_LAUNCH();
    return 0;
}
template <int N, int C>
int launch_bwd_t(const DstepBwdArgs& a, hipStream_t s) {
    constexpr size_t lds = DsLB<N>::FLOATS * sizeof(float);
    static_assert(lds <= 160 * 1024, "plane does not fit");
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_dstep_bwd<N, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { lg_set_error("dstep_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    DstepPreBwdArgs p;
    p.z = a.z; p.g = a.g; p.pan = a.pan; p.dz = a.dz; p.rw = a.rw; p.rb = a.rb; p.rtw = a.rtw; p.rtb = a.rtb; p.eta = a.eta;
    p.part = a.part_pre; p.hw4 = N * N / 4;
    static_assert((N * N / 4) % 256 == 0, "whole workgroups per sample");
    k_dstep_pre_bwd<C><<<a.B * (N * N / 4 / 256), 256, 0, s>>>(p);
    LG_CHECK_LAUNCH();
    k_dstep_bwd<N, C><<<a.B * C, DS_NT, lds, s>>>(a);
    LG_CHECK_LAUNCH();
    return 0;
}
}  // namespace

bool lg_plan::dstep_fused(int h, int w) const { return !dstep_tiles && dstep_fused_ok(cfg.C, h, w); }
bool dstep_fused_ok(int C, int H, int W) { return H == W && (H == 128 || H == 64) && (C == 4 || C == 8); }

int launch_dstep_fwd(const DstepFwdArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_DATASTEP, s);
    if (!dstep_fused_ok(a.C, a.N, a.N)) { lg_set_error("dstep_fwd: C=%d N=%d has no one-launch instance", a.C, a.N); return -2; }
    if (a.N == 128) return a.C == 4 ? launch_fwd_t<128, 4>(a, s) : launch_fwd_t<128, 8>(a, s);
    return a.C == 4 ? launch_fwd_t<64, 4>(a, s) : launch_fwd_t<64, 8>(a, s);
}

size_t dstep_bwd_part_floats(int C, int B, int N) { return (size_t)B * C * (14 + 3 * 10) + (size_t)B * (N * N / 4 / 256) * C * 5; }   // the four stages' rows | the pixelwise kernel's [wg][C][5]

// the launch pair + the five reductions of its partial rows (deferred when a reduce queue is active)
int launch_dstep_bwd(const DstepBwdArgs& a, const DstepBwdGrads& g, hipStream_t s) {
    if (!dstep_fused_ok(a.C, a.N, a.N)) { lg_set_error("dstep_bwd: C=%d N=%d has no one-launch instance", a.C, a.N); return -2; }
    if (a.g == a.dz) { lg_set_error("dstep_bwd: the incoming gradient and dz must be different buffers"); return -2; }
    if (!a.part_pre || !a.part_top || !a.part_dt1 || !a.part_d3 || !a.part_d1) { lg_set_error("dstep_bwd: partial-sum scratch missing"); return -2; }
    int rc;
    if (a.N == 128) rc = a.C == 4 ? launch_bwd_t<128, 4>(a, s) : launch_bwd_t<128, 8>(a, s);
    else rc = a.C == 4 ? launch_bwd_t<64, 4>(a, s) : launch_bwd_t<64, 8>(a, s);
    if (rc) return rc;
    ChanReduce m;
    // top stage (the pixelwise kernel's sums folded in): the slots of k_dstep_top_bwd
    memset(&m, 0, sizeof(m));
    for (int k = 0; k < 9; ++k) { m.dst[k] = g.dt3w + k; m.stride[k] = 9; }
    m.dst[9] = g.eta; m.stride[9] = 0;                          // summed over channels
    m.dst[10] = g.dt3b; m.dst2[10] = g.rtb; m.stride[10] = 1;   // d bias(DT.3) and d RT.bias
    m.dst[11] = g.rtw; m.stride[11] = 1;
    m.dst[12] = g.rw; m.stride[12] = 1;
    m.dst[13] = g.rb; m.stride[13] = 0;                         // summed over channels
    m.NK = 14; m.C = a.C; m.nslices = a.B; m.allc_mask = (1u << 9) | (1u << 13);
    if ((rc = launch_reduce_chan(a.part_top, m, s))) return rc;
    const float* parts[3] = {a.part_dt1, a.part_d3, a.part_d1};
    float* dw[3] = {g.dt1w, g.d3w, g.d1w};
    float* db[3] = {g.dt1b, g.d3b, g.d1b};
    for (int j = 0; j < 3; ++j) {
        memset(&m, 0, sizeof(m));
        for (int k = 0; k < 9; ++k) { m.dst[k] = dw[j] + k; m.stride[k] = 9; }
        m.dst[9] = db[j]; m.stride[9] = 1;
        m.NK = 10; m.C = a.C; m.nslices = a.B; m.allc_mask = 0;
        if ((rc = launch_reduce_chan(parts[j], m, s))) return rc;
    }
    return 0;
}
