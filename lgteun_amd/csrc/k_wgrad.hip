// Weight-gradient GEMMs of the 1x1 convs for gfx950:  dW[n][k] = sum_p Y[p][n] * X[p][k],  db[n] = sum_p Y[p][n]
// (autograd of bmu.point_conv, reference models/common/basic_module_unformer_v2.py:13-14).
// The reduction runs over PIXELS (up to B*H*W = 524 288 at bs=32), the output is tiny (16..256 squared), so the
// pixel axis is the MFMA K dimension: v_mfma_f32_16x16x4_f32 with lane (r,g) feeding A[i=r][k=g] = Y[p+g][n0+r] and
// B[k=g][j=r] = X[p+g][k0+r] -- both operands are read in their natural [pixel][channel] layout, 64-byte segments,
// no transposes.  Every wave owns a 64x64 block of dW over a slice of the pixels; the 4 waves of a workgroup are summed in
// LDS and the workgroup's partial goes to a slab that a second kernel sums in a fixed slice order.
#include "kernels.h"
#include "bwd_kernels.h"
#include "hstore.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));


template <bool YBF, bool XBF>
__global__ __launch_bounds__(256) void k_wgrad(WgradArgs a, int k_blocks, long px_per_wave, float* slab, float* bslab) {
    // Operand fetch: lane (r,g) loads ONE float4 of Y and ONE of X per 4-pixel step: Y[p+g][n0 + 4r .. 4r+3], X[p+g][k0 + 4r ..].
    // Element t of the float4 feeds MFMA tile t, so tile t owns the strided channel set {4r + t}: a 64-channel row is one
    // fully coalesced 256-byte segment per 16 lanes, and 2 loads feed 16 MFMAs.  Output row of acc[i][j][v] (MFMA row 4g+v,
    // col r) is therefore dW[n0 + 4(4g+v) + i][k0 + 4r + j].
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int nb = blockIdx.y / k_blocks, kb = blockIdx.y - nb * k_blocks;
    const int n0 = nb * 64, k0 = kb * 64;
    const int NW = min(64, a.N - n0), KW = min(64, a.K - k0);   // valid widths of this block (multiples of 16)
    const bool yok = 4 * r < NW, xok = 4 * r < KW;
    const long slice = (long)blockIdx.x * 4 + wave;
    const long p_begin = slice * px_per_wave;
    long p_end = p_begin + px_per_wave;
    if (p_end > a.P) p_end = a.P;
    f32x4 acc[4][4];
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (long p = p_begin; p < p_end; p += 4) {
        const long row = p + g;
        const bool valid = row < p_end;
        float4 af = make_float4(0.f, 0.f, 0.f, 0.f), bf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid && yok) af = HS<YBF>::ld4(a.Y, row * a.ldy + n0 + 4 * r);
        if (valid && xok) {
            bf = HS<XBF>::ld4(a.X, row * a.ldx + k0 + 4 * r);
        }
        bsum.x += af.x; bsum.y += af.y; bsum.z += af.z; bsum.w += af.w;
        const float av[4] = {af.x, af.y, af.z, af.w}, bv[4] = {bf.x, bf.y, bf.z, bf.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    // the 4 waves of the workgroup hold partials of the same 64x64 block: sum them in LDS, one slab slice per workgroup
    __shared__ float red[64 * 64 + 64];
    for (int i = threadIdx.x; i < 64 * 64 + 64; i += 256) red[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {   // waves take turns: fixed order, no float atomics -> bitwise reproducible
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int v = 0; v < 4; ++v) red[(4 * (4 * g + v) + i) * 64 + 4 * r + j] += acc[i][j][v];
            // bias: lanes with equal r hold the same 4 channels for different pixels
            float4 b = bsum;
            b.x += __shfl_xor(b.x, 16); b.y += __shfl_xor(b.y, 16); b.z += __shfl_xor(b.z, 16); b.w += __shfl_xor(b.w, 16);
            b.x += __shfl_xor(b.x, 32); b.y += __shfl_xor(b.y, 32); b.z += __shfl_xor(b.z, 32); b.w += __shfl_xor(b.w, 32);
            if (g == 0) {
                red[64 * 64 + 4 * r] += b.x; red[64 * 64 + 4 * r + 1] += b.y; red[64 * 64 + 4 * r + 2] += b.z; red[64 * 64 + 4 * r + 3] += b.w;
            }
        }
        __syncthreads();
    }
    float* my = slab + (long)blockIdx.x * ((long)a.N * a.K);
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int rr = i >> 6, cc = i & 63;
        if (rr < NW && cc < KW) my[(long)(n0 + rr) * a.K + k0 + cc] = red[i];
    }
    if (a.db && kb == 0 && threadIdx.x < NW) bslab[(long)blockIdx.x * a.N + n0 + threadIdx.x] = red[64 * 64 + threadIdx.x];
}

// narrow shapes (N or K < 64): scalar operand loads, only the needed 16x16 tiles are issued
template <bool YBF, bool XBF>
__global__ __launch_bounds__(256) void k_wgrad_narrow(WgradArgs a, int k_blocks, long px_per_wave, float* slab, float* bslab) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int nb = blockIdx.y / k_blocks, kb = blockIdx.y - nb * k_blocks;
    const int n0 = nb * 64, k0 = kb * 64;
    const int NT = min(4, (a.N - n0) / 16), KT = min(4, (a.K - k0) / 16);
    const long slice = (long)blockIdx.x * 4 + wave;
    const long p_begin = slice * px_per_wave;
    long p_end = p_begin + px_per_wave;
    if (p_end > a.P) p_end = a.P;
    f32x4 acc[4][4];
    float bsum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 4
    for (long p = p_begin; p < p_end; p += 4) {
        const long row = p + g;
        const bool valid = row < p_end;
        float af[4], bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            af[t] = (valid && t < NT) ? HS<YBF>::ld1(a.Y, row * a.ldy + n0 + t * 16 + r) : 0.f;
            bf[t] = (valid && t < KT) ? HS<XBF>::ld1(a.X, row * a.ldx + k0 + t * 16 + r) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bsum[i] += af[i];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i < NT && j < KT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    // the 4 waves of the workgroup hold partials of the same 64x64 block: sum them in LDS, one slab slice per workgroup
    __shared__ float red[64 * 64 + 64];
    for (int i = threadIdx.x; i < 64 * 64 + 64; i += 256) red[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {   // waves take turns: fixed order, no float atomics -> bitwise reproducible
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i < NT && j < KT) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) red[(i * 16 + 4 * g + v) * 64 + j * 16 + r] += acc[i][j][v];
                    }
            if (a.db && kb == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float s = bsum[i];
                    s += __shfl_xor(s, 16);
                    s += __shfl_xor(s, 32);
                    if (g == 0 && i < NT) red[64 * 64 + i * 16 + r] += s;
                }
            }
        }
        __syncthreads();
    }
    float* my = slab + (long)blockIdx.x * ((long)a.N * a.K);
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int rr = i >> 6, cc = i & 63;
        if (rr < NT * 16 && cc < KT * 16) my[(long)(n0 + rr) * a.K + k0 + cc] = red[i];
    }
    if (a.db && kb == 0 && threadIdx.x < NT * 16) bslab[(long)blockIdx.x * a.N + n0 + threadIdx.x] = red[64 * 64 + threadIdx.x];
}

// dst[row*ld + col] += sum_s slab[s][row*cols + col]   for row < rows_valid, col < cols_valid
// block = 64 consecutive outputs x 4 slice phases (fixed summation order: bitwise reproducible)
// block = 64 consecutive outputs x 4 slice phases (fixed summation order: bitwise reproducible)
__global__ __launch_bounds__(256) void k_reduce_slab(const float* __restrict__ slab, long nslices, int rows, int cols, float* dst,
                                                     int ld, int rows_valid, int cols_valid, const float* __restrict__ slab2, int n2,
                                                     float* dst2, int n2_valid) {
    // 16 outputs x 16 slice phases per workgroup: every thread has 8 independent loads in flight (the old 64 x 4 shape walked 128
    // slices per thread and was pure load latency, ~12 us per call); fixed summation order -> deterministic.
    __shared__ float part[16][17];
    // job 0 (blockIdx.y == 0): the [rows][cols] slab; job 1: an optional [n2] vector slab (bias) in the same launch
    if (blockIdx.y == 1) { slab = slab2; rows = 1; cols = n2; dst = dst2; ld = n2; rows_valid = 1; cols_valid = n2_valid; }
    const long n = (long)rows * cols;
    const int o = threadIdx.x & 15, ph = threadIdx.x >> 4;
    const long i = blockIdx.x * 16L + o;
    float sv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) sv[u] = 0.f;
    if (i < n) {
        long k = ph;
        for (; k + 16 * 7 < nslices; k += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) sv[u] += slab[(k + 16 * u) * n + i];
        }
        for (; k < nslices; k += 16) sv[0] += slab[k * n + i];
    }
    part[ph][o] = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
    __syncthreads();
    if (ph == 0 && i < n) {
        const int row = (int)(i / cols), col = (int)(i - (long)row * cols);
        if (row < rows_valid && col < cols_valid) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += part[q][o];
            dst[(long)row * ld + col] += t;
        }
    }
}

static int launch_reduce_slab2(const float* slab, long nslices, int rows, int cols, float* dst, int ld, int rows_valid, int cols_valid,
                               const float* slab2, int n2, float* dst2, int n2_valid, hipStream_t s) {
    long n = (long)rows * cols;
    dim3 grid((unsigned)((n + 15) / 16), slab2 ? 2 : 1);
    k_reduce_slab<<<grid, 256, 0, s>>>(slab, nslices, rows, cols, dst, ld, rows_valid, cols_valid, slab2, n2, dst2, n2_valid);
    LG_CHECK_LAUNCH();
    return 0;
}
int launch_reduce_slab(const float* slab, long nslices, int rows, int cols, float* dst, int ld, int rows_valid, int cols_valid,
                       hipStream_t s) {
    return launch_reduce_slab2(slab, nslices, rows, cols, dst, ld, rows_valid, cols_valid, nullptr, 0, nullptr, 0, s);
}

size_t wgrad_slab_floats(int N, int K, long P) {
    // sized for the launch geometry below (upper bound)
    const int blocks = ((N + 63) / 64) * ((K + 63) / 64);
    long splits = 512 / blocks;
    if (splits < 1) splits = 1;
    return (size_t)splits * ((size_t)N * K + N);
}

int launch_wgrad(const WgradArgs& a, float* slab, hipStream_t s) {
    ProfScope prof__(LG_K_WGRAD, s);
    if ((a.N & 15) || (a.K & 15) || a.N <= 0 || a.K <= 0 || a.P <= 0) { lg_set_error("wgrad: N,K must be positive multiples of 16"); return -2; }
    const int n_blocks = (a.N + 63) / 64, k_blocks = (a.K + 63) / 64;
    const int blocks = n_blocks * k_blocks;
    long splits = 512 / blocks;   // two waves per SIMD: the load -> MFMA chain needs another wave to hide HBM latency
    if (splits < 1) splits = 1;
    long nslices = splits * 4;
    long px = (a.P + nslices - 1) / nslices;
    px = (px + 3) & ~3L;
    if (px < 4) px = 4;
    // shrink the slice count if the tensor is small
    nslices = (a.P + px - 1) / px;
    splits = (nslices + 3) / 4;
    nslices = splits * 4;
    float* bslab = slab + splits * (long)a.N * a.K;
    dim3 grid((unsigned)splits, (unsigned)blocks);
    const bool wide = (a.N % 64 == 0) && (a.K % 64 == 0);
#define LG_WG(KERN)                                                                                            \
    do {                                                                                                       \
        if (a.ybf && a.xbf) KERN<true, true><<<grid, 256, 0, s>>>(a, k_blocks, px, slab, bslab);               \
        else if (a.ybf) KERN<true, false><<<grid, 256, 0, s>>>(a, k_blocks, px, slab, bslab);                  \
        else if (a.xbf) KERN<false, true><<<grid, 256, 0, s>>>(a, k_blocks, px, slab, bslab);                  \
        else KERN<false, false><<<grid, 256, 0, s>>>(a, k_blocks, px, slab, bslab);                            \
    } while (0)
    if (wide) LG_WG(k_wgrad); else LG_WG(k_wgrad_narrow);
#undef LG_WG
    LG_CHECK_LAUNCH();
    return launch_reduce_slab2(slab, splits, a.N, a.K, a.dW, a.ldw, a.n_valid, a.k_valid, a.db ? bslab : nullptr, a.N, a.db, a.n_valid, s);
}
