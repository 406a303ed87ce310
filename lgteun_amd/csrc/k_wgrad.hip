// Weight-gradient GEMMs of the 1x1 convs for gfx950:  dW[n][k] = sum_p Y[p][n] * X[p][k],  db[n] = sum_p Y[p][n]
// (autograd of bmu.point_conv, reference models/common/basic_module_unformer_v2.py:13-14).
// The reduction runs over PIXELS (up to B*H*W = 524 288 at bs=32), the output is tiny (16..256 squared), so the
// pixel axis is the MFMA K dimension: v_mfma_f32_16x16x4_f32 with lane (r,g) feeding A[i=r][k=g] = Y[p+g][n0+r] and
// B[k=g][j=r] = X[p+g][k0+r] -- both operands are read in their natural [pixel][channel] layout, 64-byte segments,
// no transposes.  Every wave owns a 64x64 block of dW over a slice of the pixels and writes its partial to a slab;
// a second kernel sums the slab in a fixed order (bitwise reproducible, no float atomics).
#include "kernels.h"
#include "bwd_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int XF>
__device__ __forceinline__ float xform(float v) { return XF == 1 ? gelu_f(v) : v; }

template <int XF>
__global__ __launch_bounds__(256) void k_wgrad(WgradArgs a, int k_blocks, long px_per_wave, float* slab, float* bslab) {
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
    for (long p = p_begin; p < p_end; p += 4) {
        const long row = p + g;
        const bool valid = row < p_end;
        float af[4], bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            af[t] = (valid && t < NT) ? a.Y[row * a.ldy + n0 + t * 16 + r] : 0.f;
            bf[t] = (valid && t < KT) ? xform<XF>(a.X[row * a.ldx + k0 + t * 16 + r]) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bsum[i] += af[i];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i < NT && j < KT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    // partial -> slab[slice][N][K]
    const long nslices = (long)gridDim.x * 4;
    (void)nslices;
    float* my = slab + slice * ((long)a.N * a.K);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i < NT && j < KT) {
#pragma unroll
                for (int v = 0; v < 4; ++v) my[(long)(n0 + i * 16 + 4 * g + v) * a.K + k0 + j * 16 + r] = acc[i][j][v];
            }
    if (a.db && kb == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = bsum[i];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (g == 0 && i < NT) bslab[slice * a.N + n0 + i * 16 + r] = s;
        }
    }
}

// dst[row*ld + col] += sum_s slab[s][row*cols + col]   for row < rows_valid, col < cols_valid
__global__ __launch_bounds__(256) void k_reduce_slab(const float* __restrict__ slab, long nslices, int rows, int cols, float* dst,
                                                     int ld, int rows_valid, int cols_valid) {
    const long n = (long)rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const int row = (int)(i / cols), col = (int)(i - (long)row * cols);
        if (row >= rows_valid || col >= cols_valid) continue;
        float s = 0.f;
        for (long k = 0; k < nslices; ++k) s += slab[k * n + i];
        dst[(long)row * ld + col] += s;
    }
}

int launch_reduce_slab(const float* slab, long nslices, int rows, int cols, float* dst, int ld, int rows_valid, int cols_valid,
                       hipStream_t s) {
    long n = (long)rows * cols;
    int grid = (int)((n + 255) / 256);
    if (grid > 1024) grid = 1024;
    k_reduce_slab<<<grid, 256, 0, s>>>(slab, nslices, rows, cols, dst, ld, rows_valid, cols_valid);
    LG_CHECK_LAUNCH();
    return 0;
}

size_t wgrad_slab_floats(int N, int K, long P) {
    // sized for the launch geometry below (upper bound)
    const int blocks = ((N + 63) / 64) * ((K + 63) / 64);
    long splits = 512 / blocks;
    if (splits < 1) splits = 1;
    return (size_t)(splits * 4) * ((size_t)N * K + N);
}

int launch_wgrad(const WgradArgs& a, float* slab, hipStream_t s) {
    if ((a.N & 15) || (a.K & 15) || a.N <= 0 || a.K <= 0 || a.P <= 0) { lg_set_error("wgrad: N,K must be positive multiples of 16"); return -2; }
    const int n_blocks = (a.N + 63) / 64, k_blocks = (a.K + 63) / 64;
    const int blocks = n_blocks * k_blocks;
    long splits = 512 / blocks;
    if (splits < 1) splits = 1;
    long nslices = splits * 4;
    long px = (a.P + nslices - 1) / nslices;
    px = (px + 3) & ~3L;
    if (px < 4) px = 4;
    // shrink the slice count if the tensor is small
    nslices = (a.P + px - 1) / px;
    splits = (nslices + 3) / 4;
    nslices = splits * 4;
    float* bslab = slab + nslices * (long)a.N * a.K;
    dim3 grid((unsigned)splits, (unsigned)blocks);
    if (a.xf == 1) k_wgrad<1><<<grid, 256, 0, s>>>(a, k_blocks, px, slab, bslab);
    else k_wgrad<0><<<grid, 256, 0, s>>>(a, k_blocks, px, slab, bslab);
    LG_CHECK_LAUNCH();
    int rc = launch_reduce_slab(slab, nslices, a.N, a.K, a.dW, a.ldw, a.n_valid, a.k_valid, s);
    if (rc) return rc;
    if (a.db) rc = launch_reduce_slab(bslab, nslices, 1, a.N, a.db, a.N, 1, a.n_valid, s);
    return rc;
}
