// Microbenchmark (gfx950): fp32-equivalent GEMM out of bf16 matrix-core instructions.
//   a = a1 + a2 + a3 with ai = bf16 pieces (8 + 8 + 8 significand bits: exact), same for b; the product keeps the six terms
//   a1b1, a1b2, a2b1, a1b3, a2b2, a3b1 (dropped: <= 2^-24 relative), accumulated in fp32 by v_mfma_f32_16x16x32_bf16.
// Questions: (1) accuracy against fp64 next to v_mfma_f32_16x16x4_f32 (exact fp32 fma chain) and a 3-term (2-piece) split;
//            (2) matrix-pipe time of each; (3) does VALU work overlap with the bf16 MFMAs / with the fp32 MFMA?
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/split_bf16_gemm.hip -o /tmp/split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---------------- accuracy: one wave computes C[16][16] = A[16][K] * B[K][16]^T (both row-major [16][K]) ----------------
__device__ inline void split3(float v, __bf16& p1, __bf16& p2, __bf16& p3) {
    p1 = (__bf16)v;
    const float r1 = v - (float)p1;
    p2 = (__bf16)r1;
    const float r2 = r1 - (float)p2;
    p3 = (__bf16)r2;
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// two fp16 pieces (11 + 11 significand bits), round-toward-zero conversions (v_cvt_pkrtz_f16_f32 saturates instead of overflowing)
__device__ inline void split2h(float v, _Float16& p1, _Float16& p2) {
    const auto h = __builtin_amdgcn_cvt_pkrtz(v, 0.f);
    p1 = (_Float16)h[0];
    const float r = v - (float)p1;
    const auto l = __builtin_amdgcn_cvt_pkrtz(r, 0.f);
    p2 = (_Float16)l[0];
}
// fp16 two-piece split, three products (a1 b2 + a2 b1 + a1 b1); SB = power-of-two pre-scale of the B ("weight") operand
template <int SB>
__global__ void k_acc_h(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
        f16x8 a1, a2, b1, b2;
        for (int j = 0; j < 8; ++j) {
            _Float16 x, y;
            split2h(A[r * K + k0 + 8 * g + j], x, y); a1[j] = x; a2[j] = y;
            split2h(B[r * K + k0 + 8 * g + j] * (float)SB, x, y); b1[j] = x; b2[j] = y;
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc, 0, 0, 0);
    }
    for (int v = 0; v < 4; ++v) C[(4 * g + v) * 16 + r] = acc[v] * (1.0f / SB);
}

template <int MODE>   // 0: fp32 MFMA, 1: 6-term split, 2: 3-term split (two pieces), 3: plain bf16
__global__ void k_acc(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k0 + g], B[r * K + k0 + g], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 32) {
            bf16x8 a1, a2, a3, b1, b2, b3;
            for (int j = 0; j < 8; ++j) {
                __bf16 x, y, z;
                split3(A[r * K + k0 + 8 * g + j], x, y, z); a1[j] = x; a2[j] = y; a3[j] = z;
                split3(B[r * K + k0 + 8 * g + j], x, y, z); b1[j] = x; b2[j] = y; b3[j] = z;
            }
            if (MODE == 1) {   // small terms first
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, acc, 0, 0, 0);
            }
            if (MODE == 1 || MODE == 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);
        }
    }
    for (int v = 0; v < 4; ++v) C[(4 * g + v) * 16 + r] = acc[v];   // C[row = 4g+v of A][col = r of B]
}

// ---------------- speed: MFMA stream alone / with VALU filler in the same wave ----------------
template <int MODE, int NVALU>   // MODE 0: fp32 16x16x4 (8 per K=32 block), 1: 6 bf16 MFMAs per K=32 block
__global__ __launch_bounds__(256) void k_speed(float* out, int iters) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a = threadIdx.x * 1e-3f, b = 1.0001f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(a + i); bb[i] = (__bf16)b; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    for (int it = 0; it < iters; ++it) {
        // one 16x16 output tile x K = 32, four independent tiles
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MODE == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < 6; ++u) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < NVALU; ++u) v[u & 7] = __builtin_fmaf(v[u & 7], 1.00001f, 0.5f);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int NVALU>
static float run_speed(float* d, int iters, int wg_per_cu) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_speed<MODE, NVALU><<<256 * wg_per_cu, 256>>>(d, iters);
    hipEventRecord(e0);
    k_speed<MODE, NVALU><<<256 * wg_per_cu, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    // ---- accuracy
    for (int K : {64, 128, 256}) {
        for (int dist = 0; dist < 3; ++dist) {
            std::vector<float> A(16 * K), B(16 * K);
            srand(K + dist);
            for (auto& x : A) x = dist == 1 ? (float)(rand() / (double)RAND_MAX) : (float)(2.0 * rand() / RAND_MAX - 1.0);           // gelu-like (>=0) / signed
            if (dist == 2) for (auto& x : A) x *= (float)exp(-8.0 * rand() / RAND_MAX);   // wide dynamic range: many tiny activations
            for (auto& x : B) x = (float)((2.0 * rand() / RAND_MAX - 1.0) / sqrt((double)K));
            std::vector<double> ref(256, 0.0), mag(256, 0.0);
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j)
                    for (int k = 0; k < K; ++k) { ref[i * 16 + j] += (double)A[i * K + k] * B[j * K + k]; mag[i * 16 + j] += fabs((double)A[i * K + k] * B[j * K + k]); }
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 256 * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            double err[7];
            for (int mode = 0; mode < 7; ++mode) {
                if (mode == 0) k_acc<0><<<1, 64>>>(dA, dB, dC, K);
                if (mode == 1) k_acc<1><<<1, 64>>>(dA, dB, dC, K);
                if (mode == 2) k_acc<2><<<1, 64>>>(dA, dB, dC, K);
                if (mode == 3) k_acc<3><<<1, 64>>>(dA, dB, dC, K);
                if (mode == 4) k_acc_h<1><<<1, 64>>>(dA, dB, dC, K);
                if (mode == 5) k_acc_h<16><<<1, 64>>>(dA, dB, dC, K);
                if (mode == 6) k_acc_h<256><<<1, 64>>>(dA, dB, dC, K);
                std::vector<float> C(256);
                hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
                double num = 0, den = 0;
                for (int i = 0; i < 256; ++i) { num += (C[i] - ref[i]) * (C[i] - ref[i]); den += ref[i] * ref[i]; }
                err[mode] = sqrt(num / den);
            }
            printf("K=%3d %s  rel-L2 vs fp64: fp32-mfma %.3e | bf16x6 %.3e | bf16x3 %.3e | bf16 %.3e | fp16x3 %.3e  (B x16) %.3e  (B x256) %.3e\n", K,
                   dist == 1 ? "A>=0  " : (dist == 2 ? "A small" : "signed"), err[0], err[1], err[2], err[3], err[4], err[5], err[6]);
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    }
    // ---- speed
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    const int iters = 2000;
    for (int wg = 1; wg <= 2; ++wg) {
        printf("%d wave(s)/SIMD, per wave per iteration 4 x (K=32 tile + N VALU fma):\n", wg);
        printf("  fp32 16x16x4 x8 : N=0 %.1f us  N=16 %.1f  N=32 %.1f  N=64 %.1f\n", run_speed<0, 0>(d, iters, wg), run_speed<0, 16>(d, iters, wg),
               run_speed<0, 32>(d, iters, wg), run_speed<0, 64>(d, iters, wg));
        printf("  bf16 16x16x32 x6: N=0 %.1f us  N=16 %.1f  N=32 %.1f  N=64 %.1f\n", run_speed<1, 0>(d, iters, wg), run_speed<1, 16>(d, iters, wg),
               run_speed<1, 32>(d, iters, wg), run_speed<1, 64>(d, iters, wg));
    }
    return 0;
}
