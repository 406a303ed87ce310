// fp32 matrix-core GEMM tile shared by the FFN forward/backward kernels (v_mfma_f32_16x16x4_f32: exact fp32).
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// acc[mt][nt] += A[mt*16.., :K] * W[nt*16.., :K]^T ; A in LDS (row stride lda), W in global ([N][K] row-major)
template <int MT, int NT, int K>
__device__ __forceinline__ void wave_gemm(f32x4 (&acc)[MT][NT], const float* A, int lda, const float* __restrict__ Wg) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
#pragma unroll 2
    for (int k0 = 0; k0 < K; k0 += 16) {
        float4 av[MT], bv[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const float4*>(A + (mt * 16 + r) * lda + k0 + 4 * g);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = *reinterpret_cast<const float4*>(Wg + (size_t)(nt * 16 + r) * K + k0 + 4 * g);
        // k sub-step outermost: consecutive MFMAs hit different accumulators (the 16x16x4 f32 MFMA has a 40-cycle dependent
        // latency against a 32-cycle issue interval)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float a_ = ks == 0 ? av[mt].x : (ks == 1 ? av[mt].y : (ks == 2 ? av[mt].z : av[mt].w));
                    const float b_ = ks == 0 ? bv[nt].x : (ks == 1 ? bv[nt].y : (ks == 2 ? bv[nt].z : bv[nt].w));
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b_, acc[mt][nt], 0, 0, 0);
                }
    }
}


// ---- register-resident weights: B fragments of a [N][K] row-major weight, loaded once and reused for every row chunk
template <int NT, int KB>
__device__ __forceinline__ void load_bfrag(float4 (&bf)[NT][KB], const float* __restrict__ Wg, int K) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) bf[nt][kb] = *reinterpret_cast<const float4*>(Wg + (size_t)(nt * 16 + r) * K + kb * 16 + 4 * g);
}
template <int MT, int NT, int KB>
__device__ __forceinline__ void wave_gemm_rb(f32x4 (&acc)[MT][NT], const float* A, int lda, const float4 (&bf)[NT][KB]) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        float4 av[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const float4*>(A + (mt * 16 + r) * lda + kb * 16 + 4 * g);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float a_ = ks == 0 ? av[mt].x : (ks == 1 ? av[mt].y : (ks == 2 ? av[mt].z : av[mt].w));
                    const float b_ = ks == 0 ? bf[nt][kb].x : (ks == 1 ? bf[nt][kb].y : (ks == 2 ? bf[nt][kb].z : bf[nt][kb].w));
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b_, acc[mt][nt], 0, 0, 0);
                }
    }
}
