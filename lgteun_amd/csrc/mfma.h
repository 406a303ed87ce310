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


// same tile with the weight rows in LDS ([N][ldb], ldb = K + 4 keeps the 16-byte fragment reads conflict-free)
template <int MT, int NT, int K>
__device__ __forceinline__ void wave_gemm_ld(f32x4 (&acc)[MT][NT], const float* A, int lda, const float* Bl, int ldb) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
#pragma unroll 2
    for (int k0 = 0; k0 < K; k0 += 16) {
        float4 av[MT], bv[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const float4*>(A + (mt * 16 + r) * lda + k0 + 4 * g);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = *reinterpret_cast<const float4*>(Bl + (nt * 16 + r) * ldb + k0 + 4 * g);
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

// ================================================================================================
// bf16-operand / fp32-accumulate tiles (throughput mode): v_mfma_f32_16x16x32_bf16, and the 16-deep form
// v_mfma_f32_16x16x16_bf16 for K = 16.  Operand maps (cdna guide section 3): lane l holds A[row l&15][k = 8(l>>4) + j] and
// B[k = 8(l>>4) + j][col l&15], j = 0..7 (4(l>>4) + j, j = 0..3 for the 16-deep form); C/D as in the fp32 tile.
// ================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// weights stay fp32 in HBM; a wave converts its B fragments once ([N][K] row-major, K multiple of 32)
template <int NT, int KB>
__device__ __forceinline__ void load_bfrag_bf16(bf16x8 (&bf)[NT][KB], const float* __restrict__ Wg, int K) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const float4 lo = *reinterpret_cast<const float4*>(Wg + (size_t)(nt * 16 + r) * K + kb * 32 + 8 * g);
            const float4 hi = *reinterpret_cast<const float4*>(Wg + (size_t)(nt * 16 + r) * K + kb * 32 + 8 * g + 4);
            bf[nt][kb] = (bf16x8){(__bf16)lo.x, (__bf16)lo.y, (__bf16)lo.z, (__bf16)lo.w, (__bf16)hi.x, (__bf16)hi.y, (__bf16)hi.z, (__bf16)hi.w};
        }
}
template <int NT>
__device__ __forceinline__ void load_bfrag_bf16_k16(s16x4 (&bf)[NT], const float* __restrict__ Wg) {   // K = 16
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const float4 v = *reinterpret_cast<const float4*>(Wg + (size_t)(nt * 16 + r) * 16 + 4 * g);
        const bf16x4 h = (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        bf[nt] = *reinterpret_cast<const s16x4*>(&h);
    }
}
// acc[mt][nt] += A[mt*16.., :32*KB] * B ; A: bf16 rows in LDS (row stride lda halves, 16-byte aligned rows)
template <int MT, int NT, int KB>
__device__ __forceinline__ void wave_gemm_bf(f32x4 (&acc)[MT][NT], const __bf16* A, int lda, const bf16x8 (&bf)[NT][KB]) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        bf16x8 av[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const bf16x8*>(A + (mt * 16 + r) * lda + kb * 32 + 8 * g);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[mt], bf[nt][kb], acc[mt][nt], 0, 0, 0);
    }
}
template <int MT, int NT>
__device__ __forceinline__ void wave_gemm_bf_k16(f32x4 (&acc)[MT][NT], const __bf16* A, int lda, const s16x4 (&bf)[NT]) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    s16x4 av[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const s16x4*>(A + (mt * 16 + r) * lda + 4 * g);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av[mt], bf[nt], acc[mt][nt], 0, 0, 0);
}
