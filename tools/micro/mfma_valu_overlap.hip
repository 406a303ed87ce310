// Microbenchmark (gfx950): can fp32 / bf16 MFMA and VALU work overlap on one SIMD?
//   mode 0: MFMA only   mode 1: VALU only   mode 2: both interleaved in every wave
//   mode 3: wave-specialised (even waves MFMA, odd waves VALU; 2 waves per SIMD)
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, bool BF>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)a; bb[i] = (__bf16)b; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 4) == 0);   // waves 0-3 -> SIMD 0-3 first wave, 4-7 second
    const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 4) != 0);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (BF) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[u & 3], 0, 0, 0);
                else acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 3], 0, 0, 0);
            }
        }
        if (do_valu) {
#pragma unroll
            for (int u = 0; u < 64; ++u) v[u & 7] = __builtin_fmaf(v[u & 7], 1.00001f, 0.5f);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, bool BF>
static float run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, BF><<<256, 512>>>(d, iters);
    hipEventRecord(e0);
    k<MODE, BF><<<256, 512>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * sizeof(float));
    const int iters = 4000;
    // per iteration per wave: 8 MFMA (fp32 16x16x4: 32 cycles each = 256; bf16 16x16x32: 16? cycles each), 64 VALU (4 cycles each = 256)
    printf("fp32 MFMA: mfma %.1f us  valu %.1f us  interleaved %.1f us  specialised %.1f us\n", run<0, false>(d, iters), run<1, false>(d, iters),
           run<2, false>(d, iters), run<3, false>(d, iters));
    printf("bf16 MFMA: mfma %.1f us  valu %.1f us  interleaved %.1f us  specialised %.1f us\n", run<0, true>(d, iters), run<1, true>(d, iters),
           run<2, true>(d, iters), run<3, true>(d, iters));
    return 0;
}
