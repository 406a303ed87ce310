// Microbenchmark (gfx950): issue cost of the VALU instruction kinds the FFN kernels are made of, 1 and 2 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rates.hip -o tools/micro/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float v[16];
    v2f p[8];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 8; ++i) p[i] = (v2f){v[2 * i], v[2 * i + 1]};
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 0) {   // 16 v_fma_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(1.00001f), "v"(0.5f));
            } else if (MODE == 1) {   // 8 v_pk_fma_f32 (same flops)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"((v2f){1.00001f, 1.00001f}), "v"((v2f){0.5f, 0.5f}));
            } else if (MODE == 2) {   // 16 v_exp_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            } else if (MODE == 3) {   // 16 v_rcp_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
            } else if (MODE == 4) {   // 16 v_perm_b32
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(0x07060302u));
                    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 3) & 7]), "v"(0x07060302u));
                }
            } else if (MODE == 5) {   // 16 v_and_b32 / v_sub_f32 pairs (8 + 8)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float t;
                    asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(t) : "v"(v[i]));
                    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(t));
                }
            } else if (MODE == 6) {   // 16 v_pk_mul_f32 -> 32 flops each... 8 instr
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"((v2f){1.00001f, 1.00001f}));
            } else if (MODE == 7) {   // 16 v_cndmask / v_cmp style: v_max_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(0.25f));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static float run(float* d, int iters, int wg) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<256 * wg, 256>>>(d, iters);
    (void)hipEventRecord(e0);
    k<MODE><<<256 * wg, 256>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}
int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    const int iters = 4000;
    const char* names[8] = {"16 v_fma_f32", "8 v_pk_fma_f32", "16 v_exp_f32", "16 v_rcp_f32", "16 v_perm_b32", "8 v_and + 8 v_sub", "8 v_pk_mul_f32", "16 v_max_f32"};
    for (int wg = 1; wg <= 2; ++wg) {
        float t[8] = {run<0>(d, iters, wg), run<1>(d, iters, wg), run<2>(d, iters, wg), run<3>(d, iters, wg), run<4>(d, iters, wg), run<5>(d, iters, wg), run<6>(d, iters, wg), run<7>(d, iters, wg)};
        const int n[8] = {16, 8, 16, 16, 16, 16, 8, 16};
        printf("%d wave(s)/SIMD (time per 4 x group per iteration; cycles per instruction per wave at 2.1 GHz):\n", wg);
        for (int m = 0; m < 8; ++m) printf("  %-20s %8.1f us   %.2f cyc/instr\n", names[m], t[m], t[m] * 1e-6 * 2.1e9 / (iters * 4.0 * n[m]) / wg);
    }
    return 0;
}
