// Microbenchmark (gfx950): issue cost of the instruction kinds k_attn_m's softmax / P V phase is made of, 1, 2 and 3 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rates2.hip -o build/valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float v[16];
    unsigned u[16];
    for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x * 1e-3f + i; u[i] = threadIdx.x * 2654435761u + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(1.00001f), "v"(0.5f));
                else if (MODE == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(v[(i + 1) & 15]));
                else if (MODE == 2) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(v[i]) : "v"(u[i]));
                else if (MODE == 3) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 1) & 15]));
                else if (MODE == 4) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[i]) : "v"(1));
                else if (MODE == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]));
                else if (MODE == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
                else if (MODE == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                else if (MODE == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(0x7feb352du));
                else if (MODE == 9) asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(u[i]));
                else if (MODE == 10) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(v[(i + 1) & 15]));
                else if (MODE == 11) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 1) & 15]));
                else if (MODE == 12) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(0.25f));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i] + (float)u[i];
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
    const int iters = 2000;
    const char* names[13] = {"v_fma_f32", "v_cvt_pk_f16_f32", "v_fma_mix_f32", "v_permlane32_swap", "v_ldexp_f32", "v_max3_f32", "v_cndmask_b32", "v_exp_f32", "v_mul_lo_u32", "v_xor_b32_sdwa", "v_cvt_pk_bf16_f32", "v_permlane16_swap", "v_sub_f32"};
    for (int wg = 1; wg <= 3; ++wg) {
        float t[13] = {run<0>(d, iters, wg), run<1>(d, iters, wg), run<2>(d, iters, wg), run<3>(d, iters, wg), run<4>(d, iters, wg), run<5>(d, iters, wg), run<6>(d, iters, wg),
                       run<7>(d, iters, wg), run<8>(d, iters, wg), run<9>(d, iters, wg), run<10>(d, iters, wg), run<11>(d, iters, wg), run<12>(d, iters, wg)};
        printf("%d wave(s)/SIMD: cycles per wave-instruction per SIMD at 2.1 GHz\n", wg);
        for (int m = 0; m < 13; ++m) printf("  %-20s %8.1f us   %.2f\n", names[m], t[m], t[m] * 1e-6 * 2.1e9 / (iters * 64.0) / wg);
    }
    return 0;
}
