// Storage of the 4e-wide hidden / saved FFN tensors (gelu(h1), gelu'(h1), h2, gelu(h3), gelu'(h3), dh2, dh1):
// fp32 in parity mode (lg_config.precision = 0), bf16 in throughput mode (precision = 1).  Arithmetic on them is fp32
// either way: values are widened on load and rounded (round-to-nearest-even, v_cvt_pk_bf16_f32) on store.
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 bf16_t;

template <bool BF>
struct HS;

// raw4 / ldraw / widen: a prefetch must keep the loaded bits untouched in its registers -- widening bf16 at load time makes the
// conversion (and so an s_waitcnt for the load) sit right behind the issue, which serialises the prefetch it was meant to hide
template <>
struct HS<false> {
    typedef float4 raw4;
    static __device__ __forceinline__ raw4 ldraw(const void* base, long idx) { return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + idx); }
    static __device__ __forceinline__ raw4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ float4 widen(raw4 u) { return u; }
    static __device__ __forceinline__ float4 ld4(const void* base, long idx) { return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + idx); }
    static __device__ __forceinline__ void st4(void* base, long idx, float4 v) { *reinterpret_cast<float4*>(static_cast<float*>(base) + idx) = v; }
    // streaming store: a saved activation is not read again before the backward (hundreds of megabytes later)
    static __device__ __forceinline__ void st4_nt(void* base, long idx, float4 v) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store((f4){v.x, v.y, v.z, v.w}, reinterpret_cast<f4*>(static_cast<float*>(base) + idx));
    }
    static __device__ __forceinline__ float ld1(const void* base, long idx) { return static_cast<const float*>(base)[idx]; }
    static __device__ __forceinline__ void st1(void* base, long idx, float v) { static_cast<float*>(base)[idx] = v; }
};

template <>
struct HS<true> {
    typedef uint2 raw4;
    static __device__ __forceinline__ raw4 ldraw(const void* base, long idx) { return *reinterpret_cast<const uint2*>(static_cast<const bf16_t*>(base) + idx); }
    static __device__ __forceinline__ raw4 zero() { return make_uint2(0u, 0u); }
    static __device__ __forceinline__ float4 widen(raw4 u) {
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xffff0000u));
    }
    static __device__ __forceinline__ float4 ld4(const void* base, long idx) {
        const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const bf16_t*>(base) + idx);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xffff0000u));
    }
    static __device__ __forceinline__ void st4(void* base, long idx, float4 v) {
        bf16_t h[4] = {static_cast<bf16_t>(v.x), static_cast<bf16_t>(v.y), static_cast<bf16_t>(v.z), static_cast<bf16_t>(v.w)};
        *reinterpret_cast<uint2*>(static_cast<bf16_t*>(base) + idx) = *reinterpret_cast<const uint2*>(h);
    }
    static __device__ __forceinline__ void st4_nt(void* base, long idx, float4 v) { st4(base, idx, v); }
    static __device__ __forceinline__ float ld1(const void* base, long idx) { return static_cast<float>(static_cast<const bf16_t*>(base)[idx]); }
    static __device__ __forceinline__ void st1(void* base, long idx, float v) { static_cast<bf16_t*>(base)[idx] = static_cast<bf16_t>(v); }
};
