// fp32-equivalent GEMM tiles out of bf16 matrix-core instructions (gfx950).
//
// Why: on gfx950 the f32-input MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 VECTOR rate and its time ADDS to the VALU time of the
// SIMD -- same wave or another (tools/micro/mfma_valu_overlap.hip, tools/micro/split_bf16_gemm.hip: 961 us of fp32 MFMA + 807 us
// of v_fma = 1768 us together).  The bf16 matrix pipe is separate (16x the rate) and does overlap with VALU work.
//
// How: every fp32 number is EXACTLY the sum of three bf16 numbers, a = a1 + a2 + a3 (8 + 8 + 8 significand bits: a1 = the high
// half of a's bit pattern, a2 = the high half of (a - a1), a3 = a - a1 - a2, all exact in fp32).  A product a*b is the sum of nine
// piece products; the six with piece indices i + j <= 4 are kept (the dropped three are <= 2^-24 relative, i.e. below fp32's own
// rounding), each one is exact in the matrix core's fp32 accumulator input, and v_mfma_f32_16x16x32_bf16 sums them in fp32:
//     a*b ~ a1 b3 + a3 b1 + a2 b2 + a1 b2 + a2 b1 + a1 b1            (small terms first)
// Measured against fp64 on random K = 64 / 128 / 256 dot products: 1.0e-7 / 1.4e-7 / 2.2e-7 relative L2, against 1.7e-7 / 2.2e-7 /
// 3.1e-7 for the fp32 MFMA (an fp32 fma chain) -- i.e. at least fp32 accuracy -- in 6 x 16 = 96 matrix-pipe cycles per 16x16x32
// block instead of 8 x 32 = 256, and off the VALU.  A 2-piece split (3 products) gives 3.5e-6 and plain bf16 2e-3.
//
// Operand maps (cdna guide section 3): for D = A * B with A [16 x K] and B [K x 16], lane l = (r = l & 15, g = l >> 4) supplies
// A[row r][k = 8 g + j] and B[k = 8 g + j][col r], j = 0..7 (16x16x32) or k = 4 g + j, j = 0..3 (16x16x16), and receives
// D[row 4 g + v][col r], v = 0..3.  The FFN kernels put the WEIGHTS on the A side (rows = output channels) and the pixels on the B
// side, so that a lane ends up with four CONSECUTIVE channels of one pixel: its results leave as one 8-byte (bf16 pieces) or 16-byte
// (fp32) access per pixel instead of four scattered 4-byte ones.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// the three pieces of v as fp32 bit patterns whose HIGH halves are the bf16 pieces (the low halves of p2 / p3 are don't-care)
struct Split3 {
    uint32_t p1, p2, p3;
};
__device__ __forceinline__ Split3 split3(float v) {
    Split3 s;
    s.p1 = __float_as_uint(v);
    const float r1 = v - __uint_as_float(s.p1 & 0xffff0000u);
    s.p2 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(s.p2 & 0xffff0000u);
    s.p3 = __float_as_uint(r2);   // <= 8 significant bits: its high half is exact
    return s;
}
// one dword = the bf16 (high halves) of two fp32 patterns: `lo` in bits 0..15, `hi` in bits 16..31   (one v_perm_b32)
__device__ __forceinline__ uint32_t pack_hi16(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// two values at once: the residual subtractions as packed fp32 (v_pk_add_f32 with neg: the same IEEE subtraction, so the pieces are bit
// for bit those of split3) -- 6 instead of 8 instructions per pair before the packing permutes
typedef float sb_v2f __attribute__((ext_vector_type(2)));
struct Split3x2 {
    u32x2_t p1, p2, p3;
};
__device__ __forceinline__ Split3x2 split3_pair(sb_v2f v) {
    Split3x2 s;
    s.p1 = __builtin_bit_cast(u32x2_t, v);
    const sb_v2f r1 = v - __builtin_bit_cast(sb_v2f, s.p1 & 0xffff0000u);
    s.p2 = __builtin_bit_cast(u32x2_t, r1);
    const sb_v2f r2 = r1 - __builtin_bit_cast(sb_v2f, s.p2 & 0xffff0000u);
    s.p3 = __builtin_bit_cast(u32x2_t, r2);
    return s;
}
// four consecutive-k values -> the 8-byte fragment of each piece
__device__ __forceinline__ void split3_x4(const float (&v)[4], u32x2_t& q1, u32x2_t& q2, u32x2_t& q3) {
    const Split3x2 ab = split3_pair((sb_v2f){v[0], v[1]}), cd = split3_pair((sb_v2f){v[2], v[3]});
    q1 = (u32x2_t){pack_hi16(ab.p1.x, ab.p1.y), pack_hi16(cd.p1.x, cd.p1.y)};
    q2 = (u32x2_t){pack_hi16(ab.p2.x, ab.p2.y), pack_hi16(cd.p2.x, cd.p2.y)};
    q3 = (u32x2_t){pack_hi16(ab.p3.x, ab.p3.y), pack_hi16(cd.p3.x, cd.p3.y)};
}

// weight fragments, split once per workgroup.  W: fp32 [rows][K] row-major, this wave's 16 rows start at W.
struct WFrag32 {   // one 16x32 block of the A operand, three pieces
    bf16x8_t p[3];
};
struct WFrag16 {   // one 16x16 block (K = 16)
    s16x4_t p[3];
};
__device__ __forceinline__ WFrag32 load_wfrag32(const float* __restrict__ W, int K, int kb) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const float4 lo = *reinterpret_cast<const float4*>(W + (size_t)r * K + kb * 32 + 8 * g);
    const float4 hi = *reinterpret_cast<const float4*>(W + (size_t)r * K + kb * 32 + 8 * g + 4);
    const float a[4] = {lo.x, lo.y, lo.z, lo.w}, b[4] = {hi.x, hi.y, hi.z, hi.w};
    u32x2_t a1, a2, a3, b1, b2, b3;
    split3_x4(a, a1, a2, a3);
    split3_x4(b, b1, b2, b3);
    WFrag32 f;
    f.p[0] = __builtin_bit_cast(bf16x8_t, (u32x4_t){a1.x, a1.y, b1.x, b1.y});
    f.p[1] = __builtin_bit_cast(bf16x8_t, (u32x4_t){a2.x, a2.y, b2.x, b2.y});
    f.p[2] = __builtin_bit_cast(bf16x8_t, (u32x4_t){a3.x, a3.y, b3.x, b3.y});
    return f;
}
__device__ __forceinline__ WFrag16 load_wfrag16(const float* __restrict__ W, int K, int k0) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const float4 v = *reinterpret_cast<const float4*>(W + (size_t)r * K + k0 + 4 * g);
    const float a[4] = {v.x, v.y, v.z, v.w};
    u32x2_t a1, a2, a3;
    split3_x4(a, a1, a2, a3);
    WFrag16 f;
    f.p[0] = __builtin_bit_cast(s16x4_t, a1);
    f.p[1] = __builtin_bit_cast(s16x4_t, a2);
    f.p[2] = __builtin_bit_cast(s16x4_t, a3);
    return f;
}

// acc += W(16 x 32 block) * X(32 x 16 pixels); x1..x3 = the three pieces of the pixel operand (16 bytes per lane each)
__device__ __forceinline__ void mfma_split32(f32x4_t& acc, const WFrag32& w, bf16x8_t x1, bf16x8_t x2, bf16x8_t x3) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[0], x3, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[2], x1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[1], x2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[0], x2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[1], x1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[0], x1, acc, 0, 0, 0);
}
// K = 16 blocks: v_mfma_f32_16x16x16_bf16 costs the matrix pipe what the 32-deep form costs (16 busy cycles each), so two piece products
// share one 32-deep instruction, their operands concatenated along K:  w1 x3 + w3 x1 | w2 x2 + w2 x1 | w1 x2 + w1 x1  (small terms first)
#ifndef LG_SPLIT16_PAIR
#define LG_SPLIT16_PAIR 1
#endif
__device__ __forceinline__ bf16x8_t sb_cat8(s16x4_t lo, s16x4_t hi) {
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ void mfma_split16_pair(f32x4_t& acc, const WFrag16& w, s16x4_t x1, s16x4_t x2, s16x4_t x3) {
    const bf16x8_t x31 = sb_cat8(x3, x1), x21 = sb_cat8(x2, x1);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sb_cat8(w.p[0], w.p[2]), x31, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sb_cat8(w.p[1], w.p[1]), x21, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sb_cat8(w.p[0], w.p[0]), x21, acc, 0, 0, 0);
}
__device__ __forceinline__ void mfma_split16(f32x4_t& acc, const WFrag16& w, s16x4_t x1, s16x4_t x2, s16x4_t x3) {
#if LG_SPLIT16_PAIR
    mfma_split16_pair(acc, w, x1, x2, x3);
    return;
#endif
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w.p[0], x3, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w.p[2], x1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w.p[1], x2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w.p[0], x2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w.p[1], x1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w.p[0], x1, acc, 0, 0, 0);
}

// ---- NP = 2: TWO f16 pieces per operand (round 5).  v = hi + lo with hi = f16(v) and lo = f16(v - hi), both round-to-nearest: |lo| <= 2^-12 |hi|,
// so the pair carries 24 significant bits as long as hi is a normal f16 and lo >= 2^-24 -- i.e. for 2^-2 <= |v| < 2^16 exactly, below that
// with an ABSOLUTE error of <= 2^-25.  Three piece products (hi hi, hi lo, lo hi; the dropped lo lo is <= 2^-24 relative) instead of six, and
// 4.2 instead of 7.9 VALU cycles per value for the split (v_cvt_pk_f16_f32 twice per pair + one v_fma_mix_f32 per value, which forms the
// exact residual v - f16(v) in one instruction; tools/micro/valu_rates2.hip).  f16 has 5 exponent bits: the caller scales every operand
// by a power of two derived from a bound it can prove (k_attn_m: p <= 2^11 by construction, v by the window's own maximum).
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t sb_cvt_f16x2(float a, float b) {
    const f16x2_t h = {(_Float16)a, (_Float16)b};   // v_cvt_pk_f16_f32 (round to nearest even)
    return __builtin_bit_cast(uint32_t, h);
}
// a - f16 (low / high half of h), exact.  The asm statements read VALU results only (never a matrix-core accumulator: an asm statement gets
// none of the wait states the hardware needs behind an MFMA)
__device__ __forceinline__ float sb_res_lo(uint32_t h, float a) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(a));
    return r;
}
__device__ __forceinline__ float sb_res_hi(uint32_t h, float b) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(b));
    return r;
}
__device__ __forceinline__ void split2_x4(const float (&v)[4], u32x2_t& q1, u32x2_t& q2) {
    const uint32_t h01 = sb_cvt_f16x2(v[0], v[1]), h23 = sb_cvt_f16x2(v[2], v[3]);
    q1 = (u32x2_t){h01, h23};
    q2 = (u32x2_t){sb_cvt_f16x2(sb_res_lo(h01, v[0]), sb_res_hi(h01, v[1])), sb_cvt_f16x2(sb_res_lo(h23, v[2]), sb_res_hi(h23, v[3]))};
}
__device__ __forceinline__ f32x4_t sb_mfma_h(bf16x8_t a, bf16x8_t b, f32x4_t c) {   // operands carried in the bf16x8_t registers of the NP = 3 code; the bits are f16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// weight fragments as f16 pairs of W * wscale (wscale a power of two): p[0] = hi, p[1] = lo
__device__ __forceinline__ WFrag32 load_wfrag32_h2(const float* __restrict__ W, int K, int kb, float wscale) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const float4 lo = *reinterpret_cast<const float4*>(W + (size_t)r * K + kb * 32 + 8 * g);
    const float4 hi = *reinterpret_cast<const float4*>(W + (size_t)r * K + kb * 32 + 8 * g + 4);
    const float a[4] = {lo.x * wscale, lo.y * wscale, lo.z * wscale, lo.w * wscale}, b[4] = {hi.x * wscale, hi.y * wscale, hi.z * wscale, hi.w * wscale};
    u32x2_t a1, a2, b1, b2;
    split2_x4(a, a1, a2);
    split2_x4(b, b1, b2);
    WFrag32 f;
    f.p[0] = __builtin_bit_cast(bf16x8_t, (u32x4_t){a1.x, a1.y, b1.x, b1.y});
    f.p[1] = __builtin_bit_cast(bf16x8_t, (u32x4_t){a2.x, a2.y, b2.x, b2.y});
    f.p[2] = f.p[0];
    return f;
}
__device__ __forceinline__ WFrag16 load_wfrag16_h2(const float* __restrict__ W, int K, int k0, float wscale) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const float4 v = *reinterpret_cast<const float4*>(W + (size_t)r * K + k0 + 4 * g);
    const float a[4] = {v.x * wscale, v.y * wscale, v.z * wscale, v.w * wscale};
    u32x2_t a1, a2;
    split2_x4(a, a1, a2);
    WFrag16 f;
    f.p[0] = __builtin_bit_cast(s16x4_t, a1);
    f.p[1] = __builtin_bit_cast(s16x4_t, a2);
    f.p[2] = f.p[0];
    return f;
}

// ---- NP = number of pieces: 3 = the fp32-equivalent arithmetic above; 1 = plain bf16 operands (round-to-nearest), the opt-in
// `precision='bf16'` throughput mode: one MFMA per block instead of six, one conversion per value pair instead of 5.5 instructions.
template <int NP>
__device__ __forceinline__ void split_x4(const float (&v)[4], u32x2_t& q1, u32x2_t& q2, u32x2_t& q3) {
    if (NP == 3) {
        split3_x4(v, q1, q2, q3);
    } else if (NP == 2) {
        split2_x4(v, q1, q2);
        q3 = q1;   // unused
    } else {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        const bf16x2_t lo = {(__bf16)v[0], (__bf16)v[1]}, hi = {(__bf16)v[2], (__bf16)v[3]};      // v_cvt_pk_bf16_f32 (RNE)
        q1 = (u32x2_t){__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
        q2 = q1; q3 = q1;   // unused
    }
}
template <int NP>
__device__ __forceinline__ void mfma_np32(f32x4_t& acc, const WFrag32& w, bf16x8_t x1, bf16x8_t x2, bf16x8_t x3) {
    if (NP == 3) mfma_split32(acc, w, x1, x2, x3);
    else if (NP == 2) {   // small terms first: lo hi, hi lo, hi hi
        acc = sb_mfma_h(w.p[1], x1, acc);
        acc = sb_mfma_h(w.p[0], x2, acc);
        acc = sb_mfma_h(w.p[0], x1, acc);
    } else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.p[0], x1, acc, 0, 0, 0);
}
template <int NP>
__device__ __forceinline__ void mfma_np16(f32x4_t& acc, const WFrag16& w, s16x4_t x1, s16x4_t x2, s16x4_t x3) {
    if (NP == 3) mfma_split16(acc, w, x1, x2, x3);
    else if (NP == 2) {   // K = 16: two piece products share one 32-deep instruction (operands concatenated along K): w_lo x_hi + w_hi x_lo | w_hi x_hi
        const s16x4_t z = {0, 0, 0, 0};
        acc = sb_mfma_h(sb_cat8(w.p[1], w.p[0]), sb_cat8(x1, x2), acc);
        acc = sb_mfma_h(sb_cat8(w.p[0], z), sb_cat8(x1, z), acc);
    } else acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w.p[0], x1, acc, 0, 0, 0);
}
// weight fragments for NP = 1: piece 0 rounded to nearest (pieces 1, 2 unused)
__device__ __forceinline__ WFrag32 load_wfrag32_rne(const float* __restrict__ W, int K, int kb) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const float4 lo = *reinterpret_cast<const float4*>(W + (size_t)r * K + kb * 32 + 8 * g);
    const float4 hi = *reinterpret_cast<const float4*>(W + (size_t)r * K + kb * 32 + 8 * g + 4);
    WFrag32 f;
    f.p[0] = (bf16x8_t){(__bf16)lo.x, (__bf16)lo.y, (__bf16)lo.z, (__bf16)lo.w, (__bf16)hi.x, (__bf16)hi.y, (__bf16)hi.z, (__bf16)hi.w};
    f.p[1] = f.p[0]; f.p[2] = f.p[0];
    return f;
}
__device__ __forceinline__ WFrag16 load_wfrag16_rne(const float* __restrict__ W, int K, int k0) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const float4 v = *reinterpret_cast<const float4*>(W + (size_t)r * K + k0 + 4 * g);
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    const bf16x4_t h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    WFrag16 f;
    f.p[0] = __builtin_bit_cast(s16x4_t, h);
    f.p[1] = f.p[0]; f.p[2] = f.p[0];
    return f;
}

// fragment `f` of a PRE-SPLIT weight (k_split_w layout, k_ffn_x32.hip: [fragment][piece][lane] 16-byte units): one coalesced 1 KB
// read per piece
template <int NP = 3>
__device__ __forceinline__ WFrag32 ld_wfrag(const u32x4_t* __restrict__ base, int f) {
    const int lane = threadIdx.x & 63;
    WFrag32 w;
    w.p[0] = __builtin_bit_cast(bf16x8_t, base[(f * 3 + 0) * 64 + lane]);
    w.p[1] = NP >= 2 ? __builtin_bit_cast(bf16x8_t, base[(f * 3 + 1) * 64 + lane]) : w.p[0];
    w.p[2] = NP == 3 ? __builtin_bit_cast(bf16x8_t, base[(f * 3 + 2) * 64 + lane]) : w.p[0];   // (unused pieces are not read)
    return w;
}
