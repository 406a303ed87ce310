// k_ffn1_bwd_xs: the pixelwise half of the feed_forward backward at e = 16 (reference models/common/LGT.py:91-109 with the pre_norm /
// residual wrappers :45-61) with EVERYTHING that hangs off dh2 in one pass, on the bf16 matrix pipe in fp32-equivalent split
// arithmetic (split_bf16.h):
//
//     h1  = W1 LN(x) + b1                 (re-computed: the forward no longer saves it)
//     dh1 = (W2^T dh2) * gelu'(h1)
//     dx  = dy + LN^T (W1^T dh1)          + d gamma, d beta of the LayerNorm
//     dW2 += dh2 (x) gelu(h1),  db2 += dh2,  dW1 += dh1 (x) LN(x),  db1 += dh1        (pixel sums)
//
// It replaces k_ffn1_bwd<16> (f32 MFMA: 101 us, reads dh2 + h1 + x + dy) AND the 64 x 64 weight-gradient launch k_wgrad_t<4,4>
// (f32 MFMA: 68 us, reads dh2 + h1 again): dh2 is read once, h1 never (profiles/r03_*).
//
// Orientation: PIXELS ride on the A side of the MFMA, weights on the B side, so a lane ends up with one hidden CHANNEL (column r
// of the wave's 16-channel block) for four consecutive PIXELS (rows 4 g .. 4 g + 3).  That is exactly the operand layout of the
// weight-gradient products, whose K dimension is the pixel axis: gelu(h1) and dh1 go from the accumulator registers straight
// into dW2's B operand and dW1's A operand (v_mfma_f32_16x16x16_bf16, k = 4 g + j) without touching LDS.  The operands that do
// live in LDS ([pixel][channel] images of the bf16 pieces of dh2 and LN(x)) are read by rows for the data GEMMs and by columns
// (ds_read_b64_tr_b16, the hardware transpose read) for the weight gradients: one image serves both.
// Wave w owns hidden channels [16 w, 16 w + 16): its dW1 rows, its dW2 columns and its K = 16 slice of W1^T dh1, whose four
// partial sums meet in LDS in front of the LayerNorm backward.
//
// Tile = 64 pixels, persistent workgroups deal tiles round-robin, next tile's operands in flight in registers.
// e = 16: 4 waves, LDS 55 KB (two workgroups per CU): D2 pieces [3][64][72] bf16 | LN(x) pieces [3][64][16] | per-wave dh1^T [4][3][16][16] |
// W1^T dh1 partials [4][64][16] fp32.  e = 32 (round 3: level 1 of the 4-band net, level 0 of the 8-band net): the same kernel with 8 waves
// (N1 / 16 hidden-channel blocks), 139 KB, one workgroup per CU; it replaces k_ffn1_bwd_x32 + the 128 x 128 k_wgrad_t launches.
#include "kernels.h"
#include "bwd_kernels.h"
#include "split_bf16.h"
#include "hstore.h"

// In-kernel phase stamps (diagnostic build only, -DLG_STAMPS: tools/build_stamps.sh + tools/bwd_stamps.py): the four waves of workgroup 0
// write s_memtime at the phase boundaries of their third tile; no stamp executes in the product build.
#ifdef LG_STAMPS
__device__ unsigned long long g_kb_stamps[4 * 16];
#define STAMP(i) do { if (blockIdx.x == 0 && lane == 0 && stamp_on) g_kb_stamps[wave * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" __attribute__((visibility("default"))) int lg_debug_kb_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_kb_stamps), sizeof(g_kb_stamps));
}
#else
#define STAMP(i) do { } while (0)
#endif

namespace {

constexpr int NPX = 64;
template <int E>
struct KB {
    static constexpr int N1 = 4 * E, NW = N1 / 16, NT = 64 * NW, LPP = E / 4;
    // row pitch of the dh2 image in halves: 2 LDP bytes = 32 mod 64.  A 16-byte MFMA-operand read (lane 16 g + r: row r, K slice g) is served in
    // 16-lane groups that hold EIGHT rows of slice g and the OTHER eight rows of slice g + 1 (MI355X_MICROARCH, LDS table): conflict-free needs the
    // pitch to be 8 dwords mod 16 -- brute force over all pitches: 8, 24, 40, 56, 72 ... dwords; N1 + 8 halves (36 dwords at e = 16) was 2-way on
    // every such read
    static constexpr int LDP = N1 + 16;
    static constexpr int D2_PIECE = NPX * LDP;      // halves
    static constexpr int XN_PIECE = NPX * E;        // halves
    static constexpr int D1T_PIECE = 16 * 16;       // halves, per wave and piece
    static constexpr size_t OFF_XN = (size_t)3 * D2_PIECE * 2;
    static constexpr size_t OFF_D1T = OFF_XN + (size_t)3 * XN_PIECE * 2;
    static constexpr size_t OFF_RED = OFF_D1T + (size_t)NW * 3 * D1T_PIECE * 2;
    static constexpr size_t LDS_BYTES = OFF_RED + (size_t)NW * NPX * E * 4;
    // slab row of a workgroup: [dW2 N1 x N1 | db2 N1 | dW1 N1 x E | db1 N1 | d gamma E | d beta E]
    static constexpr int R_B2 = N1 * N1, R_W1 = R_B2 + N1, R_B1 = R_W1 + N1 * E, R_LG = R_B1 + N1, R_LB = R_LG + E, ROW = R_LB + E;
    static_assert(OFF_XN % 16 == 0 && OFF_D1T % 16 == 0 && OFF_RED % 16 == 0, "16-byte aligned LDS regions");
    static_assert(NT / (N1 / 4) == 16 && NT / LPP == NPX, "loader / LayerNorm thread maps");
};

typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
#ifndef LG_KB_DACHAINS
#define LG_KB_DACHAINS 0
#endif
#ifndef LG_KB_STAGGER
#define LG_KB_STAGGER 0
#endif
#ifndef LG_KB_FENCE
#define LG_KB_FENCE 1
#endif
#define KB_FENCE() do { if (LG_KB_FENCE) __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ bf16x8_t lds_x8(const uint16_t* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(p)); }
__device__ __forceinline__ s16x4_t lds_x4(const uint16_t* p) { return __builtin_bit_cast(s16x4_t, *reinterpret_cast<const u32x2_t*>(p)); }
// transposed read: the 16 lanes of a group hand in the addresses of a 4-row x 16-column block of 16-bit elements (lane 4 q + p: row q,
// columns 4 p .. 4 p + 3) and lane i gets column i of the four rows
__device__ __forceinline__ s16x4_t lds_tr4(const uint16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p);
}
// the six piece products of a fp32-equivalent 16x16x16 block: a, b = the three bf16 pieces of each operand (small terms first)
__device__ __forceinline__ void mfma6_16(f32x4_t& acc, const s16x4_t (&a)[3], const s16x4_t (&b)[3]) {
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b[0], acc, 0, 0, 0);
}
__device__ __forceinline__ void mfma6_32(f32x4_t& acc, const bf16x8_t (&a)[3], const bf16x8_t (&b)[3]) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
}
__device__ __forceinline__ bf16x8_t cat8(s16x4_t lo, s16x4_t hi) {
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// The six piece products of a 16-deep block as THREE 32-deep MFMAs: two piece products share one instruction, their operands
// concatenated along K (slots 0..3 of a lane = one piece pair, 4..7 = another; both operands use the same order, and any order of K is a
// sum).  v_mfma_f32_16x16x16_bf16 costs the matrix pipe what the 32-deep form costs (16 busy cycles each, profiles/r03_sq_counters_*),
// so this halves the pipe time of every K = 16 product: a1 b3 + a3 b1 | a2 b2 + a2 b1 | a1 b2 + a1 b1   (small terms first).
#ifndef LG_KB_PAIR
#define LG_KB_PAIR 1
#endif
__device__ __forceinline__ void mfma3_16(f32x4_t& acc, const s16x4_t (&a)[3], const s16x4_t (&b)[3]) {
    const bf16x8_t b31 = cat8(b[2], b[0]), b21 = cat8(b[1], b[0]);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(a[0], a[2]), b31, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(a[1], a[1]), b21, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(a[0], a[0]), b21, acc, 0, 0, 0);
}
// NP = 3: the six piece products (fp32-equivalent); NP = 1 (precision = 'bf16'): one product of round-to-nearest bf16 operands
// NP = 2 (round 5): f16 pairs (split_bf16.h), three piece products; every operand arrives multiplied by a power of two (see the kernel)
template <int NP>
__device__ __forceinline__ void mfmaN_16(f32x4_t& acc, const s16x4_t (&a)[3], const s16x4_t (&b)[3]) {
    if (NP == 3) { if (LG_KB_PAIR) mfma3_16(acc, a, b); else mfma6_16(acc, a, b); }
    else if (NP == 2) {   // K = 16: two piece products share one 32-deep instruction: a_lo b_hi + a_hi b_lo | a_hi b_hi
        const s16x4_t z = {0, 0, 0, 0};
        acc = sb_mfma_h(cat8(a[1], a[0]), cat8(b[0], b[1]), acc);
        acc = sb_mfma_h(cat8(a[0], z), cat8(b[0], z), acc);
    } else acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], b[0], acc, 0, 0, 0);
}
template <int NP>
__device__ __forceinline__ void mfmaN_32(f32x4_t& acc, const bf16x8_t (&a)[3], const bf16x8_t (&b)[3]) {
    if (NP == 3) mfma6_32(acc, a, b);
    else if (NP == 2) {
        acc = sb_mfma_h(a[1], b[0], acc);
        acc = sb_mfma_h(a[0], b[1], acc);
        acc = sb_mfma_h(a[0], b[0], acc);
    } else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
}
template <int NP>
__device__ __forceinline__ void split4(const float (&v)[4], s16x4_t (&p)[3]) {
    u32x2_t q1, q2, q3;
    split_x4<NP>(v, q1, q2, q3);
    p[0] = __builtin_bit_cast(s16x4_t, q1); p[1] = __builtin_bit_cast(s16x4_t, q2); p[2] = __builtin_bit_cast(s16x4_t, q3);
}

// generic K block of the data GEMMs: 16-deep (e = 16: K = e) or 32-deep
template <int NP>
__device__ __forceinline__ void ld3_x4(const uint16_t* p, int piece, s16x4_t (&o)[3]) {
    o[0] = lds_x4(p);
    if (NP == 3) { o[1] = lds_x4(p + piece); o[2] = lds_x4(p + 2 * piece); } else if (NP == 2) { o[1] = lds_x4(p + piece); o[2] = o[0]; } else { o[1] = o[0]; o[2] = o[0]; }
}
template <int NP>
__device__ __forceinline__ void ld3_x8(const uint16_t* p, int piece, bf16x8_t (&o)[3]) {
    o[0] = lds_x8(p);
    if (NP == 3) { o[1] = lds_x8(p + piece); o[2] = lds_x8(p + 2 * piece); } else if (NP == 2) { o[1] = lds_x8(p + piece); o[2] = o[0]; } else { o[1] = o[0]; o[2] = o[0]; }
}
template <int NP>
__device__ __forceinline__ void ld3_tr(const uint16_t* p, int piece, s16x4_t (&o)[3]) {
    o[0] = lds_tr4(p);
    if (NP == 3) { o[1] = lds_tr4(p + piece); o[2] = lds_tr4(p + 2 * piece); } else if (NP == 2) { o[1] = lds_tr4(p + piece); o[2] = o[0]; } else { o[1] = o[0]; o[2] = o[0]; }
}
template <int NP>
__device__ __forceinline__ WFrag32 wfrag32(const float* W, int K, int kb, float ws) { return NP == 3 ? load_wfrag32(W, K, kb) : (NP == 2 ? load_wfrag32_h2(W, K, kb, ws) : load_wfrag32_rne(W, K, kb)); }
template <int NP>
__device__ __forceinline__ WFrag16 wfrag16(const float* W, int K, int k0, float ws) { return NP == 3 ? load_wfrag16(W, K, k0) : (NP == 2 ? load_wfrag16_h2(W, K, k0, ws) : load_wfrag16_rne(W, K, k0)); }
// the power of two s with bound * s in [2^14, 2^15)
__device__ __forceinline__ float kb_pow2_below(float bound) { return __builtin_amdgcn_ldexpf(1.0f, 15 - __builtin_amdgcn_frexp_expf(bound)); }

template <int E, int NP>
__global__ __launch_bounds__(KB<E>::NT) __attribute__((amdgpu_waves_per_eu(2))) void k_ffn1_bwd_xs(Ffn1BwdXArgs a, long ntiles, int uneven) {
    using C = KB<E>;
    constexpr int N1 = C::N1, NW = C::NW, NT = C::NT, LDP = C::LDP, LPP = C::LPP, D2_PIECE = C::D2_PIECE, XN_PIECE = C::XN_PIECE, D1T_PIECE = C::D1T_PIECE;
    constexpr int KB2 = N1 / 32;       // 32-deep K blocks of dh2 W2
    constexpr int NE = E / 16;         // 16-wide blocks of the e-channel axis (columns of dW1, rows of W1^T dh1)
    constexpr bool BF = (NP == 1);     // plain-bf16 mode: dh2 is stored as bf16 (hstore.h) and is its own (single) piece
    // NP = 2: f16 pairs for the W2 products only (da = dh2 W2, K = 4 e: the bulk; dW2 = dh2^T gelu(h1), whose terms do not cancel).  LN(x), dh1, W1
    // and W1^T stay bf16 TRIPLES (NPE = 3): an f16 pair holds a value to 2^-24 relative, a triple all 24 bits, and the sums over pixels that dh1 and
    // LN(x) enter -- dW1, and through W1^T dh1 the LayerNorm gradients -- cancel to ~1 % of their terms: with pairs everywhere they came out at
    // 4e-6 .. 1.5e-5 of their norm against fp64 (triples: 5e-7; profiles/r05_ffn_bwd_err.txt)
    constexpr int NPE = NP == 2 ? 3 : NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint16_t* D2 = reinterpret_cast<uint16_t*>(smem_raw);                    // [3][NPX][LDP]   bf16 pieces of dh2
    uint16_t* XN = reinterpret_cast<uint16_t*>(smem_raw + C::OFF_XN);        // [3][NPX][E]     bf16 pieces of LN(x)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    uint16_t* D1T = reinterpret_cast<uint16_t*>(smem_raw + C::OFF_D1T) + wave * 3 * D1T_PIECE;   // [3][16 ch][16 px] of this wave
    float* red = reinterpret_cast<float*>(smem_raw + C::OFF_RED);            // [NW waves][NPX][E]

    // ---- weights of this wave's hidden-channel block, split once, register-resident
    // W1 rows [16 w, 16 w + 16) (B of h1 = LN(x) W1^T): one 16-deep block at e = 16, one 32-deep block at e = 32
    // ---- NP = 2: power-of-two scales of the f16-pair operands.  gelu(h1) and W2 carry the forward's static scales (k_ffn_prep.hip: bounds that hold
    // for every input); dh2 has no static bound: it is scaled from the launch-wide max |dh2| its producer (k_ffn_dw_bwd_xs) left in scales[6].
    // One scale per operand and LAUNCH, so the dW2 accumulators -- which run across all the tiles of the workgroup -- are a fixed power of two
    // times their true value, undone where they are written.
    float S_a1 = 1.f, S_w2 = 1.f, S_d2 = 1.f;
    if constexpr (NP == 2) {
        S_a1 = a.scales[1]; S_w2 = a.scales[4];
        S_d2 = kb_pow2_below(fmaxf(a.scales[6], 8.6736174e-19f));       // (2^-60: an all-zero gradient stays finite)
    }
    const float c_d1 = 1.0f / (S_d2 * S_w2);                   // behind the dh2 W2 accumulator
    WFrag16 w1f16;
    WFrag32 w1f32;
    if constexpr (E == 16) w1f16 = wfrag16<NPE>(a.w1 + (size_t)(wave * 16) * E, E, 0, 1.0f);
    else w1f32 = wfrag32<NPE>(a.w1 + (size_t)(wave * 16) * E, E, 0, 1.0f);
    WFrag32 w2f[KB2];                  // W2^T rows [16 w, 16 w + 16) (B of da1 = dh2 W2)
#pragma unroll
    for (int kb = 0; kb < KB2; ++kb) w2f[kb] = wfrag32<NP>(a.w2t + (size_t)(wave * 16) * N1, N1, kb, S_w2);
    WFrag16 w1tf[NE];                  // W1^T [16 rb + r][16 w + 4 g ..] (A of this wave's K = 16 slice of W1^T dh1)
#pragma unroll
    for (int rb = 0; rb < NE; ++rb) w1tf[rb] = wfrag16<NPE>(a.w1t + (size_t)(rb * 16) * N1, N1, wave * 16, 1.0f);
    const float b1s = a.b1[wave * 16 + r];
    // LayerNorm role: thread = (pixel t / (e/4), channel quad t % (e/4))
    const int lpx = threadIdx.x / LPP, lq = threadIdx.x % LPP;
    const float4 lng = *reinterpret_cast<const float4*>(a.ln2g + 4 * lq), lnb = *reinterpret_cast<const float4*>(a.ln2b + 4 * lq);
    // loader role: thread = (pixel t / (N1/4) + 16 it, channel quad t % (N1/4))
    const int dq = threadIdx.x % (N1 / 4), dpx = threadIdx.x / (N1 / 4);

    f32x4_t acc2[NW], acc1[NE];        // dW2[16 nb + 4 g + v][16 w + r], dW1[16 w + 4 g + v][16 cb + r]
#pragma unroll
    for (int nb = 0; nb < NW; ++nb) acc2[nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cb = 0; cb < NE; ++cb) acc1[cb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float bs1 = 0.f;                                          // db1[16 w + r], this lane's pixels
    float4 bs2 = make_float4(0.f, 0.f, 0.f, 0.f);             // db2[4 dq ..], this thread's pixels
    float4 pg = make_float4(0.f, 0.f, 0.f, 0.f), pb = pg;     // d gamma / d beta [4 lq ..], this thread's pixels

    typename HS<BF>::raw4 d2n[4];    // raw bits: widened when they leave the prefetch registers
    float4 xnx, dyn;
    auto issue = [&](long tile_) {
        const long p0 = tile_ * NPX;
#pragma unroll
        for (int it = 0; it < 4; ++it) d2n[it] = HS<BF>::ldraw(a.dh2, (p0 + dpx + 16 * it) * N1 + 4 * dq);
        xnx = *reinterpret_cast<const float4*>(a.x + (p0 + lpx) * E + 4 * lq);
        dyn = *reinterpret_cast<const float4*>(a.dy + (p0 + lpx) * E + 4 * lq);
    };
    // uneven (launcher: two resident workgroups per CU, ntiles a multiple of 16 per pair): the workgroup a CU received first runs faster than its second (the SIMD
    // arbiter issues the older wave first; k_ffn_xr.hip has the measurement), so of every 16 tiles of a pair it takes `uneven`
    const int half = (int)gridDim.x >> 1, first = (int)blockIdx.x < half ? 1 : 0;
    const long per = uneven ? ntiles / (8 * (long)gridDim.x) : 0;                       // sixteenths of a pair's tiles
    const long nmine = uneven ? per * (first ? uneven : 16 - uneven) : (1l << 60);
    const long t0 = uneven ? (first ? (long)blockIdx.x : (long)half * per * uneven + ((long)blockIdx.x - half)) : (long)blockIdx.x;
    const long tstep = uneven ? half : (long)gridDim.x;
    if (t0 < ntiles) issue(t0);

#pragma unroll 1
    for (long tile = t0, kt = 0; tile < ntiles && kt < nmine; tile += tstep, ++kt) {
        const long p0 = tile * NPX;
#ifdef LG_STAMPS
        const bool stamp_on = kt == 2;
#endif
        STAMP(0);
        // ---- loader: dh2 -> pieces -> D2 ; LN(x) -> pieces -> XN (the previous tile's readers of both are behind its second barrier)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const float4 d = HS<BF>::widen(d2n[it]);
            bs2.x += d.x; bs2.y += d.y; bs2.z += d.z; bs2.w += d.w;
            uint16_t* dst = D2 + (dpx + 16 * it) * LDP + 4 * dq;
            if constexpr (BF) {
                *reinterpret_cast<u32x2_t*>(dst) = __builtin_bit_cast(u32x2_t, d2n[it]);   // the stored bf16 values ARE the operand
            } else {
                const float v[4] = {d.x * S_d2, d.y * S_d2, d.z * S_d2, d.w * S_d2};     // (S_d2 = 1 at NP = 3)
                u32x2_t q1, q2, q3;
                split_x4<NP>(v, q1, q2, q3);
                *reinterpret_cast<u32x2_t*>(dst) = q1;
                *reinterpret_cast<u32x2_t*>(dst + D2_PIECE) = q2;
                if (NP == 3) *reinterpret_cast<u32x2_t*>(dst + 2 * D2_PIECE) = q3;
            }
        }
        const float4 xv = xnx, dyv = dyn;
        const float mu = lane_group_sum<LPP>((xv.x + xv.y) + (xv.z + xv.w)) * (1.0f / E);
        const float c0 = xv.x - mu, c1 = xv.y - mu, c2 = xv.z - mu, c3 = xv.w - mu;
        const float rstd = __builtin_amdgcn_rsqf(lane_group_sum<LPP>((c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3)) * (1.0f / E) + LG_EPS);
        const float xh[4] = {c0 * rstd, c1 * rstd, c2 * rstd, c3 * rstd};
        {
            const float yv[4] = {xh[0] * lng.x + lnb.x, xh[1] * lng.y + lnb.y, xh[2] * lng.z + lnb.z, xh[3] * lng.w + lnb.w};
            u32x2_t q1, q2, q3;
            split_x4<NPE>(yv, q1, q2, q3);
            uint16_t* dst = XN + lpx * E + 4 * lq;
            *reinterpret_cast<u32x2_t*>(dst) = q1;
            if (NPE == 3) {
                *reinterpret_cast<u32x2_t*>(dst + XN_PIECE) = q2;
                *reinterpret_cast<u32x2_t*>(dst + 2 * XN_PIECE) = q3;
            }
        }
        if (tile + tstep < ntiles && kt + 1 < nmine) issue(tile + tstep);   // next tile's operands: in flight during the GEMM phase
        STAMP(1);
        __syncthreads();
        STAMP(2);
#if LG_KB_STAGGER
        // e = 32: the two waves of a SIMD (w and w + 4) leave the barrier together and would run their matrix bursts and their vector
        // bursts at the same time; half a block of delay for the second lets one wave's MFMAs run under the other's GELU / splitting
        if (NW == 8 && wave >= 4) __builtin_amdgcn_s_sleep(LG_KB_STAGGER);
#endif

        // ---- GEMM phase: four blocks of 16 pixels; this wave's 16 hidden channels
#pragma unroll 1
        for (int pbk = 0; pbk < NPX / 16; ++pbk) {
            // The phases below are kept apart by scheduling fences (KB_FENCE): on gfx950 a wave that alternates MFMAs and vector
            // instructions runs slower than the sum of the two (tools/micro/mfma_valu_overlap.hip: 482 + 525 us alone, 1178 us interleaved
            // in one wave, 650 us with the two kinds of work in different waves of the SIMD) -- bursts of one kind per wave let the OTHER
            // resident wave's vector work run under this wave's matrix work.  (Measured here: no difference, 101.3 vs 101.5 us; kept.)
            // ---- h1[px 4 g + v][ch 16 w + r] = LN(x) W1^T + b1 ; da1 = dh2 W2 (K = N1 in 32-deep blocks)
            f32x4_t h1 = (f32x4_t){b1s, b1s, b1s, b1s};
            if constexpr (E == 16) {
                s16x4_t xa[3];
                ld3_x4<NPE>(XN + (pbk * 16 + r) * E + 4 * g, XN_PIECE, xa);
                KB_FENCE();
                mfmaN_16<NPE>(h1, xa, w1f16.p);
            } else {
                bf16x8_t xa[3];
                ld3_x8<NPE>(XN + (pbk * 16 + r) * E + 8 * g, XN_PIECE, xa);
                KB_FENCE();
                mfmaN_32<NPE>(h1, xa, w1f32.p);
            }
            f32x4_t da = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            {
                const uint16_t* pd = D2 + (pbk * 16 + r) * LDP + 8 * g;
#if LG_KB_DACHAINS
                // K = N1 as independent accumulation chains (one per 32-deep block), summed at the end: a chain of 6 KB2 dependent MFMAs
                // otherwise
                f32x4_t dak[KB2];
#pragma unroll
                for (int kb = 0; kb < KB2; ++kb) {
                    dak[kb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                    bf16x8_t dk[3];
                    ld3_x8<NP>(pd + 32 * kb, D2_PIECE, dk);
                    mfmaN_32<NP>(dak[kb], dk, w2f[kb].p);
                }
#pragma unroll
                for (int kb = 0; kb < KB2; ++kb) da += dak[kb];
#else
#pragma unroll
                for (int kb = 0; kb < KB2; ++kb) {
                    bf16x8_t dk[3];
                    ld3_x8<NP>(pd + 32 * kb, D2_PIECE, dk);
                    mfmaN_32<NP>(da, dk, w2f[kb].p);
                }
#endif
            }
            KB_FENCE();
            // ---- gelu(h1), gelu'(h1) with one exponential for both; dh1 = da1 * gelu'(h1); pieces
            lg_v2f a01, a23, g01, g23;
            gelu2_both_t<NP == 1>((lg_v2f){h1[0], h1[1]}, a01, g01);
            gelu2_both_t<NP == 1>((lg_v2f){h1[2], h1[3]}, a23, g23);
            if (NP == 2) { a01 *= S_a1; a23 *= S_a1; g01 *= c_d1; g23 *= c_d1; }      // gelu(h1) as an f16-pair operand; dh1 = (accumulator / (S_d2 S_w2)) gelu'(h1), true scale
            const float a1v[4] = {a01.x, a01.y, a23.x, a23.y};
            const float d1v[4] = {da[0] * g01.x, da[1] * g01.y, da[2] * g23.x, da[3] * g23.y};
            bs1 += (d1v[0] + d1v[1]) + (d1v[2] + d1v[3]);
            s16x4_t a1p[3], d1p[3];
            split4<NP>(a1v, a1p);
            split4<NPE>(d1v, d1p);
            KB_FENCE();
            // ---- dW1[16 w + .][.] += dh1^T LN(x) (A = dh1 from the registers, B = LN(x) read by columns);
            //      dW2[.][16 w + .] += dh2^T gelu(h1) (A = dh2 read by columns, B = gelu(h1) from the registers)
            {
                const uint16_t* px = XN + (pbk * 16 + 4 * g + (r >> 2)) * E + 4 * (r & 3);
#pragma unroll
                for (int cb = 0; cb < NE; ++cb) {
                    s16x4_t xt[3];
                    ld3_tr<NPE>(px + 16 * cb, XN_PIECE, xt);
                    mfmaN_16<NPE>(acc1[cb], d1p, xt);
                }
                const uint16_t* pt = D2 + (pbk * 16 + 4 * g + (r >> 2)) * LDP + 4 * (r & 3);
#pragma unroll
                for (int nb = 0; nb < NW; ++nb) {
                    s16x4_t dt[3];
                    ld3_tr<NP>(pt + 16 * nb, D2_PIECE, dt);
                    mfmaN_16<NP>(acc2[nb], dt, a1p);
                }
            }
            KB_FENCE();
            // this wave's K = 16 slice of W1^T dh1: dh1 -> [channel][pixel] in the wave's own LDS region, read back by columns as the B operand
            {
                uint16_t* dst = D1T + r * 16 + 4 * g;
                *reinterpret_cast<u32x2_t*>(dst) = __builtin_bit_cast(u32x2_t, d1p[0]);
                if (NPE == 3) {
                    *reinterpret_cast<u32x2_t*>(dst + D1T_PIECE) = __builtin_bit_cast(u32x2_t, d1p[1]);
                    *reinterpret_cast<u32x2_t*>(dst + 2 * D1T_PIECE) = __builtin_bit_cast(u32x2_t, d1p[2]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                s16x4_t dtp[3];
                ld3_tr<NPE>(D1T + (4 * g + (r >> 2)) * 16 + 4 * (r & 3), D1T_PIECE, dtp);
#pragma unroll
                for (int rb = 0; rb < NE; ++rb) {
                    f32x4_t o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                    mfmaN_16<NPE>(o, w1tf[rb].p, dtp);     // o[v] = (W1^T dh1)[out channel 16 rb + 4 g + v][pixel r], hidden channels 16 w .. 16 w + 15 only
                    *reinterpret_cast<float4*>(red + ((size_t)wave * NPX + pbk * 16 + r) * E + 16 * rb + 4 * g) = make_float4(o[0], o[1], o[2], o[3]);
                }
                __builtin_amdgcn_wave_barrier();      // D1T is rewritten by the next pixel block
            }
        }
        STAMP(3);
        __syncthreads();
        STAMP(4);

        // ---- LayerNorm backward + residual (thread = pixel lpx, channels 4 lq ..): sum of the NW K slices, then the usual two moments
        {
            const float* rp = red + (size_t)lpx * E + 4 * lq;
            float dl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w2 = 0; w2 < NW; w2 += 2) {
                const float4 s0 = *reinterpret_cast<const float4*>(rp + (size_t)w2 * NPX * E), s1 = *reinterpret_cast<const float4*>(rp + (size_t)(w2 + 1) * NPX * E);
                dl[0] += s0.x + s1.x; dl[1] += s0.y + s1.y; dl[2] += s0.z + s1.z; dl[3] += s0.w + s1.w;
            }
            pg.x += dl[0] * xh[0]; pg.y += dl[1] * xh[1]; pg.z += dl[2] * xh[2]; pg.w += dl[3] * xh[3];
            pb.x += dl[0]; pb.y += dl[1]; pb.z += dl[2]; pb.w += dl[3];
            const float dxh[4] = {dl[0] * lng.x, dl[1] * lng.y, dl[2] * lng.z, dl[3] * lng.w};
            const float m1 = lane_group_sum<LPP>((dxh[0] + dxh[1]) + (dxh[2] + dxh[3])) * (1.0f / E);
            const float m2 = lane_group_sum<LPP>((dxh[0] * xh[0] + dxh[1] * xh[1]) + (dxh[2] * xh[2] + dxh[3] * xh[3])) * (1.0f / E);
            *reinterpret_cast<float4*>(a.dx + (p0 + lpx) * E + 4 * lq) =
                make_float4(dyv.x + rstd * (dxh[0] - m1 - xh[0] * m2), dyv.y + rstd * (dxh[1] - m1 - xh[1] * m2),
                            dyv.z + rstd * (dxh[2] - m1 - xh[2] * m2), dyv.w + rstd * (dxh[3] - m1 - xh[3] * m2));
        }
        STAMP(5);
        // (the next tile's loader writes D2 / XN, which this tile's GEMM phase finished reading before the barrier above; `red` is
        // rewritten only behind the next tile's first barrier)
    }

    // ---- this workgroup's partial sums -> its slab row [dW2 N1 x N1 | db2 N1 | dW1 N1 x e | db1 N1 | d gamma e | d beta e]
    float* row = a.slab + (size_t)blockIdx.x * C::ROW;
    const float inv_w2 = 1.0f / (S_d2 * S_a1);     // (1 at NP = 1, 3)
#pragma unroll
    for (int nb = 0; nb < NW; ++nb)
#pragma unroll
        for (int v = 0; v < 4; ++v) row[(16 * nb + 4 * g + v) * N1 + 16 * wave + r] = acc2[nb][v] * inv_w2;
#pragma unroll
    for (int cb = 0; cb < NE; ++cb)
#pragma unroll
        for (int v = 0; v < 4; ++v) row[C::R_W1 + (16 * wave + 4 * g + v) * E + 16 * cb + r] = acc1[cb][v];
    {
        float s = bs1;
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (g == 0) row[C::R_B1 + 16 * wave + r] = s;
    }
    __syncthreads();                     // every wave is done with the tile loop's LDS
    float* sc = reinterpret_cast<float*>(smem_raw);   // [16][N1] db2 partials | [NPX][e] d gamma | [NPX][e] d beta
    *reinterpret_cast<float4*>(sc + dpx * N1 + 4 * dq) = bs2;
    *reinterpret_cast<float4*>(sc + 16 * N1 + lpx * E + 4 * lq) = pg;
    *reinterpret_cast<float4*>(sc + 16 * N1 + NPX * E + lpx * E + 4 * lq) = pb;
    __syncthreads();
    if (threadIdx.x < N1) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += sc[k * N1 + threadIdx.x];
        row[C::R_B2 + threadIdx.x] = s;
    } else if (threadIdx.x < N1 + 2 * E) {
        const int i = threadIdx.x - N1, which = i / E, c = i % E;
        const float* src = sc + 16 * N1 + which * NPX * E + c;
        float s = 0.f;
        for (int k = 0; k < NPX; ++k) s += src[k * E];
        row[(which ? C::R_LB : C::R_LG) + c] = s;
    }
}

template <int E>
int launch_t(const Ffn1BwdXArgs& a, hipStream_t s) {
    using C = KB<E>;
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn1_bwd_xs<E, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn1_bwd_xs<E, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn1_bwd_xs<E, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        if (e != hipSuccess) { lg_set_error("ffn1_bwd_xs: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    const long ntiles = a.P / NPX;
    const int cap = ffn1_bwd_x_wgs(E);
    const int grid = (int)(ntiles < cap ? ntiles : cap);
#ifndef LG_FFN1B_UNEVEN
#define LG_FFN1B_UNEVEN 9
#endif
    const int uneven = (E == 16 && grid == 512 && ntiles % (8 * (long)grid) == 0 && LG_FFN1B_UNEVEN != 8) ? LG_FFN1B_UNEVEN : 0;   // the measured shape only (two workgroups per CU)
    if (a.hbf) k_ffn1_bwd_xs<E, 1><<<grid, C::NT, C::LDS_BYTES, s>>>(a, ntiles, uneven);      // precision = 'bf16': plain bf16 operands, dh2 stored as bf16
    else if (a.scales) k_ffn1_bwd_xs<E, 2><<<grid, C::NT, C::LDS_BYTES, s>>>(a, ntiles, uneven);   // f16 pairs, scaled operands
    else k_ffn1_bwd_xs<E, 3><<<grid, C::NT, C::LDS_BYTES, s>>>(a, ntiles, uneven);
    LG_CHECK_LAUNCH();
    // the slab rows, summed in a fixed order by the deferred reduce launch
    ReduceJob j;
    j.dst2 = nullptr; j.nslices = grid; j.slice_stride = C::ROW;
    auto job = [&](int off, float* dst, int rows, int cols) {
        j.slab = a.slab + off; j.dst = dst; j.rows = rows; j.cols = cols; j.row_stride = cols; j.ld = cols; j.rows_valid = rows; j.cols_valid = cols;
        return launch_reduce_job(j, s);
    };
    int rc = job(0, a.d_w2, C::N1, C::N1);
    if (!rc) rc = job(C::R_B2, a.d_b2, 1, C::N1);
    if (!rc) rc = job(C::R_W1, a.d_w1, C::N1, E);
    if (!rc) rc = job(C::R_B1, a.d_b1, 1, C::N1);
    if (!rc) rc = job(C::R_LG, a.d_ln2g, 1, E);
    if (!rc) rc = job(C::R_LB, a.d_ln2b, 1, E);
    return rc;
}

}   // namespace

size_t ffn1_bwd_x_slab_floats(int e) { return (size_t)ffn1_bwd_x_wgs(e) * (e == 16 ? KB<16>::ROW : KB<32>::ROW); }
bool ffn1_bwd_x32_built() { return true; }

int launch_ffn1_bwd_xs(int e, const Ffn1BwdXArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_FFN1_BWD, s);
    if (e != 16 && e != 32) { lg_set_error("ffn1_bwd_xs: e=%d unsupported", e); return -1; }
    if (a.P <= 0 || a.P % NPX) { lg_set_error("ffn1_bwd_xs: pixel count %ld is not a multiple of %d", a.P, NPX); return -2; }
    if (!a.dh2 || !a.x || !a.dy || !a.dx || !a.slab || !a.w1 || !a.w1t || !a.w2t || !a.b1 || !a.ln2g || !a.ln2b) { lg_set_error("ffn1_bwd_xs: null argument"); return -2; }
    // e = 32 (one 8-wave workgroup per CU): behind the strip-walking spatial half it replaces k_ffn1_bwd_x32 AND its two weight-gradient
    // launches (round 5: 14.36 -> 14.20 ms per c3 step) -- the pixelwise half alone was measured slower than the pair in round 4
    if (e == 32) return launch_t<32>(a, s);
    return launch_t<16>(a, s);
}
