// Local (window) mixer + LGMixer projection + residual on the gfx950 MATRIX pipe (round 5).
// Reference: models/common/LGT.py:112-146 (local_mixer), 183-219 (LGMixer), 45-61,231-248 (pre_norm / residual).
//
// One wavefront = one 8x8 window.  Every product of the half-block is an MFMA -- to_qkv, Q K^T, P V and proj -- and NO activation
// crosses LDS: one lane map carries the window through all four products.
//
//   lane l = (g = l >> 4, c = l & 15);  token tile t = 0..3: token 16 t + c (window row 2 t + (c >> 3), column c & 7);
//   of its token a lane holds the 16-byte channel chunks {4 m + g}: channels 16 m + 4 g .. + 3.
//
// That is (a) a coalesced 16-byte access per lane and chunk for x, the residual and y, (b) the B operand of v_mfma_f32_16x16x32
// when the WEIGHTS sit on the A side -- lane (g, c) supplies k-slots 8 g .. 8 g + 7 of column c, and which (channel, piece) a slot
// means is ours to choose as long as the weight fragment agrees -- and (c) the C layout of such a product: rows 4 g + v, column c.
// So proj's output lands on the lanes that hold the residual x, and, with the rows of to_qkv's weight tiles PERMUTED so that row
// 4 g + v is a q / k channel of group g's share of the head dimension, q and k of a token come out in the lane that will feed them
// to Q K^T as k-slots (the sum over the four lane groups is the sum over the head dimension).  v is made by the same instruction
// with the operand roles swapped (A = LN(x) fragment, B = Wv^T): C = [token 4 g + v][channel c], which is exactly the A operand of
// O^T = V^T P^T (row = channel on the lane, k = keys in the lane's registers) -- and the scores S^T = K Q^T leave the matrix pipe as
// [key 4 g + v][query c], exactly P^T's B operand.  O^T's C layout is again (channel chunk g, token c): the cat(o1, o2) fragment
// of proj.  Softmax: 16 in-lane values per (head, query tile) and two v_permlane*_swap steps across the lane groups.
//
// Arithmetic (NP = 3, the default): to_qkv, Q K^T and proj in the split-bf16 form of split_bf16.h (every fp32 operand = three bf16
// pieces, six piece products, fp32 accumulation -- at the head dimension of the 4-band net all six products of a score tile are ONE
// 32-deep instruction); P V on the f16 pipe with TWO pieces per operand: p in (0, 2^11] and v scaled by a per-window power of two
// into f16's range are each hi + lo with |lo| <= 2^-12 |hi| (round-to-nearest twice = 24 significant bits, v_fma_mix_f32 forms
// the residual in one instruction), three piece products.  pos_emb enters as the initial accumulator of the score tile.
// NP = 1 (precision = 'bf16'): one round-to-nearest piece everywhere.
#include "kernels.h"
#include "split_bf16.h"


namespace am {

// piece tables of the six products (small terms are summed by the matrix core in its own order)
__host__ __device__ constexpr int pw(int prod) { return prod == 0 ? 1 : prod == 1 ? 1 : prod == 2 ? 2 : prod == 3 ? 1 : prod == 4 ? 3 : 2; }   // A side
__host__ __device__ constexpr int px(int prod) { return prod == 0 ? 1 : prod == 1 ? 2 : prod == 2 ? 1 : prod == 3 ? 3 : prod == 4 ? 1 : 2; }   // B side
// NP = 2 (round 6): f16 PAIRS, three products: A side (lo, hi, hi) against B side (hi, lo, hi) -- piece 1 = hi, 2 = lo
__host__ __device__ constexpr int pw2(int prod) { return prod == 0 ? 2 : 1; }
__host__ __device__ constexpr int px2(int prod) { return prod == 1 ? 2 : 1; }
__host__ __device__ constexpr int nprod(int np) { return np == 3 ? 6 : (np == 2 ? 3 : 1); }
template <int NP> __host__ __device__ constexpr int pwN(int prod) { return NP == 2 ? pw2(prod) : pw(prod); }
template <int NP> __host__ __device__ constexpr int pxN(int prod) { return NP == 2 ? px2(prod) : px(prod); }
__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// Built with -fno-honor-nans (Makefile): fmaxf() on values the compiler cannot prove canonical (matrix-core results, v_exp_f32 results)
// otherwise gets a canonicalising v_max_f32 x, x in front of every operand (IEEE mode).  NOT inline asm: an asm statement that reads a
// matrix-core result gets none of the wait states the hardware needs between the two (NaNs on some waves of some launches).
__device__ __forceinline__ float vmax2(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float vmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float xg_sum(float v) {   // sum over the four lane groups (lanes c, c + 16, c + 32, c + 48); every lane gets it
    u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r.x) + __uint_as_float(r.y);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float xg_max(float v) {
    u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = vmax2(__uint_as_float(r.x), __uint_as_float(r.y));
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax2(__uint_as_float(r.x), __uint_as_float(r.y));
}
__device__ __forceinline__ float wave_max(float v) {   // max over all 64 lanes
    v = xg_max(v);
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) v = vmax2(v, __shfl_xor(v, o));
    return v;
}

// fp32 patterns whose high halves are the 16-bit pieces of v (NP = 3: exact three-way bf16 split; NP = 2: f16 hi + lo; NP = 1: bf16 round to nearest)
template <int NP>
struct Pat { uint32_t p[3]; };
template <int NP>
__device__ __forceinline__ Pat<NP> pat_of(float v) {
    Pat<NP> r;
    if (NP == 3) {
        const Split3 s = split3(v);
        r.p[0] = s.p1; r.p[1] = s.p2; r.p[2] = s.p3;
    } else if (NP == 2) {   // f16 pair of the (scaled) value, each piece in the HIGH half of its pattern: v_cvt_pk_f16_f32 (0, v), the exact residual, again
        r.p[0] = sb_cvt_f16x2(0.0f, v);
        r.p[1] = sb_cvt_f16x2(0.0f, sb_res_hi(r.p[0], v));
        r.p[2] = 0;
    } else {
        const __bf16 h = (__bf16)v;
        r.p[0] = (uint32_t)__builtin_bit_cast(uint16_t, h) << 16; r.p[1] = 0; r.p[2] = 0;
    }
    return r;
}

// f16 pairs: round to nearest, and the exact residual a - f16(a) in one instruction each (split_bf16.h, NP = 2)
__device__ __forceinline__ uint32_t cvt_f16x2(float a, float b) { return sb_cvt_f16x2(a, b); }
__device__ __forceinline__ float res_lo(uint32_t h, float a) { return sb_res_lo(h, a); }
__device__ __forceinline__ float res_hi(uint32_t h, float b) { return sb_res_hi(h, b); }

__device__ __forceinline__ f32x4_t mfma_bf(u32x4_t a, u32x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma_h(u32x4_t a, u32x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <int NP>
__device__ __forceinline__ f32x4_t mfma_n(u32x4_t a, u32x4_t b, f32x4_t c) { return NP == 2 ? mfma_h(a, b, c) : mfma_bf(a, b, c); }

// Operand fragment k of a lane that holds CH channels as piece patterns pat[ch].p[piece]: slot s = 8 k + j means product s / CH
// (piece table A or B side) of channel s % CH; slots past the last product are zero.  CH = 1: a dword pairs two products of the one
// channel; CH >= 2: a dword is a channel pair of one product.
template <int NP, int CH, bool ASIDE>
__device__ __forceinline__ u32x4_t build_frag(const Pat<NP> (&pat)[CH], int k) {
    constexpr int NPROD = nprod(NP);
    uint32_t d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = 8 * k + 2 * i;
        if constexpr (CH == 1) {
            const int p0 = s, p1 = s + 1;
            const uint32_t lo = p0 < NPROD ? pat[0].p[(ASIDE ? pwN<NP>(p0) : pxN<NP>(p0)) - 1] : 0u;
            const uint32_t hi = p1 < NPROD ? pat[0].p[(ASIDE ? pwN<NP>(p1) : pxN<NP>(p1)) - 1] : 0u;
            d[i] = (p0 < NPROD) ? pack_hi16(lo, hi) : 0u;
        } else {
            const int prod = s / CH, ch = s % CH;
            if (prod < NPROD) {
                const int pc = (ASIDE ? pwN<NP>(prod) : pxN<NP>(prod)) - 1;
                d[i] = pack_hi16(pat[ch].p[pc], pat[ch + 1].p[pc]);
            } else d[i] = 0u;
        }
    }
    return (u32x4_t){d[0], d[1], d[2], d[3]};
}

// Weight fragments, staged once per workgroup in LDS as [tile][k][lane] 16-byte units (a conflict-free ds_read_b128 per use).
// Element (lane = (g, r), slot s = 8 k + j) of tile `tile`: product s / CH (A-side piece table: the weights are always the "W" factor),
// input channel 16 (chl >> 2) + 4 g + (chl & 3) with chl = s % CH -- the lane map of the activation fragments -- of output channel
// oc_of(tile, r) (negative: a zero row).  ALL threads of the block call it.
template <int NP, int CH, int NK, int NTILE>
struct WStage {
    static constexpr int NPROD = nprod(NP);
    static constexpr int NF = NTILE * NK;              // fragments; thread tid owns dword (tid & 3) of lane (tid >> 2) in every one of them
    static constexpr int CHUNK = NF < 16 ? NF : 16;    // loads in flight per thread and round (every load of a round is requested before its first store)
    static constexpr bool SPLIT = NF <= 16;            // few fragments: ld() requests them all and st() stores them, so that the caller can put the requests of
                                                       // EVERY table of its prologue in front of the first wait (one round trip instead of one per table)
    float w0[CHUNK], w1[CHUNK];
    template <class OC>
    __device__ __forceinline__ void ld_chunk(int f0, const float* __restrict__ W, int ldw, int kdim, OC oc_of) {
        const int i = threadIdx.x & 3, ln = threadIdx.x >> 2, g = ln >> 4, r = ln & 15;
#pragma unroll
        for (int j = 0; j < CHUNK; ++j) {
            const int f = f0 + j < NF ? f0 + j : NF - 1, k = f % NK, tile = f / NK;
            const int s = 8 * k + 2 * i, prod = s / CH, chl = s % CH;
            const int oc = oc_of(tile, r), ic = 16 * (chl >> 2) + 4 * g + (chl & 3);
            const bool ok = prod < NPROD && oc >= 0 && ic < kdim;
            const float* src = W + (ok ? (size_t)oc * ldw + ic : 0);
            w0[j] = src[0]; w1[j] = src[1];     // unconditional (index 0, 1 for the zero slots): no exec-masked branch, no wait at a join
        }
    }
    template <class OC>
    __device__ __forceinline__ void st_chunk(int f0, u32x4_t* dst, int kdim, OC oc_of, float wscale) {
        uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
        const int i = threadIdx.x & 3, ln = threadIdx.x >> 2, g = ln >> 4, r = ln & 15;
#pragma unroll
        for (int j = 0; j < CHUNK; ++j) {
            if (f0 + j >= NF) break;
            const int f = f0 + j, k = f % NK, tile = f / NK;
            const int s = 8 * k + 2 * i, prod = s / CH, chl = s % CH;
            const int oc = oc_of(tile, r), ic = 16 * (chl >> 2) + 4 * g + (chl & 3);
            const bool ok = prod < NPROD && oc >= 0 && ic < kdim;
            const int pc = pwN<NP>(prod < NPROD ? prod : 0) - 1;
            const Pat<NP> a0 = pat_of<NP>(NP == 2 ? w0[j] * wscale : w0[j]), a1 = pat_of<NP>(NP == 2 ? w1[j] * wscale : w1[j]);
            const uint32_t val = pc == 0 ? pack_hi16(a0.p[0], a1.p[0]) : (pc == 1 ? pack_hi16(a0.p[1], a1.p[1]) : pack_hi16(a0.p[2], a1.p[2]));   // (no runtime index: that is scratch)
            d32[f * 256 + threadIdx.x] = ok ? val : 0u;
        }
    }
    template <class OC>
    __device__ __forceinline__ void ld(const float* __restrict__ W, int ldw, int kdim, OC oc_of) {
        if constexpr (SPLIT) ld_chunk(0, W, ldw, kdim, oc_of);
    }
    template <class OC>
    __device__ __forceinline__ void st(u32x4_t* dst, const float* __restrict__ W, int ldw, int kdim, OC oc_of, float wscale = 1.0f) {
        if constexpr (SPLIT) st_chunk(0, dst, kdim, oc_of, wscale);
        else {
#pragma unroll
            for (int f0 = 0; f0 < NF; f0 += CHUNK) {
                ld_chunk(f0, W, ldw, kdim, oc_of);
                st_chunk(f0, dst, kdim, oc_of, wscale);
            }
        }
    }
};

template <int HC>
struct Geo {
    static constexpr int E = 2 * HC, D = HC / 2, DG = D / 4;      // DG: channels of a head that one lane group owns
    static constexpr int NCH = E / 16;                            // 16-byte chunks of x / cat / y per lane and token
    static constexpr int NY = HC >= 16 ? HC / 16 : 1;             // chunks of the local half per lane (HC = 8: lane groups 0, 1 only)
    static constexpr int MTQK = D / 4;                            // row tiles of the q / k part of to_qkv
    static constexpr int NTV = HC >= 16 ? HC / 16 : 1;            // column tiles of its v part
    // q / k row `idx` = 4 mt + v of lane group gq: (is_q, head, dd) with head-dimension index gq * DG + dd
    __host__ __device__ static constexpr int qk_oc(int mt, int r) {
        const int gq = r >> 2, idx = mt * 4 + (r & 3);
        const int isq = idx / (2 * DG), h = (idx / DG) % 2, dd = idx % DG;
        return (isq ? 0 : HC) + h * D + gq * DG + dd;
    }
    __host__ __device__ static constexpr int v_oc(int nt, int c) { return (HC == 8 && c >= 8) ? -1 : 2 * HC + 16 * nt + c; }
};

}  // namespace am

#ifdef LG_ATTN_STAMPS   // diagnostic build of tools/micro/attn_m_check.hip only: s_memtime ticks per phase, summed over the windows of every wave
__device__ unsigned long long am_stamps[1024][4][8];
#define AM_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t__ = __builtin_amdgcn_s_memtime(); st[i] += t__ - tprev; tprev = t__; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define AM_STAMP(i) do { } while (0)
#endif
template <int HC, int NP>
__global__ __launch_bounds__(256, (HC <= 16 ? 2 : 1)) void k_attn_m(AttnArgs a, int nwin, int nquads, int uneven) {
    using namespace am;
    using G = Geo<HC>;
    constexpr int E = G::E, D = G::D, DG = G::DG, NCH = G::NCH, NY = G::NY, MTQK = G::MTQK, NTV = G::NTV;
    // NP = 2 (round 6, the default): to_qkv and Q K^T on f16 PAIRS under static operand scales (a.scales: bounds from the block's weights,
    // k_ffn_prep.hip); proj stays on bf16 triples (its cat(o1, o2) operand has no static bound: o2 is the FFT mixer's output)
    constexpr int NPQ = NP == 2 ? 2 : NP, NPP = NP == 2 ? 3 : NP;
    constexpr int CHY = 4 * NY, NKQ = cdiv(nprod(NPQ) * CHY, 8);  // to_qkv: channels per lane, instructions per output tile
    constexpr int NKS = cdiv(nprod(NPQ) * DG, 8);                 // Q K^T: instructions per 16 x 16 score tile
    constexpr int CHC = 4 * NCH, NKP = cdiv(nprod(NPP) * CHC, 8); // proj
    constexpr int MTP = NCH;
    constexpr int NPV = NP >= 2 ? 2 : 1;                          // f16 pieces of p and v
    constexpr bool RELOADX = HC >= 32;                            // the residual x is read again (L2) instead of held across the window
    constexpr float LOG2E = 1.44269504088896340736f;
    extern __shared__ __attribute__((aligned(16))) u32x4_t smem4[];
    float4* sPos = reinterpret_cast<float4*>(smem4);              // [2][4 qt][4 kt][64 lanes]: the score tile's initial accumulator
    u32x4_t* sWqk = smem4 + 2 * 4 * 4 * 64;                       // [MTQK][NKQ][64]
    u32x4_t* sWv = sWqk + MTQK * NKQ * 64;                        // [NTV][NKQ][64]
    u32x4_t* sWp = sWv + NTV * NKQ * 64;                          // [MTP][NKP][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;

#ifdef LG_ATTN_STAMPS
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#endif
    // static operand scales of the pairs (powers of two: exact; taken out again by constants that were multiplications already)
    float sy = 1.f, sw = 1.f, sq = 1.f, sk = 1.f;
    if (NP == 2) { sy = a.scales[0]; sw = a.scales[1]; sq = a.scales[2]; sk = a.scales[3]; }
    const float syw = sy * sw, inv_yw = 1.0f / syw, sqk = sq * sk, inv_qk = 1.0f / sqk;
    // ---- once per (persistent) workgroup: pos_emb in fragment order, the weight fragments, the lane constants.  Every request of the prologue goes out
    // before its first wait (as pos -> store -> table -> store -> table -> ... it was seven dependent round trips per launch at HC = 8)
    float4 pv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int u = j * 256 + threadIdx.x, ln = u & 63, kt = (u >> 6) & 3, qt = (u >> 8) & 3, h = u >> 10;
        pv[j] = *reinterpret_cast<const float4*>(a.pos + ((h * 64 + 16 * qt + (ln & 15)) * 64 + 16 * kt + 4 * (ln >> 4)));
    }
    auto oc_qk = [](int t, int r) { return G::qk_oc(t, r); };
    auto oc_v = [](int t, int r) { return G::v_oc(t, r); };
    auto oc_p = [](int t, int r) { return 16 * t + r; };
    WStage<NPQ, CHY, NKQ, MTQK> wsq;
    WStage<NPQ, CHY, NKQ, NTV> wsv;
    WStage<NPP, CHC, NKP, MTP> wsp;
    wsq.ld(a.qkvw, HC, HC, oc_qk);
    wsv.ld(a.qkvw, HC, HC, oc_v);
    wsp.ld(a.projw, E, E, oc_p);
    // lane constants: biases as initial accumulators, LayerNorm affine of the lane's local-half channels
    float bqk[MTQK][4], bv[NTV], bp[MTP][4], gam[CHY], bet[CHY];
#pragma unroll
    for (int mt = 0; mt < MTQK; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) bqk[mt][v] = a.qkvb[G::qk_oc(mt, 4 * g + v)];
#pragma unroll
    for (int nt = 0; nt < NTV; ++nt) { const int oc = G::v_oc(nt, c); bv[nt] = a.qkvb[oc >= 0 ? oc : 0]; }
#pragma unroll
    for (int mt = 0; mt < MTP; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) bp[mt][v] = a.projb[16 * mt + 4 * g + v];
#pragma unroll
    for (int i = 0; i < CHY; ++i) { const int ch = 16 * (i >> 2) + 4 * g + (i & 3); gam[i] = a.ln1g[ch]; bet[i] = a.ln1b[ch]; }
    __builtin_amdgcn_sched_barrier(0);   // requests above, conversions and stores below
#pragma unroll
    for (int j = 0; j < 8; ++j)   // scores live in the log2 domain
        sPos[j * 256 + threadIdx.x] = make_float4(pv[j].x * (LOG2E * sqk), pv[j].y * (LOG2E * sqk), pv[j].z * (LOG2E * sqk), pv[j].w * (LOG2E * sqk));   // (NP = 2: the score accumulator holds s_q s_k S)
    wsq.st(sWqk, a.qkvw, HC, HC, oc_qk, sw);
    wsv.st(sWv, a.qkvw, HC, HC, oc_v, sw);
    wsp.st(sWp, a.projw, E, E, oc_p);
#pragma unroll
    for (int mt = 0; mt < MTQK; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) bqk[mt][v] *= syw;
    // HC = 8: the v tile has eight idle columns.  Columns 8 and 12 are ONES (zero weights, bias 1, left unscaled): rows 8 and 12 of O^T = the
    // softmax denominators of the tile's queries, on lane groups 2 and 3 -- one v_permlane32_swap brings them to groups 0 (head 0) and 1 (head 1)
    constexpr bool ONES = HC == 8;
#pragma unroll
    for (int nt = 0; nt < NTV; ++nt) { const int oc = G::v_oc(nt, c); bv[nt] = oc >= 0 ? bv[nt] * syw : ((c & 3) == 0 ? 1.0f : 0.f); }
#pragma unroll
    for (int i = 0; i < CHY; ++i) { gam[i] *= sy; bet[i] *= sy; }
    __syncthreads();
    AM_STAMP(0);

    const int nwx = a.w >> 3, nwy = a.h >> 3;
    const long hw = (long)a.h * a.w;
    const float qscale = (float)(1.0 / sqrt((double)D)) * LOG2E * (sq * inv_yw), kscale = sk * inv_yw;   // accumulator (s_y s_w q) -> operand s_q q D^-1/2 log2(e); (s_y s_w k) -> s_k k
    const int lpix = (c >> 3) * a.w + (c & 7);       // the lane's token inside a window, tile 0
    const int lx = lpix * E + 4 * g;                 // ... its first chunk in x / y (floats)
    const int tstep = 2 * a.w;                       // pixels per token tile

    // uneven (launcher: 512 resident workgroups, eight window quads per pair): the dispatcher places workgroups 0 .. 255 one per CU before the second 256 and the SIMD
    // arbiter issues the older wave first, so a CU's first workgroup runs faster than its second (k_ffn_xr.hip has the measurement): it takes 5 of the pair's 8 quads
    const int half = (int)gridDim.x >> 1, first = (int)blockIdx.x < half ? 1 : 0;
    const int nmine = uneven ? (first ? uneven : 8 - uneven) : 0x7fffffff;
    const int q0 = uneven ? (first ? (int)blockIdx.x : half * uneven + ((int)blockIdx.x - half)) : (int)blockIdx.x;
    const int qstep = uneven ? half : (int)gridDim.x;
    for (int quad = q0, kq = 0; quad < nquads && kq < nmine; quad += qstep, ++kq) {
        const int win = quad * 4 + __builtin_amdgcn_readfirstlane(wave);   // provably wave-uniform: window origins stay in scalar registers
        if (win >= nwin) continue;   // no barrier inside the loop
        const int wx = win % nwx, rr = win / nwx, wy = rr % nwy;
        const long b = rr / nwy;
        const long porg = (b * a.h + wy * 8) * a.w + wx * 8;          // first pixel of the window (uniform)
        const long pix0 = porg + lpix;                                // this lane's token of tile 0; tile t: + 2 t w
        const float* __restrict__ xw = a.x + porg * E;                // uniform bases; lane offsets are 32-bit
        float* __restrict__ yw = a.y + porg * E;
        const float* __restrict__ o2w = a.o2 + b * HC * hw + (porg - b * hw);

        // ---- x, LayerNorm, y1 fragments, to_qkv
        float4 xv[4][NCH];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < NCH; ++m) xv[t][m] = *reinterpret_cast<const float4*>(xw + (t * tstep * E + 16 * m + lx));

        AM_STAMP(1);   // window addresses + x loads issued
        float qk[4][4 * MTQK];                   // the lane's q / k channels of token 16 t + c: idx = ((is_q 2 + head) DG + dd)
        float vv[4][NTV][4];                     // V[token 16 t + 4 g + v][channel (nt, c)]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < NCH; ++m) s += (xv[t][m].x + xv[t][m].y) + (xv[t][m].z + xv[t][m].w);
            const float mu = xg_sum(s) * (1.0f / E);
            float q = 0.f;
#pragma unroll
            for (int m = 0; m < NCH; ++m) {
                const float d0 = xv[t][m].x - mu, d1 = xv[t][m].y - mu, d2 = xv[t][m].z - mu, d3 = xv[t][m].w - mu;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            const float rstd = __builtin_amdgcn_rsqf(xg_sum(q) * (1.0f / E) + LG_EPS);
            Pat<NPQ> yp[CHY];
#pragma unroll
            for (int m = 0; m < NY; ++m) {
                const float xs[4] = {xv[t][m].x, xv[t][m].y, xv[t][m].z, xv[t][m].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) yp[4 * m + i] = pat_of<NPQ>((xs[i] - mu) * rstd * gam[4 * m + i] + bet[4 * m + i]);
            }
            u32x4_t yf[NKQ];
#pragma unroll
            for (int k = 0; k < NKQ; ++k) yf[k] = build_frag<NPQ, CHY, false>(yp, k);
            // q / k: weights on the A side -> [channel row 4 g + v][token c]
#pragma unroll
            for (int mt = 0; mt < MTQK; ++mt) {
                f32x4_t acc = {bqk[mt][0], bqk[mt][1], bqk[mt][2], bqk[mt][3]};
#pragma unroll
                for (int k = 0; k < NKQ; ++k) acc = mfma_n<NPQ>(sWqk[(mt * NKQ + k) * 64 + lane], yf[k], acc);
#pragma unroll
                for (int v = 0; v < 4; ++v) qk[t][4 * mt + v] = acc[v];
            }
            // v: the same y1 fragment on the A side -> [token row 4 g + v][channel c]
#pragma unroll
            for (int nt = 0; nt < NTV; ++nt) {
                f32x4_t acc = {bv[nt], bv[nt], bv[nt], bv[nt]};
#pragma unroll
                for (int k = 0; k < NKQ; ++k) acc = mfma_n<NPQ>(yf[k], sWv[(nt * NKQ + k) * 64 + lane], acc);
#pragma unroll
                for (int v = 0; v < 4; ++v) vv[t][nt][v] = acc[v];
            }
        }

        AM_STAMP(2);   // wait for x, LayerNorm, to_qkv
        // ---- cat(o1, o2) of the lane: the FFT-mixer chunks (channels >= HC) are requested now, the o1 chunks are filled per head below
        float cat[4][NCH][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < NCH; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ch = 16 * m + 4 * g + i - HC;   // channel of o2 (negative: an o1 chunk)
                    cat[t][m][i] = 0.f;
                    if (16 * m + 12 + 3 >= HC) {              // the chunk can be an o2 chunk for some lane group
                        if (ch >= 0) cat[t][m][i] = o2w[ch * (int)hw + t * tstep + lpix];
                    }
                }

        // ---- V^T fragments: f16 pieces of v 2^sh, the power of two that puts the window's largest |v| into [2^14, 2^15)
        float vmax = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < NTV; ++nt)
#pragma unroll
                for (int v = 0; v < 4; ++v) vmax = fmaxf(vmax, fabsf(vv[t][nt][v]));
        if (ONES && c >= 8) vmax = 0.f;
        vmax = wave_max(vmax);
        const int sh = 15 - __builtin_amdgcn_frexp_expf(vmax);   // vmax = f 2^e, f in [0.5, 1): |v| 2^sh < 2^15
        const int shl = (ONES && c >= 8) ? 0 : sh;
        u32x4_t Vf[NTV][2][NPV];                                 // [column tile][k step][piece]; k-slot j of step s2: token tile 2 s2 + (j >> 2), row 4 g + (j & 3)
#pragma unroll
        for (int nt = 0; nt < NTV; ++nt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float w8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) w8[j] = __builtin_amdgcn_ldexpf(vv[2 * s2 + (j >> 2)][nt][j & 3], shl);
                uint32_t hi[4], lo[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hi[i] = cvt_f16x2(w8[2 * i], w8[2 * i + 1]);
                    if (NPV == 2) lo[i] = cvt_f16x2(res_lo(hi[i], w8[2 * i]), res_hi(hi[i], w8[2 * i + 1]));
                }
                Vf[nt][s2][0] = (u32x4_t){hi[0], hi[1], hi[2], hi[3]};
                if (NPV == 2) Vf[nt][s2][1] = (u32x4_t){lo[0], lo[1], lo[2], lo[3]};
            }

        AM_STAMP(3);   // o2 requests, V^T fragments
        // ---- per head: Q K^T operand fragments from the lane's own q / k channels, then per query tile scores, softmax, P V
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4_t Kf[4][NKS], Qf[4][NKS];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                Pat<NPQ> kp[DG], qp[DG];
#pragma unroll
                for (int dd = 0; dd < DG; ++dd) {
                    kp[dd] = pat_of<NPQ>(NP == 2 ? qk[t][h * DG + dd] * kscale : qk[t][h * DG + dd]);
                    qp[dd] = pat_of<NPQ>(qk[t][(2 + h) * DG + dd] * qscale);
                }
#pragma unroll
                for (int k = 0; k < NKS; ++k) {
                    Kf[t][k] = build_frag<NPQ, DG, true>(kp, k);
                    Qf[t][k] = build_frag<NPQ, DG, false>(qp, k);
                }
            }
            // rows 4 g + v of O^T belong to this head on these lanes (HC = 8: channel chunk g of head g; HC = 16: chunks 0, 1 | 2, 3; HC = 32: column tile h)
            const bool mine = HC >= 32 ? true : (HC == 8 ? (g == h) : ((g >> 1) == h));
            const int mo = HC >= 32 ? h : 0;    // cat chunk that receives them
            const int nt = HC >= 32 ? h : 0;
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                f32x4_t S[4];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const float4 p4 = sPos[((h * 4 + qt) * 4 + kt) * 64 + lane];
                    S[kt] = (f32x4_t){p4.x, p4.y, p4.z, p4.w};
#pragma unroll
                    for (int k = 0; k < NKS; ++k) S[kt] = mfma_n<NPQ>(Kf[kt][k], Qf[qt][k], S[kt]);
                }
                float mx = vmax3(vmax3(S[0][0], S[0][1], S[0][2]), S[0][3], S[1][0]);
                mx = vmax3(vmax3(mx, S[1][1], S[1][2]), S[1][3], S[2][0]);
                mx = vmax3(vmax3(mx, S[2][1], S[2][2]), S[2][3], S[3][0]);
                mx = vmax2(vmax3(mx, S[3][1], S[3][2]), S[3][3]);
                const float c0 = __builtin_fmaf(-xg_max(mx), inv_qk, 11.0f);   // p = 2^(s - max + 11) in (0, 2^11]: f16's normal range also holds the low piece of the large ones
                                                                                // (the accumulators hold s_q s_k s: one fma takes the scale out and the maximum off)
                float l = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int v = 0; v < 4; ++v) { S[kt][v] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kt][v], inv_qk, c0)); if (!ONES) l += S[kt][v]; }
                if (!ONES) l = xg_sum(l);
                // O^T[channel][query] += V^T[channel][key] P^T[key][query]
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    uint32_t hi[4], lo[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float p0 = S[2 * s2 + (i >> 1)][2 * (i & 1)], p1 = S[2 * s2 + (i >> 1)][2 * (i & 1) + 1];
                        hi[i] = cvt_f16x2(p0, p1);
                        if (NPV == 2) lo[i] = cvt_f16x2(res_lo(hi[i], p0), res_hi(hi[i], p1));
                    }
                    const u32x4_t ph = {hi[0], hi[1], hi[2], hi[3]};
                    if (NPV == 2) {
                        const u32x4_t pl = {lo[0], lo[1], lo[2], lo[3]};
                        acc = mfma_h(Vf[nt][s2][1], ph, acc);
                        acc = mfma_h(Vf[nt][s2][0], pl, acc);
                    }
                    acc = mfma_h(Vf[nt][s2][0], ph, acc);
                }
                if (ONES) {   // lanes 0 .. 31 receive acc[0] of lanes 32 .. 63: group 0 <- row 8, group 1 <- row 12
                    const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0]), __float_as_uint(acc[0]), false, false);
                    l = __uint_as_float(r.y);
                }
                const float f = __builtin_amdgcn_ldexpf(__builtin_amdgcn_rcpf(l), -sh) * inv_yw;     // (v carries s_y s_w)
#pragma unroll
                for (int v = 0; v < 4; ++v) cat[qt][mo][v] = mine ? acc[v] * f : cat[qt][mo][v];
                if (a.save_o && mine) {   // saving launch: what the backward would otherwise re-derive with a reduction pass of its own.  p = 2^(s - L) with
                                          // L = log2(sum_j 2^s_j) = log2(l) - c0 (l = sum of the 2^(s + c0)); this lane holds channels cb .. cb + 3 of query c
                    const long pix = pix0 + qt * tstep;
                    const int cb = (HC >= 32 ? 16 * h : 0) + 4 * g;
                    *reinterpret_cast<float4*>(a.save_o + pix * HC + cb) = make_float4(acc[0] * f, acc[1] * f, acc[2] * f, acc[3] * f);
                    if (HC >= 32 ? g == 0 : (HC == 16 ? (g & 1) == 0 : true)) a.save_l[pix * 2 + h] = __builtin_amdgcn_logf(l) - c0;
                }
            }
        }

        AM_STAMP(4);   // scores, softmax, P V
        // ---- proj -> dropout -> + x
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            const long pix = pix0 + 2 * qt * a.w;
            Pat<NPP> cp[CHC];
#pragma unroll
            for (int m = 0; m < NCH; ++m)
#pragma unroll
                for (int v = 0; v < 4; ++v) cp[4 * m + v] = pat_of<NPP>(cat[qt][m][v]);
            u32x4_t cf[NKP];
#pragma unroll
            for (int k = 0; k < NKP; ++k) cf[k] = build_frag<NPP, CHC, false>(cp, k);
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt) {
                f32x4_t acc = {bp[mt][0], bp[mt][1], bp[mt][2], bp[mt][3]};
#pragma unroll
                for (int k = 0; k < NKP; ++k) acc = mfma_bf(sWp[(mt * NKP + k) * 64 + lane], cf[k], acc);
                float o[4];
                const uint64_t di = (uint64_t)(pix * E + 16 * mt + 4 * g);   // a multiple of 4: | v below never carries
#pragma unroll
                for (int v = 0; v < 4; ++v) o[v] = acc[v];
                if (a.dropout) {   // two hashes for the lane's four consecutive channels
                    float s0, s1, s2, s3;
                    dropout_scale2(a.seed, di, s0, s1);
                    dropout_scale2(a.seed, di + 2, s2, s3);
                    o[0] *= s0; o[1] *= s1; o[2] *= s2; o[3] *= s3;
                }
                const float4 xr = RELOADX ? *reinterpret_cast<const float4*>(xw + (qt * tstep * E + 16 * mt + lx)) : xv[qt][mt];
                *reinterpret_cast<float4*>(yw + (qt * tstep * E + 16 * mt + lx)) = make_float4(xr.x + o[0], xr.y + o[1], xr.z + o[2], xr.w + o[3]);
            }
        }
        AM_STAMP(5);   // proj, dropout, residual, stores issued
    }
#ifdef LG_ATTN_STAMPS
    if (lane == 0 && blockIdx.x < 1024) {
        for (int i = 0; i < 6; ++i) am_stamps[blockIdx.x][wave][i] = st[i];
        am_stamps[blockIdx.x][wave][7] = 1ull;
    }
#endif
}

template <int HC, int NP>
static int launch_attn_m_t(const AttnArgs& a, hipStream_t s) {
    using G = am::Geo<HC>;
    constexpr int NPQ = NP == 2 ? 2 : NP, NPP = NP == 2 ? 3 : NP;
    constexpr int NKQ = am::cdiv(am::nprod(NPQ) * 4 * G::NY, 8), NKP = am::cdiv(am::nprod(NPP) * 4 * G::NCH, 8);
    const int nwin = a.B * (a.h / 8) * (a.w / 8);
    const int nquads = (nwin + 3) / 4;
    const size_t lds = (size_t)(2 * 4 * 4 * 64 + (G::MTQK + G::NTV) * NKQ * 64 + G::NCH * NKP * 64) * 16;
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_m<HC, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { lg_set_error("attn_m: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    // persistent grid: the workgroups that are RESIDENT at once (registers and LDS: asked of the runtime once per device), each walking its
    // window quads with pos_emb and the weight fragments in LDS.  A larger grid runs in rounds -- 683 workgroups on 512 slots: two stagings
    // and 2 x 3 windows per slot instead of one and 4.
    static std::atomic<int> per_cu_cache[64];
    int per_cu = per_cu_cache[DeviceOnce::dev()].load(std::memory_order_acquire);
    if (per_cu <= 0) {
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_attn_m<HC, NP>, 256, lds);
        if (e != hipSuccess || nb < 1) { lg_set_error("attn_m: occupancy query: %s (%d)", hipGetErrorString(e), nb); return e != hipSuccess ? (int)e : -3; }
        per_cu = nb;
        per_cu_cache[DeviceOnce::dev()].store(per_cu, std::memory_order_release);
    }
    int ncu = 256;
    const int cap = ncu * per_cu;
    const int rounds = (nquads + cap - 1) / cap;
    const int grid = nquads < cap ? nquads : (nquads + rounds - 1) / rounds;
#ifndef LG_ATTN_UNEVEN
#define LG_ATTN_UNEVEN 5
#endif
    const int uneven = (per_cu == 2 && grid == 512 && nquads == 4 * grid && nwin == 4 * nquads) ? LG_ATTN_UNEVEN : 0;   // the measured shape only
    k_attn_m<HC, NP><<<grid, 256, lds, s>>>(a, nwin, nquads, uneven == 4 ? 0 : uneven);
    LG_CHECK_LAUNCH();
    return 0;
}

int launch_attn_m(int e, const AttnArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_ATTN, s);
    if ((a.h & 7) || (a.w & 7)) { lg_set_error("attn: h,w must be multiples of 8"); return -2; }
    if (a.bf16) {
        if (e == 16) return launch_attn_m_t<8, 1>(a, s);
        if (e == 32) return launch_attn_m_t<16, 1>(a, s);
        if (e == 64) return launch_attn_m_t<32, 1>(a, s);
    } else if (a.scales) {   // round 6: to_qkv and Q K^T on f16 pairs under the block's static scales
        if (e == 16) return launch_attn_m_t<8, 2>(a, s);
        if (e == 32) return launch_attn_m_t<16, 2>(a, s);
        if (e == 64) return launch_attn_m_t<32, 2>(a, s);
    } else {
        if (e == 16) return launch_attn_m_t<8, 3>(a, s);
        if (e == 32) return launch_attn_m_t<16, 3>(a, s);
        if (e == 64) return launch_attn_m_t<32, 3>(a, s);
    }
    lg_set_error("attn: e=%d unsupported", e);
    return -1;
}
