// Backward of the local (window) mixer on the gfx950 MATRIX pipe (round 5) -- autograd of reference models/common/LGT.py:112-146
// (local_mixer) for the widths whose half-block backward is three kernels (e = 32: level 1 of the 4-band net, level 0 of the 8-band net).
// Drop-in for k_attn_bwd_core (k_attn_bwd.hip): same arguments, same outputs (dq | dk | dv rows of a.dqkv, the attention output
// into a.cat, y1, the pos_emb gradient slab), same launch shape (one wave = one window and ONE head, blockIdx.y = head), same prologue.
//
// Why: at D = 8 the vector-pipe core streams every K / V / Q / dO element of the window through LDS once per query (lane = query) and
// once per key (lane = key): 1 280 broadcast ds_read_b64 per lane and (window, head).  The LDS return path, not the ALUs, set its pace
// (238 us per launch at configs[2]: 42 % VALU-active).  Here the five products are MFMAs whose operands are read from the same LDS
// tiles ONCE as fragments:
//
//   orientation A (rows = keys, columns = queries; the forward's lane map):  S^T = K Q^T + pos,  dP^T = V dO^T,  softmax and
//       D_i = sum_j P_ij dP_ij down the columns (16 in-lane values + two v_permlane*_swap steps), dS = P (dP - D), the pos_emb
//       gradient (64 accumulator registers per lane, across all windows of the wave), and the two products that contract over KEYS:
//       O^T = V^T P^T (the proj weight gradient's operand) and dQ^T = K^T dS^T -- P^T / dS^T go from the accumulator registers
//       straight into the B operand;
//   orientation B (rows = queries, columns = keys):  S = Q K^T + pos, dP = dO V^T again, P and dS rebuilt from the row statistics
//       that orientation A left in LDS (max, 1 / sum, D_i), and the two products that contract over QUERIES: dV^T = dO^T P, dK^T = Q^T dS.
//
// Arithmetic: f16 PAIRS (split_bf16.h NP = 2: hi + lo, three piece products, fp32 accumulation).  q, k, v, dO are scaled per (window,
// head) by the power of two that puts the tile's largest magnitude into [2^14, 2^15) (four wave-wide max reductions in the prologue);
// P by 2^14; dS by the power of two that follows from the bound |dS| <= 2 D max|v| max|dO|.  Every accumulator is a fixed power of two
// times its true value and is rescaled where it leaves the registers.
#include "kernels.h"
#include "bwd_kernels.h"
#include "split_bf16.h"

namespace abm {
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
__device__ __forceinline__ float wave_max(float v) {
    v = xg_max(v);
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) v = vmax2(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ f32x4_t mfma_h(u32x4_t a, u32x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// f16 pair of two scaled values as (hi dword, lo dword)
__device__ __forceinline__ void pair2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = sb_cvt_f16x2(a, b);
    lo = sb_cvt_f16x2(sb_res_lo(hi, a), sb_res_hi(hi, b));
}
// "row" fragment of a [D = 8][64 tokens] fp32 LDS tile for token tile t: lane (g, c) holds channels 2 g, 2 g + 1 of token 16 t + c.  A-side
// order of the three piece products {lo, hi, hi}; the B-side order {hi, lo, hi} is bside() of it (a dword permutation).
__device__ __forceinline__ u32x4_t row_frag(const float* tile, int t, int g, int c, int sh) {
    const float v0 = __builtin_amdgcn_ldexpf(tile[(2 * g) * 64 + 16 * t + c], sh), v1 = __builtin_amdgcn_ldexpf(tile[(2 * g + 1) * 64 + 16 * t + c], sh);
    uint32_t hi, lo;
    pair2(v0, v1, hi, lo);
    return (u32x4_t){lo, hi, hi, 0u};
}
__device__ __forceinline__ u32x4_t bside(u32x4_t f) { return (u32x4_t){f.y, f.x, f.z, 0u}; }
// "column" fragment (A operand whose rows are the CHANNELS, k = tokens): lane (g, c) supplies channel c & 7 (rows 8 .. 15 repeat rows 0 .. 7: their
// results are never read), k-slot j of step s2 = token 16 (2 s2 + (j >> 2)) + 4 g + (j & 3) -- the row order of the accumulator tiles that form the B operand
__device__ __forceinline__ void col_frag(const float* tile, int s2, int g, int c, int sh, u32x4_t& hi4, u32x4_t& lo4) {
    const float* p = tile + (c & 7) * 64 + 32 * s2 + 4 * g;
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 16);
    const float w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pair2(__builtin_amdgcn_ldexpf(w[2 * i], sh), __builtin_amdgcn_ldexpf(w[2 * i + 1], sh), hi[i], lo[i]);
    hi4 = (u32x4_t){hi[0], hi[1], hi[2], hi[3]};
    lo4 = (u32x4_t){lo[0], lo[1], lo[2], lo[3]};
}
// B operand of a contraction over the ROWS of two accumulator tiles (tile 2 s2, tile 2 s2 + 1): dword i = values (2 (i & 1), 2 (i & 1) + 1) of tile i >> 1
__device__ __forceinline__ void tile_frag(const f32x4_t& t0, const f32x4_t& t1, u32x4_t& hi4, u32x4_t& lo4) {
    uint32_t hi[4], lo[4];
    pair2(t0[0], t0[1], hi[0], lo[0]);
    pair2(t0[2], t0[3], hi[1], lo[1]);
    pair2(t1[0], t1[1], hi[2], lo[2]);
    pair2(t1[2], t1[3], hi[3], lo[3]);
    hi4 = (u32x4_t){hi[0], hi[1], hi[2], hi[3]};
    lo4 = (u32x4_t){lo[0], lo[1], lo[2], lo[3]};
}
// acc += A (hi, lo) * B (hi, lo): the three piece products, small terms first
__device__ __forceinline__ f32x4_t mfma3(u32x4_t ah, u32x4_t al, u32x4_t bh, u32x4_t bl, f32x4_t acc) {
    acc = mfma_h(al, bh, acc);
    acc = mfma_h(ah, bl, acc);
    return mfma_h(ah, bh, acc);
}
}  // namespace abm

// PH = 0: prologue + orientation A (O, dQ, the pos_emb gradient; the row statistics go to a.stats);  PH = 1: prologue + orientation B (dV, dK) from the
// statistics.  Two launches because of the register file: the 64 pos_emb accumulators of a lane live across all windows of its wave, and beside them one
// orientation fits 256 registers, both do not (330) -- the second launch has no accumulators and pays the prologue again (cheap on
// the matrix pipe) plus 32 bytes of statistics per pixel.
template <int HC, int NW, int PH>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2))) void k_attn_bwd_core_m(AttnBwdArgs a, int nwin, int ngroups) {
    using namespace abm;
    constexpr int E = 2 * HC, D = HC / 2;
    static_assert(D == 8, "fragment builders are written for a head dimension of 8 (e = 32)");
    constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
    constexpr int Y1LD = (HC + 15) / 16 * 16, DQLD = (3 * HC + 15) / 16 * 16;
    constexpr int PW = 4 * 64 * D + 64 * 4;       // floats of LDS per wave: K | V | Q | dO tiles [D][64] + row statistics [3][64] (+ pad)
    constexpr int PLD = 65;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int hd = blockIdx.y;
    float4* sPosA = reinterpret_cast<float4*>(smem);              // [4 qt][4 kt][64 lanes]: pos[i = 16 qt + c][j = 16 kt + 4 g + v] log2(e)
    float* sDpos = smem;                                          // [64][65] at the very end (the pos table is dead by then; its region is 4 192 floats)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    float* sK = smem + 4192 + wave * PW;   // [D][64]
    float* sV = sK + 64 * D;
    float* sQ = sV + 64 * D;               // q * scale * log2(e)
    float* sDO = sQ + 64 * D;
    float* sSt = sDO + 64 * D;             // [3][64]: row max (log2 domain), 2^14 / row sum, D_i 2^(sh_ds - 14)

    {   // pos_emb of this head in the two fragment orders
        const float* ph = a.pos + hd * 64 * 64;
        for (int u = threadIdx.x; u < 4 * 4 * 64; u += NW * 64) {
            const int ln = u & 63, kt = (u >> 6) & 3, qt = u >> 8, lg = ln >> 4, lc = ln & 15;
            const float4 pa = *reinterpret_cast<const float4*>(ph + (16 * qt + lc) * 64 + 16 * kt + 4 * lg);
            sPosA[u] = make_float4(pa.x * LOG2E, pa.y * LOG2E, pa.z * LOG2E, pa.w * LOG2E);
        }
    }
    // ---- the prologue's weights as A-operand fragments (f16 pairs of W 2^sw), once per workgroup: [fragment][lane] 16-byte units.
    //   fragments 0, 1: rows 0 .. 7 = this head's q channels, rows 8 .. 15 = its k channels;  2, 3: rows 0 .. 7 = its v channels (rows 8 .. 15 zero);
    //   K = the 16 local channels: lane group g holds channels 4 g .. 4 g + 3, k-step 0 = {lo(0,1), lo(2,3), hi(0,1), hi(2,3)}, k-step 1 = {hi(0,1), hi(2,3), 0, 0}
    //   (A-side piece order lo, hi, hi against the activation's hi, lo, hi);
    //   fragments 4, 5, 6: rows 0 .. 7 = proj^T rows of this head's channels (K = the e block channels: chunks 4 g .. of both 16-channel halves,
    //   24 slots: {lo(m0: 0,1), lo(m0: 2,3), lo(m1: 0,1), lo(m1: 2,3)}, {hi ...}, {hi ...}).
    __shared__ __attribute__((aligned(16))) u32x4_t sWf[7 * 64];
    __shared__ float sRed[NW * 64];
    float wq_max = 0.f, wp_max = 0.f;
    for (int i = threadIdx.x; i < 3 * D * HC; i += NW * 64) {
        const int t3 = i / (D * HC), rem = i - t3 * (D * HC);
        wq_max = fmaxf(wq_max, fabsf(a.qkvw[(size_t)(t3 * HC + hd * D) * HC + rem]));
    }
    for (int i = threadIdx.x; i < D * E; i += NW * 64) wp_max = fmaxf(wp_max, fabsf(a.projw[(i % E) * E + hd * D + i / E]));
    wq_max = wave_max(wq_max); wp_max = wave_max(wp_max);
    if (lane == 0) { sRed[wave] = wq_max; sRed[NW + wave] = wp_max; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) { wq_max = fmaxf(wq_max, sRed[w]); wp_max = fmaxf(wp_max, sRed[NW + w]); }
    const int sw_q = 15 - __builtin_amdgcn_frexp_expf(wq_max), sw_p = 15 - __builtin_amdgcn_frexp_expf(wp_max);
    if (threadIdx.x < 64) {   // one wave builds the seven fragments of every lane (g, r)
        const int r = c;
        // to_qkv rows: tile 0 row r: q channel r (r < 8) | k channel r - 8;  tile 1 row r: v channel r (r < 8) | zero
        for (int tile = 0; tile < 2; ++tile) {
            const bool ok = tile == 0 || r < 8;
            const int oc = tile == 0 ? (r < 8 ? hd * D + r : HC + hd * D + (r - 8)) : 2 * HC + hd * D + (r & 7);
            const float4 w4 = *reinterpret_cast<const float4*>(a.qkvw + (size_t)oc * HC + 4 * g);
            uint32_t h01, l01, h23, l23;
            pair2(__builtin_amdgcn_ldexpf(w4.x, sw_q), __builtin_amdgcn_ldexpf(w4.y, sw_q), h01, l01);
            pair2(__builtin_amdgcn_ldexpf(w4.z, sw_q), __builtin_amdgcn_ldexpf(w4.w, sw_q), h23, l23);
            sWf[(2 * tile) * 64 + lane] = ok ? (u32x4_t){l01, l23, h01, h23} : (u32x4_t){0u, 0u, 0u, 0u};
            sWf[(2 * tile + 1) * 64 + lane] = ok ? (u32x4_t){h01, h23, 0u, 0u} : (u32x4_t){0u, 0u, 0u, 0u};
        }
        {   // proj^T: row r (< 8) = projw[n][hd D + r] over n = 16 m + 4 g + i
            float wv[8];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) wv[4 * m + i] = __builtin_amdgcn_ldexpf(a.projw[(size_t)(16 * m + 4 * g + i) * E + hd * D + (r & 7)], sw_p);
            uint32_t hh[4], ll[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pair2(wv[2 * i], wv[2 * i + 1], hh[i], ll[i]);
            const bool ok = r < 8;
            const u32x4_t zz = {0u, 0u, 0u, 0u};
            sWf[4 * 64 + lane] = ok ? (u32x4_t){ll[0], ll[1], ll[2], ll[3]} : zz;
            sWf[5 * 64 + lane] = ok ? (u32x4_t){hh[0], hh[1], hh[2], hh[3]} : zz;
            sWf[6 * 64 + lane] = ok ? (u32x4_t){hh[0], hh[1], hh[2], hh[3]} : zz;
        }
    }
    // lane constants: LayerNorm affine of the lane's local-half channels 4 g .. 4 g + 3, biases of the accumulator rows 4 g + v
    float gam[4], bet[4], bqk[4], bvv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { gam[i] = a.ln1g[4 * g + i]; bet[i] = a.ln1b[4 * g + i]; }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int rr = 4 * g + v;
        bqk[v] = a.qkvb[rr < 8 ? hd * D + rr : HC + hd * D + (rr - 8)];
        bvv[v] = a.qkvb[2 * HC + hd * D + (rr & 7)];
    }
    const int nwx = a.w >> 3, nwy = a.h >> 3;
    const long hw = (long)a.h * a.w;
    const float scale = (float)(1.0 / sqrt((double)D));
    // pos_emb gradient [qt][kt]: rows = keys 16 kt + 4 g + v, column = query 16 qt + c; all windows of this wave.  (ds_add_f32 into one LDS image per
    // workgroup instead: 434 us per launch against 217; the accumulators as they are needed every trick below to fit 256 registers.)
    f32x4_t dpa[PH == 0 ? 4 : 1][4];
#pragma unroll
    for (int qt = 0; qt < (PH == 0 ? 4 : 1); ++qt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) dpa[qt][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int win = grp * NW + wave;
        if (win >= nwin) continue;          // (no workgroup barrier inside the loop)
        const int wx = win % nwx, rr = win / nwx, wy = rr % nwy;
        const long b = rr / nwy;
        const long porg = (b * a.h + wy * 8) * a.w + wx * 8;       // first pixel of the window
        __builtin_amdgcn_wave_barrier();
        float qm = 0.f, km = 0.f, vm = 0.f, dm = 0.f;
        {   // ---------------- prologue on the matrix pipe, in the forward's lane map: lane (g, c) holds of token 16 t + c the 16-byte chunks {4 m + g}
            // of x and dym.  LayerNorm -> y1 (the lane's four local channels) -> B fragments; q | k and v of this head as W y1 with the weights on
            // the A side (accumulator rows 4 g + v = channels, column c = token): straight into the [channel][token] tiles the passes read;
            // dO = (proj^T dym)[head] the same way.  (The vector-pipe prologue of k_attn_bwd_core -- lane = token, every weight a broadcast LDS
            // read -- was 91 of this kernel's first 217 us.)
            const int tstep = 2 * a.w, lpix = (c >> 3) * a.w + (c & 7);
            const float* __restrict__ xw = a.x + porg * E;
            const float* __restrict__ dw = a.dym + porg * E;
            float4 xv[4][2], dv[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m) xv[t][m] = *reinterpret_cast<const float4*>(xw + ((t * tstep + lpix) * E + 16 * m + 4 * g));
            if (PH == 0 && hd == 0) {   // the FFT-mixer half of the proj input: planar o2 -> cat[..][HC + .] (lane = token here)
                const long p = porg + (lane >> 3) * a.w + (lane & 7);
                const long sp = p - b * hw;
                float o2v[HC];
#pragma unroll
                for (int cc = 0; cc < HC; ++cc) o2v[cc] = a.o2[(b * HC + cc) * hw + sp];
                float4* co = reinterpret_cast<float4*>(a.cat + p * E + HC);
#pragma unroll
                for (int k = 0; k < HC / 4; ++k) co[k] = make_float4(o2v[4 * k], o2v[4 * k + 1], o2v[4 * k + 2], o2v[4 * k + 3]);
            }
            // dym: the window's largest magnitude -> scale; y1: likewise after the LayerNorm
            float y1[4][4];
            float ym = 0.f, gm = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float sx = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m) sx += (xv[t][m].x + xv[t][m].y) + (xv[t][m].z + xv[t][m].w);
                const float mu = xg_sum(sx) * (1.0f / E);
                float qv = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const float d0 = xv[t][m].x - mu, d1 = xv[t][m].y - mu, d2 = xv[t][m].z - mu, d3 = xv[t][m].w - mu;
                    qv += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
                const float rstd = __builtin_amdgcn_rsqf(xg_sum(qv) * (1.0f / E) + LG_EPS);
                const float xs[4] = {xv[t][0].x, xv[t][0].y, xv[t][0].z, xv[t][0].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) { y1[t][i] = (xs[i] - mu) * rstd * gam[i] + bet[i]; ym = fmaxf(ym, fabsf(y1[t][i])); }
                if (PH == 0 && hd == 0 && a.y1) *reinterpret_cast<float4*>(a.y1 + (porg + t * tstep + lpix) * Y1LD + 4 * g) = make_float4(y1[t][0], y1[t][1], y1[t][2], y1[t][3]);
            }
            // dym is requested only now (x is dead): both at once do not fit beside the 64 pos_emb accumulators
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    dv[t][m] = *reinterpret_cast<const float4*>(dw + ((t * tstep + lpix) * E + 16 * m + 4 * g));
                    gm = fmaxf(gm, fmaxf(fmaxf(fabsf(dv[t][m].x), fabsf(dv[t][m].y)), fmaxf(fabsf(dv[t][m].z), fabsf(dv[t][m].w))));
                }
            ym = wave_max(ym); gm = wave_max(gm);
            const int sy = 15 - __builtin_amdgcn_frexp_expf(ym), sg = 15 - __builtin_amdgcn_frexp_expf(gm);
            const float iq = __builtin_amdgcn_ldexpf(1.0f, -(sy + sw_q)), id = __builtin_amdgcn_ldexpf(1.0f, -(sg + sw_p));
            const u32x4_t wq0 = sWf[lane], wq1 = sWf[64 + lane], wv0 = sWf[128 + lane], wv1 = sWf[192 + lane];
            const u32x4_t wp0 = sWf[256 + lane], wp1 = sWf[320 + lane], wp2 = sWf[384 + lane];
            const f32x4_t z4p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint32_t h01, l01, h23, l23;
                pair2(__builtin_amdgcn_ldexpf(y1[t][0], sy), __builtin_amdgcn_ldexpf(y1[t][1], sy), h01, l01);
                pair2(__builtin_amdgcn_ldexpf(y1[t][2], sy), __builtin_amdgcn_ldexpf(y1[t][3], sy), h23, l23);
                const u32x4_t y0 = {h01, h23, l01, l23}, y1f = {h01, h23, 0u, 0u};         // B side: hi | lo, then hi
                f32x4_t aqk = mfma_h(wq0, y0, z4p), av = mfma_h(wv0, y0, z4p);
                aqk = mfma_h(wq1, y1f, aqk); av = mfma_h(wv1, y1f, av);
                const float dd[8] = {dv[t][0].x, dv[t][0].y, dv[t][0].z, dv[t][0].w, dv[t][1].x, dv[t][1].y, dv[t][1].z, dv[t][1].w};
                uint32_t dh[4], dl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) pair2(__builtin_amdgcn_ldexpf(dd[2 * i], sg), __builtin_amdgcn_ldexpf(dd[2 * i + 1], sg), dh[i], dl[i]);
                const u32x4_t dhi = {dh[0], dh[1], dh[2], dh[3]}, dlo = {dl[0], dl[1], dl[2], dl[3]};
                f32x4_t ado = mfma_h(wp0, dhi, z4p);      // lo hi
                ado = mfma_h(wp1, dlo, ado);              // hi lo
                ado = mfma_h(wp2, dhi, ado);              // hi hi
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int rr = 4 * g + v;             // accumulator row
                    const float qk = fmaf(aqk[v], iq, bqk[v]);
                    const float vvv = fmaf(av[v], iq, bvv[v]);
                    const float dov = ado[v] * id;
                    if (g < 2) {   // rows 0 .. 7
                        const float q = qk * (scale * LOG2E);
                        sQ[rr * 64 + 16 * t + c] = q; sV[rr * 64 + 16 * t + c] = vvv; sDO[rr * 64 + 16 * t + c] = dov;
                        qm = fmaxf(qm, fabsf(q)); vm = fmaxf(vm, fabsf(vvv)); dm = fmaxf(dm, fabsf(dov));
                    } else {       // rows 8 .. 15: k
                        sK[(rr - 8) * 64 + 16 * t + c] = qk;
                        km = fmaxf(km, fabsf(qk));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);     // one token tile at a time (registers)
            }
        }
        qm = wave_max(qm); km = wave_max(km); vm = wave_max(vm); dm = wave_max(dm);
        // operand scales of this (window, head): |t| 2^sh < 2^15
        const int sh_q = 15 - __builtin_amdgcn_frexp_expf(qm), sh_k = 15 - __builtin_amdgcn_frexp_expf(km);
        const int sh_v = 15 - __builtin_amdgcn_frexp_expf(vm), sh_d = 15 - __builtin_amdgcn_frexp_expf(dm);
        const int sh_s = 15 - __builtin_amdgcn_frexp_expf(2.0f * (float)D * vm * dm);      // |dS| <= 2 D max|v| max|dO|
        const float inv_qk = __builtin_amdgcn_ldexpf(1.0f, -(sh_q + sh_k)), c_dp = __builtin_amdgcn_ldexpf(1.0f, sh_s - 14 - (sh_v + sh_d));
        const float inv_s = __builtin_amdgcn_ldexpf(1.0f, -sh_s);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();    // the tiles are this wave's own
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

#ifndef ABM_SKIP_PASSES
        // (operand fragments are built from the LDS tiles where they are used: an MFMA operand is a tuple of four registers, {lo, hi, hi, 0} and its
        //  B-side permutation are different tuples -- 16 resident fragments were 64 registers)
        const int tok_off = (c >> 3) * a.w + (c & 7);            // pixel offset of token c of a tile (tile t: + 2 t w)
        const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};

        // ---------------- orientation A: rows = keys, columns = queries
        if constexpr (PH == 0) {
            u32x4_t fK[4], fV[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { fK[t] = row_frag(sK, t, g, c, sh_k); fV[t] = row_frag(sV, t, g, c, sh_v); }
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                f32x4_t S[4], P[4];
                const u32x4_t qb = bside(row_frag(sQ, qt, g, c, sh_q)), db = bside(row_frag(sDO, qt, g, c, sh_d));
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) { S[kt] = mfma_h(fK[kt], qb, z4); P[kt] = mfma_h(fV[kt], db, z4); }     // P: dP^T (scaled) for now
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const float4 p4 = sPosA[(qt * 4 + kt) * 64 + lane];
                    S[kt][0] = fmaf(S[kt][0], inv_qk, p4.x); S[kt][1] = fmaf(S[kt][1], inv_qk, p4.y);
                    S[kt][2] = fmaf(S[kt][2], inv_qk, p4.z); S[kt][3] = fmaf(S[kt][3], inv_qk, p4.w);
                }
                float mx = vmax3(vmax3(S[0][0], S[0][1], S[0][2]), S[0][3], S[1][0]);
                mx = vmax3(vmax3(mx, S[1][1], S[1][2]), S[1][3], S[2][0]);
                mx = vmax3(vmax3(mx, S[2][1], S[2][2]), S[2][3], S[3][0]);
                mx = vmax2(vmax3(mx, S[3][1], S[3][2]), S[3][3]);
                mx = xg_max(mx);
                float l = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int v = 0; v < 4; ++v) { S[kt][v] = __builtin_amdgcn_exp2f(S[kt][v] - mx); l += S[kt][v]; }
                l = xg_sum(l);
                const float il14 = __builtin_amdgcn_ldexpf(__builtin_amdgcn_rcpf(l), 14);
                float dsum = 0.f;      // sum_j P14_ij dP_ij (scaled by c_dp)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int v = 0; v < 4; ++v) { S[kt][v] *= il14; P[kt][v] *= c_dp; dsum = fmaf(S[kt][v], P[kt][v], dsum); }
                // D_i 2^(sh_s - 14) = 2^-14 sum_j P14 (dP c_dp)
                const float dq_s = __builtin_amdgcn_ldexpf(xg_sum(dsum), -14);
                if (g == 0)     // row statistics of query 16 qt + c for the second launch: max (log2 domain), 2^14 / sum, D_i 2^(sh_s - 14)
                    *reinterpret_cast<float4*>(a.stats + ((porg + 2 * qt * a.w + tok_off) * 2 + hd) * 4) = make_float4(mx, il14, dq_s, 0.f);
                // O^T = V^T P^T before P's registers become dS
                f32x4_t oacc = z4, qacc = z4;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4_t ph, pl, Vh, Vl;
                    tile_frag(S[2 * s2], S[2 * s2 + 1], ph, pl);
                    col_frag(sV, s2, g, c, sh_v, Vh, Vl);      // (rebuilt per query tile, as K's below: 16 registers each)
                    oacc = mfma3(Vh, Vl, ph, pl, oacc);
                }
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        P[kt][v] = S[kt][v] * (P[kt][v] - dq_s);               // dS 2^sh_s
                        dpa[qt][kt][v] = fmaf(P[kt][v], inv_s, dpa[qt][kt][v]);
                    }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4_t sh4, sl4, Kh, Kl;
                    tile_frag(P[2 * s2], P[2 * s2 + 1], sh4, sl4);
                    col_frag(sK, s2, g, c, sh_k, Kh, Kl);      // (rebuilt per query tile: 16 registers)
                    qacc = mfma3(Kh, Kl, sh4, sl4, qacc);
                }
                if (g < 2) {   // rows 4 g + v = head channels 0 .. 7 of query 16 qt + c
                    const long pix = porg + 2 * qt * a.w + tok_off;
                    const float fo = __builtin_amdgcn_ldexpf(1.0f, -(sh_v + 14)), fq = __builtin_amdgcn_ldexpf(scale, -(sh_k + sh_s));
                    *reinterpret_cast<float4*>(a.cat + pix * E + hd * D + 4 * g) = make_float4(oacc[0] * fo, oacc[1] * fo, oacc[2] * fo, oacc[3] * fo);
                    *reinterpret_cast<float4*>(a.dqkv + pix * DQLD + hd * D + 4 * g) = make_float4(qacc[0] * fq, qacc[1] * fq, qacc[2] * fq, qacc[3] * fq);
                }
                __builtin_amdgcn_sched_barrier(0);   // one query tile at a time: interleaved tiles need more registers than there are
            }
        }
        // ---------------- orientation B: rows = queries, columns = keys
#ifndef ABM_SKIP_B
        if constexpr (PH == 1) {
            {   // the statistics of this wave's 64 queries: lane = token, through the wave's LDS region
                const float4 st = *reinterpret_cast<const float4*>(a.stats + ((porg + (lane >> 3) * a.w + (lane & 7)) * 2 + hd) * 4);
                sSt[lane] = st.x; sSt[64 + lane] = st.y; sSt[128 + lane] = st.z;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            u32x4_t fQ[4], fD[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { fQ[t] = row_frag(sQ, t, g, c, sh_q); fD[t] = row_frag(sDO, t, g, c, sh_d); }
            u32x4_t Dh[2], Dl[2], Qh[2], Ql[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { col_frag(sDO, s2, g, c, sh_d, Dh[s2], Dl[s2]); col_frag(sQ, s2, g, c, sh_q, Qh[s2], Ql[s2]); }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                f32x4_t vacc = z4, kacc = z4;
                const u32x4_t kb = bside(row_frag(sK, kt, g, c, sh_k)), vb = bside(row_frag(sV, kt, g, c, sh_v));
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    f32x4_t S[2], P[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int qt = 2 * s2 + u;
                        S[u] = mfma_h(fQ[qt], kb, z4);
                        P[u] = mfma_h(fD[qt], vb, z4);
                        // pos[i = 16 qt + 4 g + v][j = 16 kt + c] out of the orientation-A table: element (qt, kt, lane (c >> 2, 4 g + v)), component c & 3 (all 64 lanes on different banks)
                        const float* pt = reinterpret_cast<const float*>(sPosA + (qt * 4 + kt) * 64) + 64 * (c >> 2) + 16 * g + (c & 3);
                        const float pv[4] = {pt[0], pt[4], pt[8], pt[12]};
                        // row statistics of queries 16 qt + 4 g .. + 3 (read per use: 48 registers otherwise)
                        const float4 rm = *reinterpret_cast<const float4*>(sSt + 16 * qt + 4 * g), ri = *reinterpret_cast<const float4*>(sSt + 64 + 16 * qt + 4 * g);
                        const float4 rd = *reinterpret_cast<const float4*>(sSt + 128 + 16 * qt + 4 * g);
                        const float m4[4] = {rm.x, rm.y, rm.z, rm.w}, i4[4] = {ri.x, ri.y, ri.z, ri.w}, d4[4] = {rd.x, rd.y, rd.z, rd.w};
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const float e = __builtin_amdgcn_exp2f(fmaf(S[u][v], inv_qk, pv[v]) - m4[v]);
                            S[u][v] = e * i4[v];                                   // P 2^14
                            P[u][v] = S[u][v] * fmaf(P[u][v], c_dp, -d4[v]);       // dS 2^sh_s
                        }
                    }
                    u32x4_t ph, pl, sh4, sl4;
                    tile_frag(S[0], S[1], ph, pl);
                    tile_frag(P[0], P[1], sh4, sl4);
                    vacc = mfma3(Dh[s2], Dl[s2], ph, pl, vacc);
                    kacc = mfma3(Qh[s2], Ql[s2], sh4, sl4, kacc);
                }
                if (g < 2) {   // rows = head channels of key 16 kt + c
                    const long pix = porg + 2 * kt * a.w + tok_off;
                    const float fv = __builtin_amdgcn_ldexpf(1.0f, -(sh_d + 14)), fk = __builtin_amdgcn_ldexpf(LN2, -(sh_q + sh_s));   // sQ carries log2(e)
                    *reinterpret_cast<float4*>(a.dqkv + pix * DQLD + HC + hd * D + 4 * g) = make_float4(kacc[0] * fk, kacc[1] * fk, kacc[2] * fk, kacc[3] * fk);
                    *reinterpret_cast<float4*>(a.dqkv + pix * DQLD + 2 * HC + hd * D + 4 * g) = make_float4(vacc[0] * fv, vacc[1] * fv, vacc[2] * fv, vacc[3] * fv);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
#endif
    }
    if constexpr (PH == 1) return;
    // ---------------- pos_emb gradient of this workgroup: the waves add their accumulators in turn (fixed order), then one slab row
    __syncthreads();     // every wave is done with the pos table
    for (int i = threadIdx.x; i < 64 * PLD; i += NW * 64) sDpos[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < NW; ++w) {
        if (wave == w) {
#pragma unroll
            for (int qt = 0; qt < 4; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int v = 0; v < 4; ++v) sDpos[(16 * qt + c) * PLD + 16 * kt + 4 * g + v] += dpa[PH == 0 ? qt : 0][kt][v];
        }
        __syncthreads();
    }
    float* slab = a.dpos_slab + ((size_t)blockIdx.x * 2 + hd) * 64 * 64;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += NW * 64) slab[idx] = sDpos[(idx >> 6) * PLD + (idx & 63)];
}

// launched by launch_attn_bwd_t (k_attn_bwd.hip) in place of k_attn_bwd_core<16, 4>
int launch_attn_bwd_core_m(int e, const AttnBwdArgs& a, int grid, int nwin, int ngroups, hipStream_t s) {
    if (e != 32) { lg_set_error("attn_bwd_core_m: e=%d unsupported", e); return -1; }
    constexpr int HC = 16, NW = 4;
    const size_t lds = (size_t)(4192 + NW * (4 * 64 * (HC / 2) + 64 * 4)) * sizeof(float);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t er = hipFuncSetAttribute((const void*)k_attn_bwd_core_m<HC, NW, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
        if (er == hipSuccess) er = hipFuncSetAttribute((const void*)k_attn_bwd_core_m<HC, NW, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
        if (er != hipSuccess) { lg_set_error("attn_bwd_core_m: hipFuncSetAttribute: %s", hipGetErrorString(er)); return (int)er; }
        attr_once.done();
    }
    if (!a.stats) { lg_set_error("attn_bwd_core_m: statistics scratch missing"); return -2; }
    k_attn_bwd_core_m<HC, NW, 0><<<dim3(grid, 2), NW * 64, lds, s>>>(a, nwin, ngroups);
    LG_CHECK_LAUNCH();
    k_attn_bwd_core_m<HC, NW, 1><<<dim3(grid, 2), NW * 64, lds, s>>>(a, nwin, ngroups);
    LG_CHECK_LAUNCH();
    return 0;
}
