// Backward of the local-mixer half-block in ONE kernel (round 4; e = 16: the four level-0 blocks of the 4-band net) -- autograd of reference
// models/common/LGT.py:112-146 (local_mixer), 183-219 (LGMixer: concat / proj / dropout), 45-61 (pre_norm, residual).
//
// Round 2/3 ran this as k_attn_bwd_core (one workgroup per window group AND head: dq / dk / dv "pieces" and the proj input to HBM, 235 MB)
// -> k_attn_bwd_epi (reads them back: to_qkv^T, LayerNorm backward, dx; 185 MB) -> k_wgrad_t (proj weight gradient: dym and cat once
// more), 184 + ~20 us and ~570 MB per level-0 block against a unit of ~135 MB (read x, dy, o2, dg; write dx).  Here nothing between
// the flash passes and dx leaves the chip:
//   * one workgroup = 8 waves = 4 windows x 2 heads (wave = (window slot, head)): the 64 pos_emb-gradient accumulators of a head stay
//     in the registers of that head's wave across all windows, exactly as before (they are why ONE wave cannot carry both heads);
//   * prologue and epilogue are split between the two waves of a window by TOKENS, not by head: lane = (token, channel half), each
//     wave takes 32 tokens.  x / dy are read once per window (not once per head), LayerNorm sums are one DPP step, q / k / v of BOTH
//     heads come out of the pair of lanes of a token (lane c computes head c) and go into the two waves' channel-major K / V / Q / dO
//     tiles; the to_qkv^T partial sums of the two heads meet in LDS, the LayerNorm backward and dx are formed by the same lanes that
//     read x, the global-mixer half (dg) joins there;
//   * every weight gradient that is a sum over pixels of an outer product runs on the matrix pipe with the TOKEN axis as K
//     (v_mfma_f32_16x16x4_f32: exact fp32 products, the same arithmetic as k_wgrad_t): dWqkv / dbqkv of a head = dqkv_h^T [12 x 64] .
//     [y1 | 1] [64 x 9] (16 MFMAs per window and wave, the bias through a column of ones), dWproj = dym^T [16 x 32] . cat [32 x 16]
//     (8 MFMAs; the two waves split the tokens).  The accumulators are 8 registers for the whole kernel; the 48 + 16 per-lane
//     accumulators of k_attn_bwd_epi<16,true> are gone;
//   * LayerNorm gamma / beta gradients: 16 per-lane sums in lane-owned LDS slots (two 16-byte read-modify-writes per window);
//   * dropout: k_proj_o2_bwd_k (which has to run in front of the FFT-mixer backward anyway) stores ONE keep-bit word per pixel instead of
//     the masked copy of dy (2 MB instead of 33.5 MB written and read back), and sums the proj bias gradient on the way.
// Three workgroup barriers per window group (tiles complete / partial sums complete / LDS free for the next group).
#include "kernels.h"
#include "bwd_kernels.h"
#include "mfma.h"
#include <utility>

// In-kernel phase stamps (diagnostic build only, -DLG_STAMPS: tools/build_stamps.sh + tools/attn_bwd_stamps.py): the eight waves of workgroup 0
// write s_memtime at the phase boundaries of their third window group; no stamp executes in the product build.
#ifdef LG_STAMPS
// branch-free: every wave of every workgroup stores every stamp (a conditional store splits the loop body into basic blocks and the
// register allocator then spills > 1000 registers: the instrumented kernel ran 19 x slower than the product one)
__device__ unsigned long long g_kf_stamps[512 * 8 * 8 * 16];   // [workgroup][wave][group of the workgroup][stamp]
#define STAMP(i) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)); \
                      g_kf_stamps[((blockIdx.x * 8 + wave) * 8 + (stamp_it & 7)) * 16 + (i)] = t__; } while (0)
extern "C" __attribute__((visibility("default"))) int lg_debug_kf_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_kf_stamps), sizeof(g_kf_stamps));
}
#else
#define STAMP(i) do { } while (0)
#endif

#ifndef LG_ATTNF_SB
#define LG_ATTNF_SB 2   // a scheduling fence behind every LG_ATTNF_SB-th key / query group of the flash passes (0: none).  Same-box A/B of the kernel's
                        // average launch: none 140.5 us, every group 123.3, every 2nd 122.9 (kept), every 4th 128.9, every 8th 130.3
#endif
#ifndef LG_ATTNF_DPP
#define LG_ATTNF_DPP 0    // A/B build (-DLG_ATTNF_DPP=1): broadcast operands of the recomputing flash passes through DPP row broadcasts instead of wave-uniform LDS reads.
                          // Measured SLOWER (142.7 against 124.7 us): a DPP multiply-add issues at the 4-cycle rate of the SDWA / conversion class and does ONE
                          // multiply-add per lane where v_pk_fma_f32 does two in 4.7 cycles -- the LDS pipe was relieved, the vector pipe got twice the work
#endif
#ifndef LG_ATTNF_PRIO
#define LG_ATTNF_PRIO 1
#endif
// (one opaque asm statement with its own scalar branch inside: as a C++ `if` around two s_setprio builtins the basic block was split in front of each pass and the
// register allocator spilled 470 registers)
#define ATTNF_PRIO(lead) do { if (LG_ATTNF_PRIO) asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 0\n\ts_branch 2f\n1:\ts_setprio 3\n2:" :: "s"(__builtin_amdgcn_readfirstlane((int)(lead))) : "scc"); } while (0)
#define PASS_FENCE(g) do { if (LG_ATTNF_SB && ((g) % LG_ATTNF_SB) == LG_ATTNF_SB - 1) __builtin_amdgcn_sched_barrier(0); } while (0)
namespace {
#ifndef LG_ATTNF_NS
#define LG_ATTNF_NS 4   // window slots per workgroup: 4 = one 8-wave workgroup per CU (155 KB of LDS); 2 = 4-wave workgroups, TWO per CU (81 KB each), so that
                        // one workgroup's prologue / epilogue could run beside the other's flash passes -- measured 142.8 vs 135.8 us: the kernel is
                        // vector-issue bound in EVERY phase (stamps: a window group costs 33 k ticks either way), decoupling the phases buys nothing
#endif
constexpr int F_HC = 8, F_E = 16, F_D = 4, F_NS = LG_ATTNF_NS, F_NW = 2 * F_NS, F_NT = 64 * F_NW;
constexpr int F_PLD = 68;                       // padded pos_emb row (16-byte aligned): lane = query reads its row as 16-byte pieces, lane = key a column;
                                                // both conflict-free (68 = 4 mod 64: the 16 lanes of a ds_read_b128 group sit 4 banks apart)
constexpr int F_Y1LD = 12;                      // [y1 (8) | 1 | 0 0 0] per token: B operand of the to_qkv weight-gradient product
constexpr int F_QLD = 12;                       // staged dqkv rows of one head per token: A operand
constexpr int F_WPS = 68;                       // lane-half stride of the dO weights (two lane-dependent addresses on different banks)
// LDS map (floats)
constexpr int F_OFF_WQ = 2 * 64 * F_PLD;        // [2 heads][12 rows = q c | k c | v c][8]
constexpr int F_OFF_BQ = F_OFF_WQ + 192;        // [2][12] (+ pad)
constexpr int F_OFF_WP = F_OFF_BQ + 32;         // [2 halves][8 n][8 = head * 4 + k], half stride F_WPS
constexpr int F_OFF_LN = F_OFF_WP + 144;        // gamma[16] | beta[16]
constexpr int F_OFF_WAVE = F_OFF_LN + 32;
constexpr int F_PW = 1856;                      // per wave: K | V | Q | dO [4][64] each, stats [3][64], to_qkv^T partial [64][8], LayerNorm-gradient slots [4 rows][2 halves][16]
constexpr int F_T_K = 0, F_T_V = 256, F_T_Q = 512, F_T_DO = 768, F_T_ST = 1024, F_T_PART = 1216, F_T_SLOT = 1728;
constexpr int F_OFF_SLOT = F_OFF_WAVE + F_NW * F_PW;
constexpr int F_PS = 1920;                      // per window slot: y1 image [64][12], cat image [64][16], (mu, rstd) [64][2]
constexpr int F_S_Y1 = 0, F_S_CAT = 768, F_S_MR = 1792;
constexpr int F_LOOP_FLOATS = F_OFF_SLOT + F_NS * F_PS;
constexpr int F_DPW = 64 * 65;                  // write-out: a wave's pos_emb-gradient columns [i][65] ...
constexpr int F_OUT_FLOATS = F_NW * F_DPW + F_NW * (32 + 512);   // ... + its LayerNorm sums [32] and the two MFMA accumulators [256] each
constexpr int F_LDS_FLOATS = F_LOOP_FLOATS > F_OUT_FLOATS ? F_LOOP_FLOATS : F_OUT_FLOATS;
static_assert(F_LDS_FLOATS * 4 <= (F_NS == 2 ? 80 : 160) * 1024, "LDS budget: two workgroups per CU at F_NS = 2");
static_assert(F_OFF_WAVE % 4 == 0 && F_PW % 4 == 0 && F_PS % 4 == 0, "16-byte alignment of the LDS regions");

__device__ __forceinline__ float dpp_xor1(float v) {   // the other lane of the token's pair
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
// workgroup barrier that orders LDS only: the operands requested ahead and the dx stores must stay in flight across it.  (On this toolchain
// __syncthreads() compiles to the same s_waitcnt lgkmcnt(0) + s_barrier as long as no LDS-DMA is pending -- checked in the assembly; what
// did drain the loads in the first version was the reload of a spilled register next to a barrier: scratch traffic shares the in-order
// vector-memory counter.  The explicit form states the intent and does not depend on that.)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// sum over the 8 lanes of a 16-lane row that share this lane's parity (= the row's 8 tokens of one channel half): xor 2, then rotations by
// 4 and 8 inside the row; every lane of the class ends up with the sum
__device__ __forceinline__ float row8_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));   // row_ror:8
    return v;
}
// Lane-broadcast operands (round 6).  In the flash passes every lane needs the SAME k / v (pass 1) or q / dO / row statistics (pass 2) of one token per
// multiply-add.  Read from LDS those are wave-uniform 16-byte reads: 64 lanes x 16 bytes through a 128-byte port = 8 cycles of the CU's ONE LDS pipe for 16
// useful bytes, ~450 of them per wave and window -- 8 waves x 3.6 k cycles = 29 k of a window group's 33.7 k cycles (SQ counters: 62 % LDS-busy, 56 %
// vector-busy; profiles/r06_sq_counters_step.txt).  Here a lane keeps the values of the four tokens 16 r + (lane & 15) in registers (one conflict-free
// ds_read_b32 each) and the multiply-adds take them through the DPP row broadcast of gfx90a+ (row_newbcast:n = lane n of the reader's own 16-lane row):
// the same fused multiply-adds in the same order, as plain instead of packed fp32 (same issue rate on this chip), and no LDS traffic.
// DPP hazard (2 wait states between a VALU write of a register and its use as the DPP source): the broadcast sources are only ever registers loaded from
// LDS a phase earlier; tools/check_dpp_hazards.py verifies it on the assembly.
template <int N> __device__ __forceinline__ float fmac_bc(float acc, float src, float b) {   // acc + src[lane N of the row] * b
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(b), "n"(N));
    return acc;
}
template <int N> __device__ __forceinline__ float sub_bc(float a, float src) {               // a - src[lane N of the row]
    float d;
    asm("v_subrev_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(src), "v"(a), "n"(N));
    return d;
}
template <int N> __device__ __forceinline__ float mul_bc(float a, float src) {               // a * src[lane N of the row]
    float d;
    asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(src), "v"(a), "n"(N));
    return d;
}
template <class F, int... Is> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
__device__ __forceinline__ float dpp_even(float v) {   // the even lane's value on both lanes of the pair
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xA0, 0xF, 0xF, true));   // quad_perm [0,0,2,2]
}
}  // namespace

// STATS (round 6, VERDICT r5 lever a in its light form): the forward's saving launch left, per token and head, the log-sum-exp of the score row (log2 domain)
// and the attention output O (k_attn_m: a.sl, a.so).  Pass 1 then needs no row maximum, no row sum, no normalisation and no P V product -- ONE loop over the
// keys (scores -> P = 2^(s - L) -> dP -> dS -> dq) that reads k and v once; D_i = dO_i . O_i and the cat image come out of the prologue.
template <bool STATS>
__global__ __launch_bounds__(F_NT) __attribute__((amdgpu_waves_per_eu(2))) void k_attn_bwd_f(AttnBwdFArgs a, int nwin, int ngroups) {
    constexpr int HC = F_HC, E = F_E, D = F_D, PLD = F_PLD;
    constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int uw = __builtin_amdgcn_readfirstlane(wave);   // provably wave-uniform
    const int ws = wave >> 1, hd = wave & 1;
    float* const sPosH = smem + hd * 64 * PLD;                 // pos_emb[hd][i][j] * log2(e)
    float* const sWq = smem + F_OFF_WQ;
    float* const sBq = smem + F_OFF_BQ;
    float* const sWp = smem + F_OFF_WP;
    float* const sLn = smem + F_OFF_LN;
    float* const mine = smem + F_OFF_WAVE + wave * F_PW;       // this wave's region
    float* const sK = mine + F_T_K;
    float* const sV = mine + F_T_V;
    float* const sQ = mine + F_T_Q;
    float* const sDO = mine + F_T_DO;
    float* const sSt = mine + F_T_ST;
    float* const sSlot = mine + F_T_SLOT;
    float* const slotw = smem + F_OFF_SLOT + ws * F_PS;        // this window slot's images
    float* const sY1 = slotw + F_S_Y1;
    float* const sCat = slotw + F_S_CAT;
    float* const sMr = slotw + F_S_MR;

#ifdef LG_STAMPS
    { const int stamp_it = 0; STAMP(11); }
#endif
    // ---- staging, once per workgroup
    {   // every staged value of a thread is requested before the first one is stored (a load + wait per loop trip, or per exec-masked `if`, is a
        // dependent L2 round trip each: 16 + 4 of them in the first version), unconditionally from clamped indices
        constexpr int NPV = 2 * 64 * 64 / F_NT;
        float pv[NPV];
#pragma unroll
        for (int k = 0; k < NPV; ++k) pv[k] = a.pos[k * F_NT + threadIdx.x];
        const int t = threadIdx.x;
        // sWq[h][third * 4 + c][k] = qkvw[third * HC + h * D + c][k]   (t < 192)
        const int tq = t < 192 ? t : 191, hq = tq / 96, rq = (tq % 96) >> 3, kq = tq & 7;
        const float vwq = a.qkvw[((rq >> 2) * HC + hq * D + (rq & 3)) * HC + kq];
        const int tb = t < 24 ? t : 23, hb = tb / 12, rb = tb % 12;
        const float vbq = a.qkvb[(rb >> 2) * HC + hb * D + (rb & 3)];
        // sWp[c][u][hk] = projw[HC * c + u][hk]   (hk = head * 4 + k: the attention columns of proj; t < 128)
        const int tp = t < 128 ? t : 127, cp = tp >> 6, up = (tp >> 3) & 7, hkp = tp & 7;
        const float vwp = a.projw[(HC * cp + up) * E + hkp];
        const int tl = t < 32 ? t : 31;
        const float vlg = a.ln1g[tl & 15], vlb = a.ln1b[tl & 15];
        const float vln = tl < 16 ? vlg : vlb;
#pragma unroll
        for (int k = 0; k < NPV; ++k) {
            const int i = k * F_NT + threadIdx.x;
            smem[(i >> 12) * 64 * PLD + ((i >> 6) & 63) * PLD + (i & 63)] = pv[k] * LOG2E;
        }
        if (t < 192) sWq[t] = vwq;
        if (t < 24) sBq[t] = vbq;
        if (t < 128) sWp[cp * F_WPS + up * 8 + hkp] = vwp;
        if (t < 32) sLn[t] = vln;
    }
    // LayerNorm gamma / beta gradient slots: the 8 lanes of a 16-lane row that hold one channel half pre-sum their 8 tokens with three DPP
    // steps and share ONE slot of 16 floats (every one of them writes the same sum to it: 512 bytes per wave instead of 4 KB)
    float* const myslot = sSlot + ((lane >> 4) * 2 + (lane & 1)) * 16;
    sSlot[lane] = 0.f;
    sSlot[64 + lane] = 0.f;
    const int nwx = a.w >> 3, nwy = a.h >> 3;
    const long hw = (long)a.h * a.w;
    const float scale = 0.5f;   // D^-1/2, D = 4
    // pos_emb gradient: in pass 2 lane j owns column j of dS of THIS wave's head, accumulated in 64 registers across all windows
    lg_v2f dpacc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) dpacc[i] = (lg_v2f){0.f, 0.f};
    f32x4 accq = {0.f, 0.f, 0.f, 0.f};   // dWqkv / dbqkv of this head: D[row = third * 4 + c][col = k | 8 = bias]
    f32x4 accp = {0.f, 0.f, 0.f, 0.f};   // dWproj partial: D[n][k]
    const int mr = lane & 15, mg = lane >> 4;   // MFMA operand coordinates of this lane
    // lane = (token of this wave's half of the window, channel half) in the prologue and the epilogue
    const int tk = 32 * hd + (lane >> 1), ch = lane & 1;
    __syncthreads();

    // pixel of (window, token tk): plane index bT, offset in the plane sT, NHWC pixel (return value).  Recomputed where needed rather than
    // kept across the flash passes (6 registers at the kernel's peak)
    auto pixel_of = [&](int win, long& bT, long& sT) -> long {
        // divisions by the window counts through host-made reciprocals (exact for win < 2^24 and counts < 256: launcher)
        const int rr = (int)__umulhi((unsigned)win, a.rcp_nwx), wx = win - rr * nwx;
        const int bi = (int)__umulhi((unsigned)rr, a.rcp_nwy), wy = rr - bi * nwy;
        bT = bi;
        sT = (long)(wy * 8 + (tk >> 3)) * a.w + wx * 8 + (tk & 7);
        return bT * hw + sT;
    };
    // Every wave of the workgroup is in the same phase (three barriers per group), so an HBM round trip that is waited for where it is
    // issued is paid by the whole CU.  All global operands are therefore requested a phase or more ahead, unconditionally:
    //   * the prologue's x / dy rows and keep word of the NEXT window group behind the flash passes of this one (n*: 17 registers across
    //     the low-pressure epilogue phases);
    //   * the epilogue's operands of THIS group (x / dy again -- an L2 hit --, dg, o2, keep) behind pass 1, across pass 2 (e*: 29 registers).
    const uint32_t* const keepp = a.keep ? a.keep : reinterpret_cast<const uint32_t*>(a.x);   // no dropout: any valid address, value unused
    float4 nx0, nx1, nd0, nd1, nO;
    float nL;
    uint32_t nkw;
    auto issue_prologue = [&](int win) {
        long bT, sT;
        const long pT = pixel_of(win, bT, sT);
        const float4* xs = reinterpret_cast<const float4*>(a.x + pT * E + HC * ch);
        const float4* ds = reinterpret_cast<const float4*>(a.dy + pT * E + HC * ch);
        nx0 = xs[0]; nx1 = xs[1]; nd0 = ds[0]; nd1 = ds[1];
        nkw = keepp[pT];
        if constexpr (STATS) {   // head `ch` of the token: the forward's O (4 channels) and row log-sum-exp
            nO = *reinterpret_cast<const float4*>(a.so + pT * HC + D * ch);
            nL = a.sl[pT * 2 + ch];
        }
    };
    if (blockIdx.x < ngroups) issue_prologue(blockIdx.x * F_NS + ws);
#pragma unroll 1
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int win = grp * F_NS + ws;   // nwin is a multiple of F_NS (launcher): every window slot is live
#ifdef LG_STAMPS
        const int stamp_it = (grp - (int)blockIdx.x) / (int)gridDim.x;
#endif
        STAMP(0);
        {
            // ---------------- prologue: lane = (token tk, channel half ch)
            const float4 x0 = nx0, x1 = nx1, d0 = nd0, d1 = nd1;
            const uint32_t kw = a.keep ? nkw : 0xffffffffu;
            float4 O4 = make_float4(0.f, 0.f, 0.f, 0.f);
            float Lrow = 0.f;
            if constexpr (STATS) { O4 = nO; Lrow = nL; }
            float xh[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            float sm = ((xh[0] + xh[1]) + (xh[2] + xh[3])) + ((xh[4] + xh[5]) + (xh[6] + xh[7]));
            sm += dpp_xor1(sm);
            const float mu = sm * (1.0f / E);
            float vs = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) { xh[u] -= mu; vs += xh[u] * xh[u]; }
            vs += dpp_xor1(vs);
            const float rstd = __builtin_amdgcn_rsqf(vs * (1.0f / E) + LG_EPS);
            float y1[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) y1[k] = dpp_even(xh[k] * rstd * sLn[k] + sLn[16 + k]);   // LN1(x)[:8] of the token on both lanes
            // q, k, v of head `ch` for this token -> that head's wave tiles (channel-major [c][token])
            float* const tile = smem + F_OFF_WAVE + (2 * ws + ch) * F_PW;
            const float4* wq4 = reinterpret_cast<const float4*>(sWq + ch * 96);
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                const float4 wa = wq4[2 * r], wb = wq4[2 * r + 1];
                float acc = sBq[ch * 12 + r];
                acc += wa.x * y1[0]; acc += wa.y * y1[1]; acc += wa.z * y1[2]; acc += wa.w * y1[3];
                acc += wb.x * y1[4]; acc += wb.y * y1[5]; acc += wb.z * y1[6]; acc += wb.w * y1[7];
                if (r < 4) tile[F_T_Q + r * 64 + tk] = acc * (scale * LOG2E);   // scores live in the log2 domain
                else if (r < 8) tile[F_T_K + (r - 4) * 64 + tk] = acc;
                else tile[F_T_V + (r - 8) * 64 + tk] = acc;
            }
            // dO = (proj^T dym)[attention columns]: this lane's 8 channels of dym against both heads, then the pair adds up
            const float dyv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            float pr[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) pr[k] = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float dm = ((kw >> (8 * ch + u)) & 1u) ? (a.keep ? dyv[u] * (1.0f / 0.9f) : dyv[u]) : 0.0f;
                const float4 wa = *reinterpret_cast<const float4*>(sWp + ch * F_WPS + u * 8);
                const float4 wb = *reinterpret_cast<const float4*>(sWp + ch * F_WPS + u * 8 + 4);
                pr[0] += wa.x * dm; pr[1] += wa.y * dm; pr[2] += wa.z * dm; pr[3] += wa.w * dm;
                pr[4] += wb.x * dm; pr[5] += wb.y * dm; pr[6] += wb.z * dm; pr[7] += wb.w * dm;
            }
            float dOh[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float send = ch ? pr[k] : pr[4 + k];     // what the other lane keeps
                const float keepv = ch ? pr[4 + k] : pr[k];
                dOh[k] = keepv + dpp_xor1(send);
                tile[F_T_DO + k * 64 + tk] = dOh[k];
            }
            if constexpr (STATS) {   // row statistics of (token tk, head ch) for both passes; the head's attention output into the window's cat image
                const float Dv = ((dOh[0] * O4.x + dOh[1] * O4.y) + dOh[2] * O4.z) + dOh[3] * O4.w;   // D_i = sum_j P_ij dP_ij = dO_i . O_i
                tile[F_T_ST + tk] = Lrow;
                tile[F_T_ST + 128 + tk] = Dv;
                *reinterpret_cast<float4*>(sCat + tk * E + D * ch) = O4;
            }
            // y1 image of the window (B operand of the to_qkv weight-gradient product; column 8 = 1 carries the bias gradient)
            float4* y4 = reinterpret_cast<float4*>(sY1 + tk * F_Y1LD);
            if (ch == 0) {
                y4[0] = make_float4(y1[0], y1[1], y1[2], y1[3]);
                y4[1] = make_float4(y1[4], y1[5], y1[6], y1[7]);
                *reinterpret_cast<float2*>(sMr + 2 * tk) = make_float2(mu, rstd);
            } else {
                y4[2] = make_float4(1.f, 0.f, 0.f, 0.f);
            }
        }
        STAMP(1);
        lds_barrier();   // B1: the tiles of every wave are complete
        STAMP(2);
        // The two waves of a SIMD (wave w and w + 4: slots of different windows) are not served alike: the arbiter issues the OLDER one first (stamps of round 4: pass 1
        // takes waves 0 - 3 10 k cycles and waves 4 - 7 13.4 k, and the first four then wait 5.8 k at the barrier).  LG_ATTNF_PRIO: the older half leads in pass 1, the
        // younger half in pass 2 -- each wave has the SIMD for one of the two long passes and both halves reach the barrier together.
        ATTNF_PRIO(uw < 4);
        float dqkv[12];
        {
            // ---------------- pass 1: lane = query i, packed over KEY pairs (as k_attn_bwd_core)
            float q[D], dOi[D];
#pragma unroll
            for (int c = 0; c < D; ++c) { q[c] = sQ[c * 64 + lane]; dOi[c] = sDO[c * 64 + lane]; }
            if constexpr (STATS) {
                const float* prow = sPosH + lane * PLD;
                const float Li = sSt[lane], Dv = sSt[128 + lane];
                const lg_v2f L2v = (lg_v2f){Li, Li}, Dv2 = (lg_v2f){Dv, Dv};
                lg_v2f dq2[D];
#pragma unroll
                for (int c = 0; c < D; ++c) dq2[c] = (lg_v2f){0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float4 pr4 = reinterpret_cast<const float4*>(prow)[g];
                    lg_v2f sp0 = (lg_v2f){pr4.x, pr4.y}, sp1 = (lg_v2f){pr4.z, pr4.w};
                    lg_v2f dP0 = (lg_v2f){0.f, 0.f}, dP1 = (lg_v2f){0.f, 0.f};
                    float4 kv[D];
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        kv[c] = reinterpret_cast<const float4*>(sK)[c * 16 + g];
                        const float4 vv = reinterpret_cast<const float4*>(sV)[c * 16 + g];
                        const lg_v2f qq = (lg_v2f){q[c], q[c]}, dd = (lg_v2f){dOi[c], dOi[c]};
                        sp0 = qq * (lg_v2f){kv[c].x, kv[c].y} + sp0;
                        sp1 = qq * (lg_v2f){kv[c].z, kv[c].w} + sp1;
                        dP0 = dd * (lg_v2f){vv.x, vv.y} + dP0;
                        dP1 = dd * (lg_v2f){vv.z, vv.w} + dP1;
                    }
                    const lg_v2f e0 = sp0 - L2v, e1 = sp1 - L2v;
                    const lg_v2f P0 = (lg_v2f){__builtin_amdgcn_exp2f(e0.x), __builtin_amdgcn_exp2f(e0.y)};
                    const lg_v2f P1 = (lg_v2f){__builtin_amdgcn_exp2f(e1.x), __builtin_amdgcn_exp2f(e1.y)};
                    const lg_v2f dS0 = P0 * (dP0 - Dv2), dS1 = P1 * (dP1 - Dv2);
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        dq2[c] = dS0 * (lg_v2f){kv[c].x, kv[c].y} + dq2[c];
                        dq2[c] = dS1 * (lg_v2f){kv[c].z, kv[c].w} + dq2[c];
                    }
                    PASS_FENCE(g);
                }
#pragma unroll
                for (int c = 0; c < D; ++c) dqkv[c] = (dq2[c].x + dq2[c].y) * scale;
            } else {
#if LG_ATTNF_DPP
            const int l15 = lane & 15;
            float Kr[4][D], Vr[4][D];   // [r][c] = k / v channel c of token 16 r + (lane & 15)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < D; ++c) { Kr[r][c] = sK[c * 64 + 16 * r + l15]; Vr[r][c] = sV[c * 64 + 16 * r + l15]; }
            const float* prow = sPosH + lane * PLD;
            float sc[64];
            float mx = -3.0e38f;
            static_for<16>([&](auto gg) {
                constexpr int g = decltype(gg)::value, r = g >> 2, n0 = 4 * (g & 3);
                const float4 pr4 = reinterpret_cast<const float4*>(prow)[g];
                float s0 = pr4.x, s1 = pr4.y, s2 = pr4.z, s3 = pr4.w;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    s0 = fmac_bc<n0>(s0, Kr[r][c], q[c]);
                    s1 = fmac_bc<n0 + 1>(s1, Kr[r][c], q[c]);
                    s2 = fmac_bc<n0 + 2>(s2, Kr[r][c], q[c]);
                    s3 = fmac_bc<n0 + 3>(s3, Kr[r][c], q[c]);
                }
                sc[4 * g] = s0; sc[4 * g + 1] = s1; sc[4 * g + 2] = s2; sc[4 * g + 3] = s3;
                mx = fmaxf(mx, fmaxf(fmaxf(s0, s1), fmaxf(s2, s3)));
                PASS_FENCE(g);
            });
            float le = 0.f, lo = 0.f;   // even / odd keys: the two halves of the packed sum of rounds 4 - 5 (same additions in the same order)
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                sc[2 * k] = __builtin_amdgcn_exp2f(sc[2 * k] - mx);
                sc[2 * k + 1] = __builtin_amdgcn_exp2f(sc[2 * k + 1] - mx);
                le += sc[2 * k]; lo += sc[2 * k + 1];
            }
            const float inv = __builtin_amdgcn_rcpf(le + lo);
            float Oe[D], Oo[D];
#pragma unroll
            for (int c = 0; c < D; ++c) { Oe[c] = 0.f; Oo[c] = 0.f; }
            static_for<16>([&](auto gg) {
                constexpr int g = decltype(gg)::value, r = g >> 2, n0 = 4 * (g & 3);
                sc[4 * g] *= inv; sc[4 * g + 1] *= inv; sc[4 * g + 2] *= inv; sc[4 * g + 3] *= inv;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    Oe[c] = fmac_bc<n0>(Oe[c], Vr[r][c], sc[4 * g]);
                    Oo[c] = fmac_bc<n0 + 1>(Oo[c], Vr[r][c], sc[4 * g + 1]);
                    Oe[c] = fmac_bc<n0 + 2>(Oe[c], Vr[r][c], sc[4 * g + 2]);
                    Oo[c] = fmac_bc<n0 + 3>(Oo[c], Vr[r][c], sc[4 * g + 3]);
                }
                PASS_FENCE(g);
            });
            float O[D];
            float Dv = 0.f;   // D_i = sum_j P_ij dP_ij = dO_i . O_i
#pragma unroll
            for (int c = 0; c < D; ++c) { O[c] = Oe[c] + Oo[c]; Dv += dOi[c] * O[c]; }
            float dqe[D], dqo[D];
#pragma unroll
            for (int c = 0; c < D; ++c) { dqe[c] = 0.f; dqo[c] = 0.f; }
            static_for<16>([&](auto gg) {
                constexpr int g = decltype(gg)::value, r = g >> 2, n0 = 4 * (g & 3);
                float dP0 = 0.f, dP1 = 0.f, dP2 = 0.f, dP3 = 0.f;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    dP0 = fmac_bc<n0>(dP0, Vr[r][c], dOi[c]);
                    dP1 = fmac_bc<n0 + 1>(dP1, Vr[r][c], dOi[c]);
                    dP2 = fmac_bc<n0 + 2>(dP2, Vr[r][c], dOi[c]);
                    dP3 = fmac_bc<n0 + 3>(dP3, Vr[r][c], dOi[c]);
                }
                const float dS0 = sc[4 * g] * (dP0 - Dv), dS1 = sc[4 * g + 1] * (dP1 - Dv), dS2 = sc[4 * g + 2] * (dP2 - Dv), dS3 = sc[4 * g + 3] * (dP3 - Dv);
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    dqe[c] = fmac_bc<n0>(dqe[c], Kr[r][c], dS0);
                    dqo[c] = fmac_bc<n0 + 1>(dqo[c], Kr[r][c], dS1);
                    dqe[c] = fmac_bc<n0 + 2>(dqe[c], Kr[r][c], dS2);
                    dqo[c] = fmac_bc<n0 + 3>(dqo[c], Kr[r][c], dS3);
                }
                PASS_FENCE(g);
            });
#pragma unroll
            for (int c = 0; c < D; ++c) dqkv[c] = (dqe[c] + dqo[c]) * scale;
#else
            const float* prow = sPosH + lane * PLD;
            lg_v2f sc[32];
            float mx = -3.0e38f;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float4 pr4 = reinterpret_cast<const float4*>(prow)[g];
                lg_v2f sp0 = (lg_v2f){pr4.x, pr4.y}, sp1 = (lg_v2f){pr4.z, pr4.w};
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    const float4 kv = reinterpret_cast<const float4*>(sK)[c * 16 + g];
                    const lg_v2f qq = (lg_v2f){q[c], q[c]};
                    sp0 = qq * (lg_v2f){kv.x, kv.y} + sp0;
                    sp1 = qq * (lg_v2f){kv.z, kv.w} + sp1;
                }
                sc[2 * g] = sp0; sc[2 * g + 1] = sp1;
#ifndef LG_ATTNF_DIAG_FWDSTATS
                mx = fmaxf(mx, fmaxf(fmaxf(sp0.x, sp0.y), fmaxf(sp1.x, sp1.y)));
#endif
                PASS_FENCE(g);
            }
            asm volatile("" ::: "memory");
#ifdef LG_ATTNF_DIAG_FWDSTATS
            // TIMING-ONLY diagnostic (never in the product build; results are wrong): what pass 1 would cost if the forward had saved the rows'
            // log-sum-exp and the attention output (VERDICT r5 lever a, "light" form): no row maximum, no row sum, no normalisation, no P V
            mx = q[0];
#endif
            lg_v2f l2 = (lg_v2f){0.f, 0.f};
            const lg_v2f mx2 = (lg_v2f){mx, mx};
#pragma unroll
            for (int g = 0; g < 32; ++g) {
                const lg_v2f t = sc[g] - mx2;
                sc[g] = (lg_v2f){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
#ifndef LG_ATTNF_DIAG_FWDSTATS
                l2 += sc[g];
#endif
            }
#ifdef LG_ATTNF_DIAG_FWDSTATS
            l2 = (lg_v2f){dOi[0], dOi[1]};
#endif
            const float inv = __builtin_amdgcn_rcpf(l2.x + l2.y);
            const lg_v2f inv2 = (lg_v2f){inv, inv};
            lg_v2f O2[D];
#pragma unroll
            for (int c = 0; c < D; ++c) O2[c] = (lg_v2f){0.f, 0.f};
#pragma unroll
#ifdef LG_ATTNF_DIAG_FWDSTATS
            for (int c = 0; c < D; ++c) O2[c] = (lg_v2f){q[c] * inv, dOi[c]};
#else
            for (int g = 0; g < 16; ++g) {
                sc[2 * g] *= inv2; sc[2 * g + 1] *= inv2;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    const float4 vv = reinterpret_cast<const float4*>(sV)[c * 16 + g];
                    O2[c] = sc[2 * g] * (lg_v2f){vv.x, vv.y} + O2[c];
                    O2[c] = sc[2 * g + 1] * (lg_v2f){vv.z, vv.w} + O2[c];
                }
                PASS_FENCE(g);
            }
#endif
            float O[D];
            float Dv = 0.f;   // D_i = sum_j P_ij dP_ij = dO_i . O_i
#pragma unroll
            for (int c = 0; c < D; ++c) { O[c] = O2[c].x + O2[c].y; Dv += dOi[c] * O[c]; }
            asm volatile("" ::: "memory");   // re-read K / V from LDS below instead of keeping 64 x 2D values live
            const lg_v2f Dv2 = (lg_v2f){Dv, Dv};
            lg_v2f dq2[D];
#pragma unroll
            for (int c = 0; c < D; ++c) dq2[c] = (lg_v2f){0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                lg_v2f dP0 = (lg_v2f){0.f, 0.f}, dP1 = (lg_v2f){0.f, 0.f};
                float4 kv[D];
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    const float4 vv = reinterpret_cast<const float4*>(sV)[c * 16 + g];
                    kv[c] = reinterpret_cast<const float4*>(sK)[c * 16 + g];
                    const lg_v2f dd = (lg_v2f){dOi[c], dOi[c]};
                    dP0 = dd * (lg_v2f){vv.x, vv.y} + dP0;
                    dP1 = dd * (lg_v2f){vv.z, vv.w} + dP1;
                }
                const lg_v2f dS0 = sc[2 * g] * (dP0 - Dv2), dS1 = sc[2 * g + 1] * (dP1 - Dv2);
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    dq2[c] = dS0 * (lg_v2f){kv[c].x, kv[c].y} + dq2[c];
                    dq2[c] = dS1 * (lg_v2f){kv[c].z, kv[c].w} + dq2[c];
                }
                PASS_FENCE(g);
            }
#pragma unroll
            for (int c = 0; c < D; ++c) dqkv[c] = (dq2[c].x + dq2[c].y) * scale;
#endif
            // this head's attention output into the window's cat image (proj input: B operand of the proj weight-gradient product)
            *reinterpret_cast<float4*>(sCat + lane * E + D * hd) = make_float4(O[0], O[1], O[2], O[3]);
            sSt[lane] = mx;
            sSt[64 + lane] = inv;
            sSt[128 + lane] = Dv;
                    }
        }
        STAMP(3);
        ATTNF_PRIO(uw >= 4);
        // the epilogue's global operands of this group, requested across pass 2
        float4 ex0, ex1, ed0, ed1;
        uint32_t ekw;
        float edg[8], eo2[4];
        {
            long bT, sT;
            const long pT = pixel_of(win, bT, sT);
            const float4* xs = reinterpret_cast<const float4*>(a.x + pT * E + HC * ch);
            const float4* ds = reinterpret_cast<const float4*>(a.dy + pT * E + HC * ch);
            ex0 = xs[0]; ex1 = xs[1]; ed0 = ds[0]; ed1 = ds[1];
            ekw = keepp[pT];
#pragma unroll
            for (int k = 0; k < 8; ++k) edg[k] = a.dg[(bT * HC + k) * hw + sT];           // both lanes of the pair ask for the same address
#pragma unroll
            for (int k = 0; k < 4; ++k) eo2[k] = a.o2[(bT * HC + 4 * ch + k) * hw + sT];   // lane ch stages o2 channels [4 ch, 4 ch + 4)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the row statistics are this wave's own
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            // ---------------- pass 2: lane = key j, packed over QUERY pairs
            // the lane's own k / v as FULL register pairs: a splat by op_sel reads one half of a pair whose other half the allocator
            // gives to anything -- here to the destinations of the loads in flight, and the wait for those then sits in front of pass 2
#if LG_ATTNF_DPP
            const int l15 = lane & 15;
            float Qr[4][D], dOr[4][D], Mr[4], Ir[4], Dr[4];   // q / dO channel c and the row statistics of query 16 r + (lane & 15)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int c = 0; c < D; ++c) { Qr[r][c] = sQ[c * 64 + 16 * r + l15]; dOr[r][c] = sDO[c * 64 + 16 * r + l15]; }
                Mr[r] = sSt[16 * r + l15]; Ir[r] = sSt[64 + 16 * r + l15]; Dr[r] = sSt[128 + 16 * r + l15];
            }
            float kj[D], vj[D], dke[D], dko[D], dve[D], dvo[D];
#pragma unroll
            for (int c = 0; c < D; ++c) { kj[c] = sK[c * 64 + lane]; vj[c] = sV[c * 64 + lane]; dke[c] = 0.f; dko[c] = 0.f; dve[c] = 0.f; dvo[c] = 0.f; }
            const float* pcol = sPosH + lane;
            static_for<16>([&](auto gg) {
                constexpr int g = decltype(gg)::value, r = g >> 2, n0 = 4 * (g & 3);
                float t0 = pcol[(4 * g) * PLD], t1 = pcol[(4 * g + 1) * PLD], t2 = pcol[(4 * g + 2) * PLD], t3 = pcol[(4 * g + 3) * PLD];
                float dP0 = 0.f, dP1 = 0.f, dP2 = 0.f, dP3 = 0.f;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    t0 = fmac_bc<n0>(t0, Qr[r][c], kj[c]);
                    t1 = fmac_bc<n0 + 1>(t1, Qr[r][c], kj[c]);
                    t2 = fmac_bc<n0 + 2>(t2, Qr[r][c], kj[c]);
                    t3 = fmac_bc<n0 + 3>(t3, Qr[r][c], kj[c]);
                    dP0 = fmac_bc<n0>(dP0, dOr[r][c], vj[c]);
                    dP1 = fmac_bc<n0 + 1>(dP1, dOr[r][c], vj[c]);
                    dP2 = fmac_bc<n0 + 2>(dP2, dOr[r][c], vj[c]);
                    dP3 = fmac_bc<n0 + 3>(dP3, dOr[r][c], vj[c]);
                }
                const float P0 = mul_bc<n0>(__builtin_amdgcn_exp2f(sub_bc<n0>(t0, Mr[r])), Ir[r]);
                const float P1 = mul_bc<n0 + 1>(__builtin_amdgcn_exp2f(sub_bc<n0 + 1>(t1, Mr[r])), Ir[r]);
                const float P2 = mul_bc<n0 + 2>(__builtin_amdgcn_exp2f(sub_bc<n0 + 2>(t2, Mr[r])), Ir[r]);
                const float P3 = mul_bc<n0 + 3>(__builtin_amdgcn_exp2f(sub_bc<n0 + 3>(t3, Mr[r])), Ir[r]);
                const float dS0 = P0 * sub_bc<n0>(dP0, Dr[r]), dS1 = P1 * sub_bc<n0 + 1>(dP1, Dr[r]);
                const float dS2 = P2 * sub_bc<n0 + 2>(dP2, Dr[r]), dS3 = P3 * sub_bc<n0 + 3>(dP3, Dr[r]);
#ifdef LG_DEGRADE_DPOS   // diagnostic variant only (never in the product build): a 5 % error in what enters the pos_emb gradient -- what
                        // tests/test_gpu_benchsize.py::test_cancelling_sum_gradient_kinds_band_against_band must turn red on
                dpacc[2 * g] += (lg_v2f){dS0, dS1} * 1.05f;
                dpacc[2 * g + 1] += (lg_v2f){dS2, dS3} * 1.05f;
#else
                dpacc[2 * g] += (lg_v2f){dS0, dS1};
                dpacc[2 * g + 1] += (lg_v2f){dS2, dS3};
#endif
                PASS_FENCE(g);
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    dve[c] = fmac_bc<n0>(dve[c], dOr[r][c], P0);
                    dvo[c] = fmac_bc<n0 + 1>(dvo[c], dOr[r][c], P1);
                    dve[c] = fmac_bc<n0 + 2>(dve[c], dOr[r][c], P2);
                    dvo[c] = fmac_bc<n0 + 3>(dvo[c], dOr[r][c], P3);
                    dke[c] = fmac_bc<n0>(dke[c], Qr[r][c], dS0);
                    dko[c] = fmac_bc<n0 + 1>(dko[c], Qr[r][c], dS1);
                    dke[c] = fmac_bc<n0 + 2>(dke[c], Qr[r][c], dS2);
                    dko[c] = fmac_bc<n0 + 3>(dko[c], Qr[r][c], dS3);
                }
            });
#pragma unroll
            for (int c = 0; c < D; ++c) { dqkv[4 + c] = (dke[c] + dko[c]) * LN2; dqkv[8 + c] = dve[c] + dvo[c]; }   // sQ carries log2(e)
#else
            lg_v2f kj2[D], vj2[D];
            lg_v2f dk2[D], dv2[D];
#pragma unroll
            for (int c = 0; c < D; ++c) {
                const float kc = sK[c * 64 + lane], vc = sV[c * 64 + lane];
                kj2[c] = (lg_v2f){kc, kc}; vj2[c] = (lg_v2f){vc, vc};
                asm volatile("" : "+v"(kj2[c]), "+v"(vj2[c]));
                dk2[c] = (lg_v2f){0.f, 0.f}; dv2[c] = (lg_v2f){0.f, 0.f};
            }
            const float* pcol = sPosH + lane;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                lg_v2f t0 = (lg_v2f){pcol[(4 * g) * PLD], pcol[(4 * g + 1) * PLD]}, t1 = (lg_v2f){pcol[(4 * g + 2) * PLD], pcol[(4 * g + 3) * PLD]};
                lg_v2f dP0 = (lg_v2f){0.f, 0.f}, dP1 = (lg_v2f){0.f, 0.f};
                float4 qi[D], doi[D];
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    qi[c] = reinterpret_cast<const float4*>(sQ)[c * 16 + g];
                    doi[c] = reinterpret_cast<const float4*>(sDO)[c * 16 + g];
                    const lg_v2f kk = kj2[c], vv = vj2[c];
                    t0 = kk * (lg_v2f){qi[c].x, qi[c].y} + t0;
                    t1 = kk * (lg_v2f){qi[c].z, qi[c].w} + t1;
                    dP0 = vv * (lg_v2f){doi[c].x, doi[c].y} + dP0;
                    dP1 = vv * (lg_v2f){doi[c].z, doi[c].w} + dP1;
                }
                const float4 smx = reinterpret_cast<const float4*>(sSt)[g];
                float4 sinv = make_float4(1.f, 1.f, 1.f, 1.f);
                if constexpr (!STATS) sinv = reinterpret_cast<const float4*>(sSt)[16 + g];
                const float4 sdv = reinterpret_cast<const float4*>(sSt)[32 + g];
                const lg_v2f e0 = t0 - (lg_v2f){smx.x, smx.y}, e1 = t1 - (lg_v2f){smx.z, smx.w};
                lg_v2f P0 = (lg_v2f){__builtin_amdgcn_exp2f(e0.x), __builtin_amdgcn_exp2f(e0.y)};
                lg_v2f P1 = (lg_v2f){__builtin_amdgcn_exp2f(e1.x), __builtin_amdgcn_exp2f(e1.y)};
                if constexpr (!STATS) { P0 *= (lg_v2f){sinv.x, sinv.y}; P1 *= (lg_v2f){sinv.z, sinv.w}; }   // (STATS: the row's log-sum-exp is in smx)
                const lg_v2f dS0 = P0 * (dP0 - (lg_v2f){sdv.x, sdv.y}), dS1 = P1 * (dP1 - (lg_v2f){sdv.z, sdv.w});
#ifdef LG_DEGRADE_DPOS   // diagnostic variant only (never in the product build): a 5 % error in what enters the pos_emb gradient -- what
                        // tests/test_gpu_benchsize.py::test_cancelling_sum_gradient_kinds_band_against_band must turn red on
                dpacc[2 * g] += dS0 * 1.05f;
                dpacc[2 * g + 1] += dS1 * 1.05f;
#else
                dpacc[2 * g] += dS0;
                dpacc[2 * g + 1] += dS1;
#endif
                PASS_FENCE(g);
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    dv2[c] = P0 * (lg_v2f){doi[c].x, doi[c].y} + dv2[c];
                    dv2[c] = P1 * (lg_v2f){doi[c].z, doi[c].w} + dv2[c];
                    dk2[c] = dS0 * (lg_v2f){qi[c].x, qi[c].y} + dk2[c];
                    dk2[c] = dS1 * (lg_v2f){qi[c].z, qi[c].w} + dk2[c];
                }
            }
#pragma unroll
            for (int c = 0; c < D; ++c) { dqkv[4 + c] = (dk2[c].x + dk2[c].y) * LN2; dqkv[8 + c] = dv2[c].x + dv2[c].y; }   // sQ carries log2(e)
        #endif
        }
        STAMP(4);
        if (LG_ATTNF_PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // every lane is done with the wave's K / V / Q / dO tiles: they become the dqkv image
        {   // the next group's prologue operands (a repeat of this group's addresses when there is no next one)
            const int gn = grp + (int)gridDim.x;
            issue_prologue((gn < ngroups ? gn : grp) * F_NS + ws);
        }
        {
            // ---------------- E1: lane = token: this head's share of to_qkv^T dqkv, and its to_qkv weight gradient on the matrix pipe
            float pt[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) pt[k] = 0.f;
            const float4* wq4 = reinterpret_cast<const float4*>(sWq + hd * 96);
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                const float4 wa = wq4[2 * r], wb = wq4[2 * r + 1];
                pt[0] += wa.x * dqkv[r]; pt[1] += wa.y * dqkv[r]; pt[2] += wa.z * dqkv[r]; pt[3] += wa.w * dqkv[r];
                pt[4] += wb.x * dqkv[r]; pt[5] += wb.y * dqkv[r]; pt[6] += wb.z * dqkv[r]; pt[7] += wb.w * dqkv[r];
            }
            float4* pp = reinterpret_cast<float4*>(mine + F_T_PART + lane * 8);
            pp[0] = make_float4(pt[0], pt[1], pt[2], pt[3]);
            pp[1] = make_float4(pt[4], pt[5], pt[6], pt[7]);
            float4* st = reinterpret_cast<float4*>(mine + lane * F_QLD);   // dqkv image [token][12] over the (dead) tiles
            st[0] = make_float4(dqkv[0], dqkv[1], dqkv[2], dqkv[3]);
            st[1] = make_float4(dqkv[4], dqkv[5], dqkv[6], dqkv[7]);
            st[2] = make_float4(dqkv[8], dqkv[9], dqkv[10], dqkv[11]);
        }
        STAMP(5);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            // dW_h[r][k] += sum_tokens dqkv[token][r] * [y1 | 1][token][k]: token 4 s + mg is the MFMA's k index of lane (mr, mg)
            const bool live = mr < 12;
            const float* const ap = mine + mg * F_QLD + mr;    // one base register each; the token step is an immediate offset
            const float* const bp = sY1 + mg * F_Y1LD + mr;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float av = ap[s * 4 * F_QLD], bv = bp[s * 4 * F_Y1LD];
                av = live ? av : 0.f;
                bv = live ? bv : 0.f;
                accq = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accq, 0, 0, 0);
            }
        }
        STAMP(6);
        lds_barrier();   // B2: both heads' partial sums and attention outputs of the window are in LDS
        STAMP(7);
        {
            // ---------------- E3: lane = (token tk, channel half ch): join the halves, LayerNorm backward, dx; proj weight gradient
            long bT, sT;
            const long pT = pixel_of(win, bT, sT);
            const float4 x0 = ex0, x1 = ex1, d0 = ed0, d1 = ed1;
            const uint32_t kw = a.keep ? ekw : 0xffffffffu;
            const float* const dgv = edg;
            const float* const o2v = eo2;
            const float2 mrs = *reinterpret_cast<const float2*>(sMr + 2 * tk);
            const float mu = mrs.x, rstd = mrs.y;
            const float4* pa = reinterpret_cast<const float4*>(smem + F_OFF_WAVE + (2 * ws) * F_PW + F_T_PART + tk * 8);
            const float4* pb = reinterpret_cast<const float4*>(smem + F_OFF_WAVE + (2 * ws + 1) * F_PW + F_T_PART + tk * 8);
            const float4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
            const float att[8] = {a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w, a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w};
            const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            const float dyv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            float xh[8], dyf[8], gg[8];
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                xh[u] = (xv[u] - mu) * rstd;
                dyf[u] = ch ? dgv[u] : att[u];
                gg[u] = dyf[u] * sLn[8 * ch + u];
                m1 += gg[u];
                m2 += gg[u] * xh[u];
            }
            m1 += dpp_xor1(m1);
            m2 += dpp_xor1(m2);
            m1 *= (1.0f / E);
            m2 *= (1.0f / E);
            float4* dxo = reinterpret_cast<float4*>(a.dx + pT * E + HC * ch);   // residual with the UNMASKED upstream gradient
            dxo[0] = make_float4(dyv[0] + rstd * (gg[0] - m1 - xh[0] * m2), dyv[1] + rstd * (gg[1] - m1 - xh[1] * m2),
                                 dyv[2] + rstd * (gg[2] - m1 - xh[2] * m2), dyv[3] + rstd * (gg[3] - m1 - xh[3] * m2));
            dxo[1] = make_float4(dyv[4] + rstd * (gg[4] - m1 - xh[4] * m2), dyv[5] + rstd * (gg[5] - m1 - xh[5] * m2),
                                 dyv[6] + rstd * (gg[6] - m1 - xh[6] * m2), dyv[7] + rstd * (gg[7] - m1 - xh[7] * m2));
            {   // d gamma / d beta of this lane's 8 channels: summed over the row's 8 tokens of this channel half, then into the row's slot
                float gs[16];
#pragma unroll
                for (int u = 0; u < 8; ++u) { gs[u] = row8_sum(dyf[u] * xh[u]); gs[8 + u] = row8_sum(dyf[u]); }
                float4* sl = reinterpret_cast<float4*>(myslot);
                float4 s0 = sl[0], s1 = sl[1], s2 = sl[2], s3 = sl[3];
                s0.x += gs[0]; s0.y += gs[1]; s0.z += gs[2]; s0.w += gs[3];
                s1.x += gs[4]; s1.y += gs[5]; s1.z += gs[6]; s1.w += gs[7];
                s2.x += gs[8]; s2.y += gs[9]; s2.z += gs[10]; s2.w += gs[11];
                s3.x += gs[12]; s3.y += gs[13]; s3.z += gs[14]; s3.w += gs[15];
                sl[0] = s0; sl[1] = s1; sl[2] = s2; sl[3] = s3;
            }
            // dym image [32 tokens of this wave][16] over the tiles (the dqkv image was consumed in front of B2), o2 into the cat image
            float dm[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) dm[u] = ((kw >> (8 * ch + u)) & 1u) ? (a.keep ? dyv[u] * (1.0f / 0.9f) : dyv[u]) : 0.0f;
            float4* dst = reinterpret_cast<float4*>(mine + (lane >> 1) * E + HC * ch);
            dst[0] = make_float4(dm[0], dm[1], dm[2], dm[3]);
            dst[1] = make_float4(dm[4], dm[5], dm[6], dm[7]);
            *reinterpret_cast<float4*>(sCat + tk * E + HC + 4 * ch) = make_float4(o2v[0], o2v[1], o2v[2], o2v[3]);
        }
        STAMP(8);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            // dWproj[n][k] += sum over this wave's 32 tokens of dym[token][n] * cat[token][k]
            const float* const ap = mine + mg * E + mr;
            const float* const bp = sCat + (32 * hd + mg) * E + mr;
#pragma unroll
            for (int s = 0; s < 8; ++s) accp = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[s * 4 * E], bp[s * 4 * E], accp, 0, 0, 0);
        }
        STAMP(9);
        lds_barrier();   // B3: tiles and images are free for the next window group
        STAMP(10);
    }

#ifdef LG_STAMPS
    { const int stamp_it = 0; STAMP(12); }
#endif
    // ---------------- write-out: one slab row per workgroup, summed over the workgroups by the deferred reduce launch.
    // Two barriers: every wave parks its 64 x 64 pos_emb-gradient columns in a region of its own (8 x 16.6 KB over the dead pos_emb, tiles
    // and images) next to its small sums, then all threads add the waves' shares in a fixed order on the way to global memory.
    // (Round 3's form -- the waves of a head taking turns at one LDS copy, a read-modify-write per value and a barrier per turn -- was
    // ~25 us of this kernel's 146.)
    float* const row = a.slab + (size_t)blockIdx.x * ATTN_BWD_F_ROW;
    constexpr int DPW = F_DPW, OLD = 65;            // floats of a wave's pos_emb-gradient region [i][65]
    float* const sRed = smem + F_NW * DPW;          // [waves][32] LayerNorm sums | [waves][256] accq | [waves][256] accp
    float v[16];   // lanes 0 / 1: the wave's d gamma (8) | d beta (8) sums of channel half 0 / 1 over its four rows
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = ((sSlot[(lane & 1) * 16 + i] + sSlot[(2 + (lane & 1)) * 16 + i]) + sSlot[(4 + (lane & 1)) * 16 + i]) + sSlot[(6 + (lane & 1)) * 16 + i];
    __syncthreads();   // every wave has read its slots: the per-wave regions are free
    {
        float* const dp = smem + wave * DPW + lane;
#pragma unroll
        for (int i = 0; i < 64; ++i) dp[i * OLD] = (i & 1) ? dpacc[i >> 1].y : dpacc[i >> 1].x;
        if (lane < 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { sRed[wave * 32 + 8 * lane + u] = v[u]; sRed[wave * 32 + 16 + 8 * lane + u] = v[8 + u]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // accumulator element i of lane (mr, mg) = D[4 mg + i][mr]
            sRed[32 * F_NW + wave * 256 + (4 * mg + i) * 16 + mr] = accq[i];
            sRed[32 * F_NW + 256 * F_NW + wave * 256 + (4 * mg + i) * 16 + mr] = accp[i];
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int idx = threadIdx.x; idx < 2 * 64 * 64; idx += F_NT) {   // pos_emb [h][i][j]: the waves of head h, window slots in order
        const int h = idx >> 12, o = ((idx >> 6) & 63) * OLD + (idx & 63);
        float r = smem[h * DPW + o];
#pragma unroll
        for (int s4 = 1; s4 < F_NS; ++s4) r += smem[(2 * s4 + h) * DPW + o];
        row[idx] = r;
    }
    for (int i = threadIdx.x; i < ATTN_BWD_F_ROW - ATTN_BWD_F_WQ; i += F_NT) {
        float r = 0.f;
        if (i < 192 + 24) {          // dWqkv [24][8] | dbqkv [24]: the four waves of the row's head
            const int rowq = i < 192 ? i >> 3 : i - 192, k = i < 192 ? (i & 7) : 8;
            const int third = rowq / HC, h = (rowq % HC) / D, c = rowq % D;
#pragma unroll
            for (int s4 = 0; s4 < F_NS; ++s4) r += sRed[32 * F_NW + (2 * s4 + h) * 256 + (third * 4 + c) * 16 + k];
        } else if (i < 192 + 24 + 256) {   // dWproj [16][16]
            const int j = i - 216;
#pragma unroll
            for (int w8 = 0; w8 < F_NW; ++w8) r += sRed[32 * F_NW + 256 * F_NW + w8 * 256 + j];
        } else {                     // d gamma [16] | d beta [16]
            const int j = i - 472;
#pragma unroll
            for (int w8 = 0; w8 < F_NW; ++w8) r += sRed[w8 * 32 + j];
        }
        row[ATTN_BWD_F_WQ + i] = r;
    }
#ifdef LG_STAMPS
    { const int stamp_it = 0; STAMP(13); }
#endif
}

// ------------------------------------------------------------------------------------------------
// proj backward towards the global-mixer half, in front of the FFT-mixer backward: do2[b,c,y,x] = sum_n projw[n][e/2+c] dym[p][n] with
// dym = dropout mask * dy.  Writes ONE keep-bit word per pixel (bit n = channel n kept) for k_attn_bwd_f instead of dym, and sums the
// proj bias gradient db[n] = sum_p dym[p][n] on the way (one partial row per workgroup).  One lane per pixel: the planar output wants it.
// ------------------------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(256) void k_proj_o2_bwd_k(ProjO2BwdKArgs a) {
    constexpr int HC = E / 2;
    __shared__ float sProj[E * HC];   // [n][c] = projw[n][HC + c]
    __shared__ float red[4][E];
    for (int i = threadIdx.x; i < E * HC; i += 256) sProj[i] = a.projw[(i / HC) * E + HC + (i % HC)];
    __syncthreads();
    float db[E];
#pragma unroll
    for (int n = 0; n < E; ++n) db[n] = 0.f;
    // the next pixel's row is requested before this one is worked on (a grid-stride loop of load -> hash -> matvec -> store trips is one
    // exposed round trip per trip)
    const long stride = (long)gridDim.x * 256L;
    long p = blockIdx.x * 256L + threadIdx.x;
    float4 nx[E / 4];
    {
        const float4* src = reinterpret_cast<const float4*>(a.dy + (p < a.total ? p : 0) * E);
#pragma unroll
        for (int k = 0; k < E / 4; ++k) nx[k] = src[k];
    }
    for (; p < a.total; p += stride) {
        const long b = p / a.HW, s = p - b * a.HW;
        float dy[E];
#pragma unroll
        for (int k = 0; k < E / 4; ++k) { dy[4 * k] = nx[k].x; dy[4 * k + 1] = nx[k].y; dy[4 * k + 2] = nx[k].z; dy[4 * k + 3] = nx[k].w; }
#ifndef LG_O2_PREF
#define LG_O2_PREF 1
#endif
        if (LG_O2_PREF) {
            const long pn = p + stride < a.total ? p + stride : p;
            const float4* src = reinterpret_cast<const float4*>(a.dy + pn * E);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) nx[k] = src[k];
        }
        if (a.keep) {
            uint32_t kw = 0;
#pragma unroll
            for (int n = 0; n < E; n += 2) {   // one hash per channel pair (common.h)
                float s0, s1;
                dropout_scale2(a.seed, (uint64_t)(p * E + n), s0, s1);
                kw |= (s0 != 0.0f ? (1u << n) : 0u) | (s1 != 0.0f ? (2u << n) : 0u);
                dy[n] *= s0; dy[n + 1] *= s1;
            }
            a.keep[p] = kw;
        }
#pragma unroll
        for (int n = 0; n < E; ++n) db[n] += dy[n];
#pragma unroll
        for (int c = 0; c < HC; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int n = 0; n < E; ++n) acc += sProj[n * HC + c] * dy[n];
            a.do2[(b * HC + c) * a.HW + s] = acc;
        }
        if (!LG_O2_PREF && p + stride < a.total) {
            const float4* src = reinterpret_cast<const float4*>(a.dy + (p + stride) * E);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) nx[k] = src[k];
        }
    }
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
#pragma unroll
    for (int n = 0; n < E; ++n) {
        float v = db[n];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (ln == 0) red[wv][n] = v;
    }
    __syncthreads();
    if (threadIdx.x < E) a.slab[(size_t)blockIdx.x * E + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

int attn_bwd_f_grid(int B, int h, int w) {
    const int nwin = B * (h / 8) * (w / 8);
    const int ngroups = (nwin + F_NS - 1) / F_NS;
    return ngroups < ATTN_BWD_F_WGS ? ngroups : ATTN_BWD_F_WGS;   // the resident workgroups: two per CU (4 waves, 80 KB of LDS each)
}

int launch_attn_bwd_f(int e, const AttnBwdFArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_ATTN_BWD, s);
    if (e != 16) { lg_set_error("attn_bwd_f: e=%d unsupported", e); return -1; }
    if ((a.h & 7) || (a.w & 7)) { lg_set_error("attn_bwd_f: h,w must be multiples of 8"); return -2; }
    if (!a.slab || !a.d_pos || !a.d_qkvw || !a.d_qkvb || !a.d_projw || !a.d_ln1g || !a.d_ln1b) { lg_set_error("attn_bwd_f: null destination"); return -2; }
    const int nwin = a.B * (a.h / 8) * (a.w / 8);
    if (nwin % F_NS) { lg_set_error("attn_bwd_f: %d windows are not a multiple of %d", nwin, F_NS); return -2; }   // level 0: h, w are multiples of 16
    const int ngroups = nwin / F_NS;
    const size_t lds = (size_t)F_LDS_FLOATS * sizeof(float);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t er = hipFuncSetAttribute((const void*)k_attn_bwd_f<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (er == hipSuccess) er = hipFuncSetAttribute((const void*)k_attn_bwd_f<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (er != hipSuccess) { lg_set_error("attn_bwd_f: hipFuncSetAttribute: %s", hipGetErrorString(er)); return (int)er; }
        attr_once.done();
    }
    const int grid = attn_bwd_f_grid(a.B, a.h, a.w);
    if (nwin >= (1 << 24) || a.w / 8 >= 256 || a.h / 8 >= 256) { lg_set_error("attn_bwd_f: %d windows of a %d x %d plane are out of range", nwin, a.h, a.w); return -2; }
    AttnBwdFArgs ak = a;
    ak.rcp_nwx = (unsigned)((1ull << 32) / (unsigned)(a.w / 8)) + 1u;   // floor(n / d) = umulhi(n, 2^32 / d + 1) for n d < 2^32
    ak.rcp_nwy = (unsigned)((1ull << 32) / (unsigned)(a.h / 8)) + 1u;
    if ((a.so == nullptr) != (a.sl == nullptr)) { lg_set_error("attn_bwd_f: the forward's row statistics come as a pair (so, sl)"); return -2; }
    if (a.so) k_attn_bwd_f<true><<<grid, F_NT, lds, s>>>(ak, nwin, ngroups);
    else k_attn_bwd_f<false><<<grid, F_NT, lds, s>>>(ak, nwin, ngroups);
    LG_CHECK_LAUNCH();
    // partial rows -> gradients (+=), in the block's deferred reduce launch
    ReduceJob j;
    j.dst2 = nullptr; j.nslices = grid; j.slice_stride = ATTN_BWD_F_ROW;
    auto job = [&](int off, float* dst, int rows, int cols) {
        j.slab = a.slab + off; j.dst = dst; j.rows = rows; j.cols = cols; j.row_stride = cols; j.ld = cols; j.rows_valid = rows; j.cols_valid = cols;
        return launch_reduce_job(j, s);
    };
    int rc = job(0, a.d_pos, 1, 2 * 64 * 64);
    if (!rc) rc = job(ATTN_BWD_F_WQ, a.d_qkvw, 3 * F_HC, F_HC);
    if (!rc) rc = job(ATTN_BWD_F_WQ + 192, a.d_qkvb, 1, 3 * F_HC);
    if (!rc) rc = job(ATTN_BWD_F_WQ + 216, a.d_projw, F_E, F_E);
    if (!rc) rc = job(ATTN_BWD_F_WQ + 472, a.d_ln1g, 1, F_E);
    if (!rc) rc = job(ATTN_BWD_F_WQ + 488, a.d_ln1b, 1, F_E);
    return rc;
}

int launch_proj_o2_bwd_k(int e, const ProjO2BwdKArgs& a, hipStream_t s) {
    if (e != 16) { lg_set_error("proj_o2_bwd_k: e=%d unsupported", e); return -1; }
    const long nb = (a.total + 255) / 256;
    const int grid = (int)(nb < PROJ_O2_K_WGS ? nb : PROJ_O2_K_WGS);
    k_proj_o2_bwd_k<16><<<grid, 256, 0, s>>>(a);
    LG_CHECK_LAUNCH();
    ReduceJob j;
    j.slab = a.slab; j.dst = a.d_projb; j.dst2 = nullptr; j.nslices = grid; j.slice_stride = e;
    j.rows = 1; j.cols = e; j.row_stride = e; j.ld = e; j.rows_valid = 1; j.cols_valid = e;
    return launch_reduce_job(j, s);
}
