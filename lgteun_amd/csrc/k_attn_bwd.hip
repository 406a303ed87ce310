// Backward of the local (window) mixer half-block (k_attn.hip) for gfx950 -- autograd of reference
// models/common/LGT.py:112-146 (local_mixer), 183-219 (LGMixer: proj / dropout / concat), 45-61 (pre_norm, residual).
//
// One wavefront = one 8x8 window, flash-attention-style recompute in two passes so that no cross-lane reduction
// is ever needed:
//   pass 1, lane = query i : scores, softmax stats, O_i, D_i = dO_i.O_i, dq_i = sum_j dS_ij k_j, and the pos_emb
//                            gradient row dS_i. (accumulated in LDS, lane i owns row i -> conflict-free ds_add_f32)
//   pass 2, lane = key j   : recomputes P_ij from the saved row stats, dv_j = sum_i P_ij dO_i, dk_j = sum_i dS_ij q_i
// dq/dk/dv go straight to the buffer the to_qkv weight-gradient GEMM reads (with cat and LN1(x)[:e/2], k_wgrad.hip); a
// per-pixel epilogue kernel pushes them through to_qkv^T, joins the FFT-mixer gradient for the other channel half and
// applies the LayerNorm backward + residual.  pos_emb sits in LDS once, rows padded to 65 floats so that both access
// patterns (lane = query, lane = key) are bank-conflict free; its gradient partials leave through a slab.
#include "kernels.h"
#include "bwd_kernels.h"

// STATS (round 6): the forward's saving launch left the log-sum-exp of every score row (log2 domain, a.sl) and the attention output (a.so): pass 1 is then ONE loop
// over the keys (scores -> P = 2^(s - L) -> dP -> dS -> dq) that reads k and v once; D_i = dO_i . O_i and the cat image come out of the prologue (k_attn_bwd_f.hip).
template <int HC, int NW, bool STATS>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(HC == 32 ? 1 : 2))) void k_attn_bwd_core(AttnBwdArgs a, int nwin, int ngroups) {
    // blockIdx.y = head: the two heads of a window are independent, so each workgroup keeps only one head's pos_emb /
    // dpos / K,V,Q,dO tiles in LDS -> half the LDS, twice the waves per CU.
    constexpr int E = 2 * HC, D = HC / 2;
    constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
    constexpr int Y1LD = (HC + 15) / 16 * 16, DQLD = (3 * HC + 15) / 16 * 16;  // wgrad operands are padded to 16 columns
    constexpr int PW = 4 * 64 * D + 64 * 4;       // floats of LDS per wave
    constexpr int PLD = 65;                       // padded pos_emb row: conflict-free for lane = query AND lane = key
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int hd = blockIdx.y;
    float* sPos = smem;                    // [64 i][65]  pos_emb[hd][i][j]
    float* sDpos = smem + 64 * PLD;        // [64 i][65]  accumulated dS
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // per-wave tiles, CHANNEL-major [c][token]: a 16-byte broadcast read is one channel of FOUR tokens, so both passes run their
    // inner products as packed fp32 over token pairs (v_pk_fma_f32, the lane's own operand splat by op_sel) -- k_attn.hip has the details
    float* sK = smem + 2 * 64 * PLD + wave * PW;   // [D][64]
    float* sV = sK + 64 * D;
    float* sQ = sV + 64 * D;               // q * scale * log2(e): scores live in the log2 domain (softmax = exp2(s - max))
    float* sDO = sQ + 64 * D;
    float* sSt = sDO + 64 * D;             // [3][64]  row max (log2 domain), 1/row sum, D_i
    // tokens per LDS broadcast: four (16 bytes) at D = 4; two (8 bytes) for the wider heads, whose register budget does not hold 4 x D
    // values of two operands per group
    constexpr int TG = D == 4 ? 4 : 2, NP = TG / 2, NG = 64 / TG;
    auto ld_tokens = [&](const float* tile, int c, int g, lg_v2f (&out)[NP]) {
        if constexpr (TG == 4) {
            const float4 v = reinterpret_cast<const float4*>(tile)[c * 16 + g];
            out[0] = (lg_v2f){v.x, v.y}; out[NP - 1] = (lg_v2f){v.z, v.w};
        } else {
            const float2 v = reinterpret_cast<const float2*>(tile)[c * 32 + g];
            out[0] = (lg_v2f){v.x, v.y};
        }
    };
    for (int i = threadIdx.x; i < 64 * 64; i += NW * 64) {
        const int ii = i >> 6, j = i & 63;
        sPos[ii * PLD + j] = a.pos[hd * 64 * 64 + i] * LOG2E;
        sDpos[ii * PLD + j] = 0.f;
    }
    // this head's slices of to_qkv (rows hd*D + c of each third) and of proj^T, once per workgroup: read as LDS broadcasts in the
    // window prologue (as ~44 dependent vector loads per window they were most of the prologue, 44 % of the kernel)
    __shared__ __attribute__((aligned(16))) float sWq[3 * D * HC];   // [q|k|v][c][k]
    __shared__ __attribute__((aligned(16))) float sWp[D * E];        // [k][n] = projw[n][hd*D + k]
    __shared__ float sBq[3 * D];
    __shared__ float sLn[2 * HC];   // LN1 gamma | beta of the local half
    if (threadIdx.x < HC) { sLn[threadIdx.x] = a.ln1g[threadIdx.x]; sLn[HC + threadIdx.x] = a.ln1b[threadIdx.x]; }
    for (int i = threadIdx.x; i < 3 * D * HC; i += NW * 64) {
        const int t3 = i / (D * HC), rem = i - t3 * (D * HC);
        sWq[i] = a.qkvw[(size_t)(t3 * HC + hd * D) * HC + rem];
    }
    for (int i = threadIdx.x; i < D * E; i += NW * 64) {
        const int k = i / E, n = i - k * E;
        sWp[i] = a.projw[n * E + hd * D + k];
    }
    if (threadIdx.x < 3 * D) sBq[threadIdx.x] = a.qkvb[(threadIdx.x / D) * HC + hd * D + threadIdx.x % D];
    const int nwx = a.w >> 3, nwy = a.h >> 3;
    const long hw = (long)a.h * a.w;
    const float scale = (float)(1.0 / sqrt((double)D));
    // pos_emb gradient: in pass 2 lane j owns column j of dS, so dpos[hd][i][j] accumulates in 64 registers across all the
    // windows of this wave (LDS float atomics here cost more than the rest of the kernel)
    lg_v2f dpacc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) dpacc[i] = (lg_v2f){0.f, 0.f};

    // pixel of this lane in window `win` (row-major windows, row-major tokens)
    auto pixel_of = [&](int win, long& b, long& s) -> long {
        const int wx = win % nwx;
        const int rr = win / nwx;
        const int wy = rr % nwy;
        b = rr / nwy;
        const int y = wy * 8 + (lane >> 3), x = wx * 8 + (lane & 7);
        s = (long)y * a.w + x;
        return b * hw + s;
    };
    __syncthreads();   // pos_emb and the weight slices are staged; inside the loop every wave only touches its own K / V / Q / dO tiles
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int win = grp * NW + wave;
        const bool active = win < nwin;
        long p = 0;
        __builtin_amdgcn_wave_barrier();
        if (active) {
            long b, s;
            p = pixel_of(win, b, s);
            // every global operand of the prologue is requested here, in one batch: x, dym and (head 0) the planar FFT-mixer half
            // used to be three dependent HBM round trips
            float4 xq4[E / 4], dq4[E / 4];
            float o2v[HC];
#pragma unroll
            for (int k = 0; k < E / 4; ++k) {
                xq4[k] = reinterpret_cast<const float4*>(a.x + p * E)[k];
                dq4[k] = reinterpret_cast<const float4*>(a.dym + p * E)[k];
            }
            if (hd == 0) {
#pragma unroll
                for (int c = 0; c < HC; ++c) o2v[c] = a.o2[(b * HC + c) * hw + s];
            }
            float4 Of[D / 4];
            float Lrow = 0.f;
            if constexpr (STATS) {
#pragma unroll
                for (int k = 0; k < D / 4; ++k) Of[k] = reinterpret_cast<const float4*>(a.so + p * HC + hd * D)[k];
                Lrow = a.sl[p * 2 + hd];
            }
            {
                float xv[E];
#pragma unroll
                for (int k = 0; k < E / 4; ++k) {
                    const float4 v = xq4[k];
                    xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
                }
                float mu, rstd;
                ln_stats<E>(xv, mu, rstd);
                float y1[HC];
#pragma unroll
                for (int c = 0; c < HC; ++c) y1[c] = (xv[c] - mu) * rstd * sLn[c] + sLn[HC + c];
                if (hd == 0 && a.y1) {   // to_qkv's conv input for its weight gradient (null when k_attn_bwd_epi accumulates that itself)
                    float4* y1o = reinterpret_cast<float4*>(a.y1 + p * Y1LD);
#pragma unroll
                    for (int k = 0; k < Y1LD / 4; ++k)
                        y1o[k] = (4 * k < HC) ? make_float4(y1[(4 * k) % HC], y1[(4 * k + 1) % HC], y1[(4 * k + 2) % HC], y1[(4 * k + 3) % HC])
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                // this head's q, k, v channels: rows hd*D + c of each third of to_qkv
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    float vq = 0.f, vk = 0.f, vv = 0.f;
#pragma unroll
                    for (int k = 0; k < HC; ++k) {
                        vq += sWq[c * HC + k] * y1[k];
                        vk += sWq[(D + c) * HC + k] * y1[k];
                        vv += sWq[(2 * D + c) * HC + k] * y1[k];
                    }
                    sQ[c * 64 + lane] = (vq + sBq[c]) * (scale * LOG2E);
                    sK[c * 64 + lane] = vk + sBq[D + c];
                    sV[c * 64 + lane] = vv + sBq[2 * D + c];
                }
            }
            {
                // dO = grad wrt this head's attention output channels = (proj^T dym)[hd*D : hd*D + D]
                float dym[E];
#pragma unroll
                for (int k = 0; k < E / 4; ++k) {
                    const float4 v = dq4[k];
                    dym[4 * k] = v.x; dym[4 * k + 1] = v.y; dym[4 * k + 2] = v.z; dym[4 * k + 3] = v.w;
                }
                float Dsum = 0.f;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    float acc = 0.f;
#pragma unroll
                    for (int n = 0; n < E; ++n) acc += sWp[k * E + n] * dym[n];
                    sDO[k * 64 + lane] = acc;
                    if constexpr (STATS) {
                        const float4 o4 = Of[k >> 2];
                        Dsum += acc * ((k & 3) == 0 ? o4.x : (k & 3) == 1 ? o4.y : (k & 3) == 2 ? o4.z : o4.w);
                    }
                }
                if constexpr (STATS) {   // row statistics of (token = lane, head hd) for both passes; the head's attention output into the proj-input image
                    sSt[lane] = Lrow;
                    sSt[128 + lane] = Dsum;
                    float4* co = reinterpret_cast<float4*>(a.cat + p * E + hd * D);
#pragma unroll
                    for (int c4 = 0; c4 < D / 4; ++c4) co[c4] = Of[c4];
                }
                if (hd == 0) {   // global-mixer half of the proj input, and the zero padding of the dqkv rows
                    float* co = a.cat + p * E;
#pragma unroll
                    for (int c = 0; c < HC; ++c) co[HC + c] = o2v[c];
                    if (DQLD > 3 * HC) {
                        float* pad = a.dqkv + p * DQLD + 3 * HC;
#pragma unroll
                        for (int c = 0; c < DQLD - 3 * HC; ++c) pad[c] = 0.f;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // per-wave tiles: no workgroup barrier needed
        if (active) {
            // ---------------- pass 1: lane = query i, packed over KEY pairs
            float q[D], dOi[D];
#pragma unroll
            for (int c = 0; c < D; ++c) { q[c] = sQ[c * 64 + lane]; dOi[c] = sDO[c * 64 + lane]; }
            const float* prow = sPos + lane * PLD;
            lg_v2f dq2[D];
#pragma unroll
            for (int c = 0; c < D; ++c) dq2[c] = (lg_v2f){0.f, 0.f};
            float O[D], mx = 0.f, inv = 1.f, Dv = 0.f;
            if constexpr (STATS) {
                const float Li = sSt[lane];
                Dv = sSt[128 + lane];
                const lg_v2f L2v = (lg_v2f){Li, Li}, Dv2 = (lg_v2f){Dv, Dv};
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    lg_v2f sp[NP], dP[NP], kv[D][NP];
#pragma unroll
                    for (int u = 0; u < NP; ++u) { sp[u] = (lg_v2f){prow[TG * g + 2 * u], prow[TG * g + 2 * u + 1]}; dP[u] = (lg_v2f){0.f, 0.f}; }
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        lg_v2f vv[NP];
                        ld_tokens(sK, c, g, kv[c]);
                        ld_tokens(sV, c, g, vv);
                        const lg_v2f qq = (lg_v2f){q[c], q[c]}, dd = (lg_v2f){dOi[c], dOi[c]};
#pragma unroll
                        for (int u = 0; u < NP; ++u) { sp[u] = qq * kv[c][u] + sp[u]; dP[u] = dd * vv[u] + dP[u]; }
                    }
#pragma unroll
                    for (int u = 0; u < NP; ++u) {
                        const lg_v2f e = sp[u] - L2v;
                        const lg_v2f P = (lg_v2f){__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
                        const lg_v2f dS = P * (dP[u] - Dv2);
#pragma unroll
                        for (int c = 0; c < D; ++c) dq2[c] = dS * kv[c][u] + dq2[c];
                    }
                    if ((TG * (g + 1)) % 8 == 0) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            lg_v2f sc[32];
            mx = -3.0e38f;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                lg_v2f sp[NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) sp[u] = (lg_v2f){prow[TG * g + 2 * u], prow[TG * g + 2 * u + 1]};
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    lg_v2f kv[NP];
                    ld_tokens(sK, c, g, kv);
                    const lg_v2f qq = (lg_v2f){q[c], q[c]};
#pragma unroll
                    for (int u = 0; u < NP; ++u) sp[u] = qq * kv[u] + sp[u];
                }
#pragma unroll
                for (int u = 0; u < NP; ++u) { sc[NP * g + u] = sp[u]; mx = fmaxf(mx, fmaxf(sp[u].x, sp[u].y)); }
                if ((TG * (g + 1)) % 8 == 0) __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("" ::: "memory");
            lg_v2f l2 = (lg_v2f){0.f, 0.f};
            const lg_v2f mx2 = (lg_v2f){mx, mx};
#pragma unroll
            for (int g = 0; g < 32; ++g) {
                const lg_v2f t = sc[g] - mx2;
                sc[g] = (lg_v2f){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                l2 += sc[g];
            }
            inv = __builtin_amdgcn_rcpf(l2.x + l2.y);
            const lg_v2f inv2 = (lg_v2f){inv, inv};
            lg_v2f O2[D];   // (even keys, odd keys) partials
#pragma unroll
            for (int c = 0; c < D; ++c) O2[c] = (lg_v2f){0.f, 0.f};
            // D_i = sum_j P_ij dP_ij = dO_i . O_i  (dP_ij = dO_i . v_j): no second pass over V for it
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int u = 0; u < NP; ++u) sc[NP * g + u] *= inv2;
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    lg_v2f vv[NP];
                    ld_tokens(sV, c, g, vv);
#pragma unroll
                    for (int u = 0; u < NP; ++u) O2[c] = sc[NP * g + u] * vv[u] + O2[c];
                }
                if ((TG * (g + 1)) % 8 == 0) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int c = 0; c < D; ++c) { O[c] = O2[c].x + O2[c].y; Dv += dOi[c] * O[c]; }
            asm volatile("" ::: "memory");   // re-read K / V from LDS below instead of keeping 64 x 2D values live
            const lg_v2f Dv2 = (lg_v2f){Dv, Dv};
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                lg_v2f dP[NP], kv[D][NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) dP[u] = (lg_v2f){0.f, 0.f};
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    lg_v2f vv[NP];
                    ld_tokens(sV, c, g, vv);
                    ld_tokens(sK, c, g, kv[c]);
                    const lg_v2f dd = (lg_v2f){dOi[c], dOi[c]};
#pragma unroll
                    for (int u = 0; u < NP; ++u) dP[u] = dd * vv[u] + dP[u];
                }
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const lg_v2f dS = sc[NP * g + u] * (dP[u] - Dv2);
#pragma unroll
                    for (int c = 0; c < D; ++c) dq2[c] = dS * kv[c][u] + dq2[c];
                }
                if ((TG * (g + 1)) % 8 == 0) __builtin_amdgcn_sched_barrier(0);
            }
            }
            float dqh[D];
#pragma unroll
            for (int c = 0; c < D; ++c) dqh[c] = dq2[c].x + dq2[c].y;
            float4* co = reinterpret_cast<float4*>(a.cat + p * E + hd * D);      // 16-byte stores (as scalar loops the two possibly
            float4* dq_o = reinterpret_cast<float4*>(a.dqkv + p * DQLD + hd * D);  // aliasing destinations compiled to 2 D dword stores)
#pragma unroll
            for (int c4 = 0; c4 < D / 4; ++c4) {
                if constexpr (!STATS) co[c4] = make_float4(O[4 * c4], O[4 * c4 + 1], O[4 * c4 + 2], O[4 * c4 + 3]);
                dq_o[c4] = make_float4(dqh[4 * c4] * scale, dqh[4 * c4 + 1] * scale, dqh[4 * c4 + 2] * scale, dqh[4 * c4 + 3] * scale);
            }
            if constexpr (!STATS) {
                sSt[lane] = mx;
                sSt[64 + lane] = inv;
                sSt[128 + lane] = Dv;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // per-wave tiles: no workgroup barrier needed
        if (active) {
            // ---------------- pass 2: lane = key j, packed over QUERY pairs
            float kj[D], vj[D];
            lg_v2f dk2[D], dv2[D];
#pragma unroll
            for (int c = 0; c < D; ++c) { kj[c] = sK[c * 64 + lane]; vj[c] = sV[c * 64 + lane]; dk2[c] = (lg_v2f){0.f, 0.f}; dv2[c] = (lg_v2f){0.f, 0.f}; }
            const float* pcol = sPos + lane;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                lg_v2f t[NP], dP[NP], qi[D][NP], doi[D][NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) { t[u] = (lg_v2f){pcol[(TG * g + 2 * u) * PLD], pcol[(TG * g + 2 * u + 1) * PLD]}; dP[u] = (lg_v2f){0.f, 0.f}; }
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    ld_tokens(sQ, c, g, qi[c]);
                    ld_tokens(sDO, c, g, doi[c]);
                    const lg_v2f kk = (lg_v2f){kj[c], kj[c]}, vv = (lg_v2f){vj[c], vj[c]};
#pragma unroll
                    for (int u = 0; u < NP; ++u) { t[u] = kk * qi[c][u] + t[u]; dP[u] = vv * doi[c][u] + dP[u]; }
                }
                lg_v2f smx[NP], sinv[NP], sdv[NP];   // row max (log2 domain), 1 / row sum, D_i of the TG queries
                ld_tokens(sSt, 0, g, smx);
                if constexpr (!STATS) ld_tokens(sSt, 1, g, sinv);
                ld_tokens(sSt, 2, g, sdv);
                lg_v2f P[NP], dS[NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const lg_v2f e = t[u] - smx[u];
                    P[u] = (lg_v2f){__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
                    if constexpr (!STATS) P[u] *= sinv[u];   // (STATS: smx holds the row's log-sum-exp)
                    dS[u] = P[u] * (dP[u] - sdv[u]);
                    dpacc[NP * g + u] += dS[u];
                }
                if ((TG * (g + 1)) % 8 == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < D; ++c) {
#pragma unroll
                    for (int u = 0; u < NP; ++u) { dv2[c] = P[u] * doi[c][u] + dv2[c]; dk2[c] = dS[u] * qi[c][u] + dk2[c]; }
                }
            }
            float dkh[D], dvh[D];
#pragma unroll
            for (int c = 0; c < D; ++c) { dkh[c] = (dk2[c].x + dk2[c].y) * LN2; dvh[c] = dv2[c].x + dv2[c].y; }   // sQ carries log2(e)
            float4* dk_o = reinterpret_cast<float4*>(a.dqkv + p * DQLD + HC + hd * D);
            float4* dv_o = reinterpret_cast<float4*>(a.dqkv + p * DQLD + 2 * HC + hd * D);
#pragma unroll
            for (int c4 = 0; c4 < D / 4; ++c4) {
                dk_o[c4] = make_float4(dkh[4 * c4], dkh[4 * c4 + 1], dkh[4 * c4 + 2], dkh[4 * c4 + 3]);
                dv_o[c4] = make_float4(dvh[4 * c4], dvh[4 * c4 + 1], dvh[4 * c4 + 2], dvh[4 * c4 + 3]);
            }
        }
    }
    __syncthreads();
    for (int w = 0; w < NW; ++w) {   // waves take turns (fixed order): sDpos[i][j] += this wave's column sums
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < 64; ++i) sDpos[i * PLD + lane] += (i & 1) ? dpacc[i >> 1].y : dpacc[i >> 1].x;
        }
        __syncthreads();
    }
    // pos_emb partial of this workgroup -> slab[blockIdx.x][hd][i][j]
    float* slab = a.dpos_slab + ((size_t)blockIdx.x * 2 + hd) * 64 * 64;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += NW * 64) slab[idx] = sDpos[(idx >> 6) * PLD + (idx & 63)];
}

// per-pixel epilogue, one lane per pixel (e = 64; the narrower blocks use k_attn_bwd_epi below): dqkv -> to_qkv^T, join the FFT-mixer gradient, LayerNorm-1 backward + residual, LN1 param grads
template <int E>
__global__ __launch_bounds__(256) void k_attn_bwd_epi_px(AttnBwdArgs a, long total) {
    constexpr int HC = E / 2, DQLD = (3 * HC + 15) / 16 * 16;
    __shared__ float red[4 * 2 * E];
    float pl[2 * E];
#pragma unroll
    for (int i = 0; i < 2 * E; ++i) pl[i] = 0.f;
    for (long p = blockIdx.x * 256L + threadIdx.x; p < total; p += (long)gridDim.x * 256L) {
        const long hw = (long)a.h * a.w;
        const long b = p / hw, s = p - b * hw;
        float dqkv[3 * HC];
        const float4* dq4 = reinterpret_cast<const float4*>(a.dqkv + p * DQLD);
#pragma unroll
        for (int k = 0; k < 3 * HC / 4; ++k) {
            float4 v = dq4[k];
            dqkv[4 * k] = v.x; dqkv[4 * k + 1] = v.y; dqkv[4 * k + 2] = v.z; dqkv[4 * k + 3] = v.w;
        }
        float dyf[E];
#pragma unroll
        for (int k = 0; k < HC; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 3 * HC; ++c) acc += a.qkvw[c * HC + k] * dqkv[c];
            dyf[k] = acc;
        }
#pragma unroll
        for (int c = 0; c < HC; ++c) dyf[HC + c] = a.dg[(b * HC + c) * hw + s];
        float xv[E];
        const float4* src = reinterpret_cast<const float4*>(a.x + p * E);
#pragma unroll
        for (int k = 0; k < E / 4; ++k) {
            float4 v = src[k];
            xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
        }
        float mu, rstd;
        ln_stats<E>(xv, mu, rstd);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int c = 0; c < E; ++c) {
            const float xh = (xv[c] - mu) * rstd;
            pl[c] += dyf[c] * xh;
            pl[E + c] += dyf[c];
            dyf[c] *= a.ln1g[c];
            m1 += dyf[c];
            m2 += dyf[c] * xh;
            xv[c] = xh;
        }
        m1 *= (1.0f / E);
        m2 *= (1.0f / E);
        // residual with the UNMASKED upstream gradient
        const float4* dys = reinterpret_cast<const float4*>(a.dy + p * E);
        float4* dxo = reinterpret_cast<float4*>(a.dx + p * E);
#pragma unroll
        for (int k = 0; k < E / 4; ++k) {
            float4 dv4 = dys[k];
            float o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = rstd * (dyf[4 * k + u] - m1 - xv[4 * k + u] * m2);
            dxo[k] = make_float4(dv4.x + o[0], dv4.y + o[1], dv4.z + o[2], dv4.w + o[3]);
        }
    }
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 2 * E; ++i) {
        float v = pl[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (ln == 0) red[wv * 2 * E + i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 2 * E) {
        const float v = red[threadIdx.x] + red[2 * E + threadIdx.x] + red[4 * E + threadIdx.x] + red[6 * E + threadIdx.x];
        // partial rows [grid][E] d gamma | [grid][E] d beta, summed by launch_reduce_slab_pair
        if (threadIdx.x < E) a.part[blockIdx.x * (size_t)E + threadIdx.x] = v;
        else a.part[(size_t)gridDim.x * E + blockIdx.x * (size_t)E + threadIdx.x - E] = v;
    }
}

// Per-pixel epilogue in the lane = (pixel, channel quad) layout (e = 16, 32): dqkv -> to_qkv^T, join the FFT-mixer gradient, LayerNorm-1
// backward + residual, LN1 parameter gradients.  The E/4 lanes of a pixel own four channels each, so x / dy / dx move as contiguous 1 KB
// rows per wave; lane q also owns rows 6q .. 6q+5 of dqkv: it forms their share of to_qkv^T dqkv for every attention channel (the pixel's
// lanes add up with DPP) and -- FQ, e = 16 -- accumulates those rows of the to_qkv weight gradient dW[c][k] = sum_p dqkv[p][c] y1[p][k]
// and its bias gradient right here (y1 = LN1(x)[:e/2] is on chip), so the core kernel no longer writes y1 and the separate
// weight-gradient launch with its second pass over dqkv is gone.
#define ATTN_EPI_WGS 1024
template <int E, bool FQ>
__global__ __launch_bounds__(256) void k_attn_bwd_epi(AttnBwdArgs a, long total) {
    constexpr int HC = E / 2, DQLD = (3 * HC + 15) / 16 * 16, LPP = E / 4, PPW = 256 / LPP, R = 3 * HC / LPP, LDX = HC + 4;
    static_assert(R == 6, "six dqkv rows per lane");
    constexpr int NW_ = FQ ? R * HC + R : 0, NACC = 8 + NW_;
    __shared__ float red[4 * LPP * NACC];
    __shared__ __attribute__((aligned(16))) float ex[FQ ? PPW * LDX : 4];
    const int q = threadIdx.x % LPP, slot = threadIdx.x / LPP, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool attn_half = q < LPP / 2;
    float wr[R][HC], g1[4], b1[4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < HC; ++k) wr[r][k] = a.qkvw[(R * q + r) * HC + k];
#pragma unroll
    for (int u = 0; u < 4; ++u) { g1[u] = a.ln1g[4 * q + u]; b1[u] = a.ln1b[4 * q + u]; }
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    const long hw = (long)a.h * a.w;
    // The operands of a pixel group are requested ONE GROUP AHEAD into a second, statically named register set (A / B alternate: a
    // rotating set would need its moves in front of the back edge, i.e. a wait for the loads issued in the same iteration): the kernel
    // runs two waves per SIMD, too few to hide an HBM round trip per iteration by occupancy.  Every load is unconditional (the planar
    // dg channel is clamped for the attention-half lanes, whose value is not used).
    struct Ops4 { float4 x4, dy4; float2 dq2[R / 2]; float dgv[4]; };
    auto issue = [&](long p, Ops4& o) {
        const long b = p / hw, s = p - b * hw;
        o.x4 = *reinterpret_cast<const float4*>(a.x + p * E + 4 * q);
        o.dy4 = *reinterpret_cast<const float4*>(a.dy + p * E + 4 * q);
#pragma unroll
        for (int r2 = 0; r2 < R / 2; ++r2) o.dq2[r2] = *reinterpret_cast<const float2*>(a.dqkv + p * DQLD + R * q + 2 * r2);
        const int cg = attn_half ? 0 : 4 * q - HC;
#pragma unroll
        for (int u = 0; u < 4; ++u) o.dgv[u] = a.dg[(b * HC + cg + u) * hw + s];
    };
    const long stride = (long)gridDim.x * PPW;
    auto body = [&](long p, const Ops4& o) {
        const float4 x4 = o.x4, dy4 = o.dy4;
        float dq[R];
#pragma unroll
        for (int r2 = 0; r2 < R / 2; ++r2) { dq[2 * r2] = o.dq2[r2].x; dq[2 * r2 + 1] = o.dq2[r2].y; }
        float dgv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) dgv[u] = attn_half ? 0.f : o.dgv[u];
        // LayerNorm statistics over the pixel's E channels
        const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
        const float mu = lane_group_sum<LPP>((xv[0] + xv[1]) + (xv[2] + xv[3])) * (1.0f / E);
        float xh[4], vs = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) { xh[u] = xv[u] - mu; vs += xh[u] * xh[u]; }
        const float rstd = __builtin_amdgcn_rsqf(lane_group_sum<LPP>(vs) * (1.0f / E) + LG_EPS);
#pragma unroll
        for (int u = 0; u < 4; ++u) xh[u] *= rstd;
        // to_qkv^T dqkv: this lane's six rows for every attention channel, summed over the pixel's lanes
        float dyf[4] = {dgv[0], dgv[1], dgv[2], dgv[3]};
#pragma unroll
        for (int k = 0; k < HC; ++k) {
            float pt = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) pt += wr[r][k] * dq[r];
            const float full = lane_group_sum<LPP>(pt);
            if (q == k / 4) dyf[k & 3] = full;     // k < HC: only attention-half lanes match
        }
        if constexpr (FQ) {
            if (attn_half) *reinterpret_cast<float4*>(ex + slot * LDX + 4 * q) =
                make_float4(xh[0] * g1[0] + b1[0], xh[1] * g1[1] + b1[1], xh[2] * g1[2] + b1[2], xh[3] * g1[3] + b1[3]);
        }
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[u] += dyf[u] * xh[u];       // d gamma
            acc[4 + u] += dyf[u];           // d beta
            dyf[u] *= g1[u];
            m1 += dyf[u];
            m2 += dyf[u] * xh[u];
        }
        m1 = lane_group_sum<LPP>(m1) * (1.0f / E);
        m2 = lane_group_sum<LPP>(m2) * (1.0f / E);
        // residual with the UNMASKED upstream gradient
        *reinterpret_cast<float4*>(a.dx + p * E + 4 * q) =
            make_float4(dy4.x + rstd * (dyf[0] - m1 - xh[0] * m2), dy4.y + rstd * (dyf[1] - m1 - xh[1] * m2),
                        dy4.z + rstd * (dyf[2] - m1 - xh[2] * m2), dy4.w + rstd * (dyf[3] - m1 - xh[3] * m2));
        if constexpr (FQ) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();     // the lanes of a pixel sit in one wave
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float yv[HC];
#pragma unroll
            for (int k4 = 0; k4 < HC / 4; ++k4) {
                const float4 t = *reinterpret_cast<const float4*>(ex + slot * LDX + 4 * k4);
                yv[4 * k4] = t.x; yv[4 * k4 + 1] = t.y; yv[4 * k4 + 2] = t.z; yv[4 * k4 + 3] = t.w;
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int k = 0; k < HC; ++k) acc[8 + r * HC + k] += dq[r] * yv[k];
                acc[8 + R * HC + r] += dq[r];
            }
            __builtin_amdgcn_wave_barrier();     // ex is rewritten by the next pixel group
        }
    };
    {
        Ops4 oa, ob;
        long p = (long)blockIdx.x * PPW + slot;          // total is a multiple of PPW (launcher): a group is whole or absent
        if (p < total) issue(p, oa);
        while (p < total) {
            const long p1 = p + stride, p2 = p1 + stride;
            issue(p1 < total ? p1 : p, ob);             // clamped: a repeat of the current group when there is no next one
            body(p, oa);
            if (p1 >= total) break;
            issue(p2 < total ? p2 : p1, oa);
            body(p1, ob);
            p = p2;
        }
    }
    // lanes with equal q hold partials of the same outputs: across the wave, then the 4 waves in LDS (fixed order)
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = LPP; off < 64; off <<= 1) v += __shfl_xor(v, off);
        if (lane < LPP) red[(wave * LPP + q) * NACC + i] = v;
    }
    __syncthreads();
    // partial rows of this workgroup: [grid][E] d gamma | [grid][E] d beta | (FQ) [grid][3HC * HC] dWqkv | [grid][3HC] dbqkv
    const size_t nwg = gridDim.x, wg = blockIdx.x;
    for (int i = threadIdx.x; i < LPP * NACC; i += 256) {
        const float v = (red[i] + red[LPP * NACC + i]) + (red[2 * LPP * NACC + i] + red[3 * LPP * NACC + i]);
        const int qq = i / NACC, k = i - qq * NACC;
        if (k < 4) a.part[wg * E + 4 * qq + k] = v;
        else if (k < 8) a.part[nwg * E + wg * E + 4 * qq + (k - 4)] = v;
        else if (k < 8 + R * HC) a.part[2 * nwg * E + wg * (3 * HC * HC) + (R * qq) * HC + (k - 8)] = v;
        else a.part[2 * nwg * E + nwg * (3 * HC * HC) + wg * (3 * HC) + R * qq + (k - 8 - R * HC)] = v;
    }
}

template <int HC, int NW>
static int grid_t(int B, int h, int w) {
    int nwin = B * (h / 8) * (w / 8);
    int ngroups = (nwin + NW - 1) / NW;
    // one resident workgroup per (CU, head): 256 VGPRs allow 2 waves per SIMD = 8 waves per CU anyway, and every extra round of
    // workgroups pays the pos_emb / weight staging and the dpos write-out again (grid 512 -> 128 at NW = 8: 250 -> 218 us)
    constexpr int cap = 1024 / NW;
    return ngroups < cap ? ngroups : cap;
}
int attn_bwd_grid(int e, int B, int h, int w) {
    if (e == 16) return grid_t<8, 8>(B, h, w);
    if (e == 32) return grid_t<16, 4>(B, h, w);
    return grid_t<32, 4>(B, h, w);
}

template <int HC, int NW>
static int launch_attn_bwd_t(const AttnBwdArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_ATTN_BWD, s);
    int nwin = a.B * (a.h / 8) * (a.w / 8);
    int ngroups = (nwin + NW - 1) / NW;
    size_t lds = (size_t)(2 * 64 * 65 + NW * (4 * 64 * (HC / 2) + 64 * 4)) * sizeof(float);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_bwd_core<HC, NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_attn_bwd_core<HC, NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
        if (e != hipSuccess) { lg_set_error("attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    int grid = grid_t<HC, NW>(a.B, a.h, a.w);
    if (HC == 16 && a.core_m) {      // e = 32, A/B: the matrix-pipe core (k_attn_bwd_m.hip; same arguments, outputs and launch shape)
        int rc = launch_attn_bwd_core_m(2 * HC, a, grid, nwin, ngroups, s);
        if (rc) return rc;
    } else {
        if ((a.so == nullptr) != (a.sl == nullptr)) { lg_set_error("attn_bwd: the forward's row statistics come as a pair (so, sl)"); return -2; }
        if (a.so) k_attn_bwd_core<HC, NW, true><<<dim3(grid, 2), NW * 64, lds, s>>>(a, nwin, ngroups);
        else k_attn_bwd_core<HC, NW, false><<<dim3(grid, 2), NW * 64, lds, s>>>(a, nwin, ngroups);
        LG_CHECK_LAUNCH();
    }
    const long total = (long)a.B * a.h * a.w;
    if (!a.part) { lg_set_error("attn_bwd: partial-sum scratch missing"); return -2; }
    constexpr int E = 2 * HC;
    if constexpr (E <= 32) {
        constexpr int PPW = 256 / (E / 4);
        constexpr bool FQ = attn_bwd_fuses_qkv(E);
        if (total % PPW) { lg_set_error("attn_bwd: %ld pixels are not a multiple of %d", total, PPW); return -2; }
        if (FQ && (!a.d_qkvw || !a.d_qkvb)) { lg_set_error("attn_bwd: to_qkv gradient destinations missing"); return -2; }
        const long ng = total / PPW;
        const int egrid = (int)(ng < ATTN_EPI_WGS ? ng : ATTN_EPI_WGS);
        k_attn_bwd_epi<E, FQ><<<egrid, 256, 0, s>>>(a, total);
        LG_CHECK_LAUNCH();
        const size_t g = (size_t)egrid;
        int rc = launch_reduce_slab_pair(a.part, a.part + g * E, egrid, E, a.d_ln1g, a.d_ln1b, s);
        if (rc || !FQ) return rc;
        const float* pw = a.part + 2 * g * E;
        return launch_reduce_slab_wb(pw, pw + g * (3 * HC * HC), egrid, 3 * HC, HC, a.d_qkvw, HC, a.d_qkvb, s);
    } else {
        const long nb = (total + 255) / 256;
        const int egrid = (int)(nb < PIXEL_PART_WGS ? nb : PIXEL_PART_WGS);
        k_attn_bwd_epi_px<E><<<egrid, 256, 0, s>>>(a, total);
        LG_CHECK_LAUNCH();
        return launch_reduce_slab_pair(a.part, a.part + (size_t)egrid * E, egrid, E, a.d_ln1g, a.d_ln1b, s);
    }
}

int launch_attn_bwd(int e, const AttnBwdArgs& a, hipStream_t s) {
    if ((a.h & 7) || (a.w & 7)) { lg_set_error("attn_bwd: h,w must be multiples of 8"); return -2; }
    if (e == 16) return launch_attn_bwd_t<8, 8>(a, s);
    if (e == 32) return launch_attn_bwd_t<16, 4>(a, s);
    if (e == 64) return launch_attn_bwd_t<32, 4>(a, s);
    lg_set_error("attn_bwd: e=%d unsupported", e);
    return -1;
}

// ------------------------------------------------------------------------------------------------
// proj backward towards the global-mixer half: do2[b,c,y,x] = sum_n projw[n][e/2+c] * dy[p][n] * mask
// ------------------------------------------------------------------------------------------------
// (one lane per pixel: a lane = (pixel, channel quad) form measured no faster at e = 16 and slower at e = 32 -- the planar output wants a
// lane per pixel and the per-channel lane-group sums cost more than the coalesced dy / dym rows gain)
template <int E>
__global__ __launch_bounds__(256) void k_proj_o2_bwd(ProjO2BwdArgs a) {
    constexpr int HC = E / 2;
    __shared__ float sProj[E * HC];   // [n][c] = projw[n][HC + c]
    for (int i = threadIdx.x; i < E * HC; i += 256) sProj[i] = a.projw[(i / HC) * E + HC + (i % HC)];
    __syncthreads();
    long p = blockIdx.x * 256L + threadIdx.x;
    if (p >= a.total) return;
    long b = p / a.HW, s = p - b * a.HW;
    float dy[E];
    const float4* src = reinterpret_cast<const float4*>(a.dy + p * E);
#pragma unroll
    for (int k = 0; k < E / 4; ++k) {
        float4 v = src[k];
        dy[4 * k] = v.x; dy[4 * k + 1] = v.y; dy[4 * k + 2] = v.z; dy[4 * k + 3] = v.w;
    }
    if (a.dropout) {
#pragma unroll
        for (int n = 0; n < E; ++n) dy[n] *= dropout_scale(a.seed, (uint64_t)(p * E + n));
        float4* dst = reinterpret_cast<float4*>(a.dym + p * E);
#pragma unroll
        for (int k = 0; k < E / 4; ++k) dst[k] = make_float4(dy[4 * k], dy[4 * k + 1], dy[4 * k + 2], dy[4 * k + 3]);
    }
#pragma unroll
    for (int c = 0; c < HC; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int n = 0; n < E; ++n) acc += sProj[n * HC + c] * dy[n];
        a.do2[(b * HC + c) * a.HW + s] = acc;
    }
}

int launch_proj_o2_bwd(int e, const ProjO2BwdArgs& a, hipStream_t s) {
    int grid = (int)((a.total + 255) / 256);
    if (e == 16) k_proj_o2_bwd<16><<<grid, 256, 0, s>>>(a);
    else if (e == 32) k_proj_o2_bwd<32><<<grid, 256, 0, s>>>(a);
    else if (e == 64) k_proj_o2_bwd<64><<<grid, 256, 0, s>>>(a);
    else { lg_set_error("proj_o2_bwd: e=%d unsupported", e); return -1; }
    LG_CHECK_LAUNCH();
    return 0;
}
