// Backward of the local (window) mixer half-block (k_attn.hip) for gfx950 -- autograd of reference
// models/common/LGT.py:112-146 (local_mixer), 183-219 (LGMixer: proj / dropout / concat), 45-61 (pre_norm, residual).
//
// One wavefront = one 8x8 window, flash-attention-style recompute in two passes so that no cross-lane reduction
// is ever needed:
//   pass 1, lane = query i : scores, softmax stats, O_i, D_i = dO_i.O_i, dq_i = sum_j dS_ij k_j, and the pos_emb
//                            gradient row dS_i. (accumulated in LDS, lane i owns row i -> conflict-free ds_add_f32)
//   pass 2, lane = key j   : recomputes P_ij from the saved row stats, dv_j = sum_i P_ij dO_i, dk_j = sum_i dS_ij q_i
// then, back on its own pixel, the lane pushes dq/dk/dv through to_qkv^T, joins the FFT-mixer gradient for the other
// channel half and applies the LayerNorm backward + residual.  Operands of the proj / to_qkv weight-gradient GEMMs
// (cat, LN1(x)[:e/2], dqkv) are written out for k_wgrad.hip.  pos_emb partials leave through a slab (fixed-order sum).
#include "kernels.h"
#include "bwd_kernels.h"

template <int HC, int NW>
__global__ __launch_bounds__(NW * 64) void k_attn_bwd(AttnBwdArgs a, int nwin, int ngroups) {
    constexpr int E = 2 * HC, D = HC / 2;
    constexpr int Y1LD = (HC + 15) / 16 * 16, DQLD = (3 * HC + 15) / 16 * 16;  // wgrad operands are padded to 16 columns
    constexpr int PW = 4 * 64 * HC + 2 * 64 * 3;  // floats of LDS per wave
    extern __shared__ float smem[];
    float* sDpos = smem;  // [2][64(j)][64(i)]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* sK = smem + 2 * 64 * 64 + wave * PW;
    float* sV = sK + 64 * HC;
    float* sQ = sV + 64 * HC;
    float* sDO = sQ + 64 * HC;
    float* sSt = sDO + 64 * HC;  // [2][64][3]
    for (int i = threadIdx.x; i < 2 * 64 * 64; i += NW * 64) sDpos[i] = 0.f;
    const int nwx = a.w >> 3, nwy = a.h >> 3;
    const long hw = (long)a.h * a.w;
    const float scale = (float)(1.0 / sqrt((double)D));
    float dgam[E], dbet[E];
#pragma unroll
    for (int c = 0; c < E; ++c) { dgam[c] = 0.f; dbet[c] = 0.f; }

    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int win = grp * NW + wave;
        const bool active = win < nwin;
        long p = 0, b = 0, s = 0;
        float kreg[HC], vreg[HC], dO[HC], o1[HC], dq[HC];
        float mu = 0.f, rstd = 0.f;
        __syncthreads();
        if (active) {
            const int wx = win % nwx;
            const int rr = win / nwx;
            const int wy = rr % nwy;
            b = rr / nwy;
            const int y = wy * 8 + (lane >> 3), x = wx * 8 + (lane & 7);
            s = (long)y * a.w + x;
            p = b * hw + s;
            float xv[E];
            const float4* src = reinterpret_cast<const float4*>(a.x + p * E);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) {
                float4 v = src[k];
                xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
            }
            ln_stats<E>(xv, mu, rstd);
            float y1[HC];
#pragma unroll
            for (int c = 0; c < HC; ++c) y1[c] = (xv[c] - mu) * rstd * a.ln1g[c] + a.ln1b[c];
            float4* y1o = reinterpret_cast<float4*>(a.y1 + p * Y1LD);
#pragma unroll
            for (int k = 0; k < Y1LD / 4; ++k)
                y1o[k] = (4 * k < HC) ? make_float4(y1[(4 * k) % HC], y1[(4 * k + 1) % HC], y1[(4 * k + 2) % HC], y1[(4 * k + 3) % HC])
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < HC; ++c) {
                float vq = 0.f, vk = 0.f, vv = 0.f;
#pragma unroll
                for (int k = 0; k < HC; ++k) {
                    vq += a.qkvw[c * HC + k] * y1[k];
                    vk += a.qkvw[(HC + c) * HC + k] * y1[k];
                    vv += a.qkvw[(2 * HC + c) * HC + k] * y1[k];
                }
                kreg[c] = vk + a.qkvb[HC + c];
                vreg[c] = vv + a.qkvb[2 * HC + c];
                sQ[lane * HC + c] = (vq + a.qkvb[c]) * scale;
                sK[lane * HC + c] = kreg[c];
                sV[lane * HC + c] = vreg[c];
            }
            // dO = grad wrt the attention output = (proj^T dym)[:HC]
            float dym[E];
            const float4* ds = reinterpret_cast<const float4*>(a.dym + p * E);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) {
                float4 v = ds[k];
                dym[4 * k] = v.x; dym[4 * k + 1] = v.y; dym[4 * k + 2] = v.z; dym[4 * k + 3] = v.w;
            }
#pragma unroll
            for (int k = 0; k < HC; ++k) {
                float acc = 0.f;
#pragma unroll
                for (int n = 0; n < E; ++n) acc += a.projw[n * E + k] * dym[n];
                dO[k] = acc;
                sDO[lane * HC + k] = acc;
            }
        }
        __syncthreads();
        if (active) {
            // ---------------- pass 1: lane = query
#pragma unroll
            for (int hd = 0; hd < 2; ++hd) {
                float q[D];
#pragma unroll
                for (int c = 0; c < D; ++c) q[c] = sQ[lane * HC + hd * D + c];
                float sc[64];
                float mx = -3.0e38f;
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    float t = 0.f;
#pragma unroll
                    for (int c = 0; c < D; ++c) t += q[c] * sK[j * HC + hd * D + c];
                    t += a.posT[(hd * 64 + j) * 64 + lane];
                    sc[j] = t;
                    mx = fmaxf(mx, t);
                }
                float l = 0.f;
#pragma unroll
                for (int j = 0; j < 64; ++j) { sc[j] = expf(sc[j] - mx); l += sc[j]; }
                const float inv = 1.0f / l;
                float O[D];
#pragma unroll
                for (int c = 0; c < D; ++c) O[c] = 0.f;
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    sc[j] *= inv;
#pragma unroll
                    for (int c = 0; c < D; ++c) O[c] += sc[j] * sV[j * HC + hd * D + c];
                }
                // D_i = sum_j P_ij dP_ij, summed the way softmax-backward does (not as dO.O: rounding of near-zero
                // pos_emb gradients follows the reference more closely)
                float Dv = 0.f;
#pragma unroll
                for (int c = 0; c < D; ++c) o1[hd * D + c] = O[c];
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    float dP = 0.f;
#pragma unroll
                    for (int c = 0; c < D; ++c) dP += dO[hd * D + c] * sV[j * HC + hd * D + c];
                    Dv += sc[j] * dP;
                }
                float dqh[D];
#pragma unroll
                for (int c = 0; c < D; ++c) dqh[c] = 0.f;
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    float dP = 0.f;
#pragma unroll
                    for (int c = 0; c < D; ++c) dP += dO[hd * D + c] * sV[j * HC + hd * D + c];
                    const float dS = sc[j] * (dP - Dv);
#pragma unroll
                    for (int c = 0; c < D; ++c) dqh[c] += dS * sK[j * HC + hd * D + c];
                    atomicAdd(&sDpos[(hd * 64 + j) * 64 + lane], dS);
                }
#pragma unroll
                for (int c = 0; c < D; ++c) dq[hd * D + c] = dqh[c] * scale;
                sSt[(hd * 64 + lane) * 3 + 0] = mx;
                sSt[(hd * 64 + lane) * 3 + 1] = inv;
                sSt[(hd * 64 + lane) * 3 + 2] = Dv;
            }
        }
        __syncthreads();
        if (active) {
            // ---------------- pass 2: lane = key
            float dk[HC], dv[HC];
#pragma unroll
            for (int hd = 0; hd < 2; ++hd) {
                float dkh[D], dvh[D];
#pragma unroll
                for (int c = 0; c < D; ++c) { dkh[c] = 0.f; dvh[c] = 0.f; }
#pragma unroll 8
                for (int i = 0; i < 64; ++i) {
                    float t = 0.f, dP = 0.f;
                    float qi[D], doi[D];
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        qi[c] = sQ[i * HC + hd * D + c];
                        doi[c] = sDO[i * HC + hd * D + c];
                        t += qi[c] * kreg[hd * D + c];
                        dP += doi[c] * vreg[hd * D + c];
                    }
                    t += a.pos[(hd * 64 + i) * 64 + lane];
                    const float P = expf(t - sSt[(hd * 64 + i) * 3 + 0]) * sSt[(hd * 64 + i) * 3 + 1];
                    const float dS = P * (dP - sSt[(hd * 64 + i) * 3 + 2]);
#pragma unroll
                    for (int c = 0; c < D; ++c) { dvh[c] += P * doi[c]; dkh[c] += dS * qi[c]; }
                }
#pragma unroll
                for (int c = 0; c < D; ++c) { dk[hd * D + c] = dkh[c]; dv[hd * D + c] = dvh[c]; }
            }
            // ---------------- back on the lane's own pixel
            float* dq_o = a.dqkv + p * DQLD;
#pragma unroll
            for (int c = 0; c < HC; ++c) { dq_o[c] = dq[c]; dq_o[HC + c] = dk[c]; dq_o[2 * HC + c] = dv[c]; }
#pragma unroll
            for (int c = 3 * HC; c < DQLD; ++c) dq_o[c] = 0.f;
            float dyf[E];
#pragma unroll
            for (int k = 0; k < HC; ++k) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < HC; ++c)
                    acc += a.qkvw[c * HC + k] * dq[c] + a.qkvw[(HC + c) * HC + k] * dk[c] + a.qkvw[(2 * HC + c) * HC + k] * dv[c];
                dyf[k] = acc;
            }
#pragma unroll
            for (int c = 0; c < HC; ++c) dyf[HC + c] = a.dg[(b * HC + c) * hw + s];
            // LayerNorm backward (stats recomputed above), residual with the UNMASKED upstream gradient
            float xv[E];
            const float4* src = reinterpret_cast<const float4*>(a.x + p * E);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) {
                float4 v = src[k];
                xv[4 * k] = v.x; xv[4 * k + 1] = v.y; xv[4 * k + 2] = v.z; xv[4 * k + 3] = v.w;
            }
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int c = 0; c < E; ++c) {
                const float xh = (xv[c] - mu) * rstd;
                dgam[c] += dyf[c] * xh;
                dbet[c] += dyf[c];
                dyf[c] *= a.ln1g[c];
                m1 += dyf[c];
                m2 += dyf[c] * xh;
                xv[c] = xh;
            }
            m1 *= (1.0f / E);
            m2 *= (1.0f / E);
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
            // proj input for its weight gradient
            float* co = a.cat + p * E;
#pragma unroll
            for (int c = 0; c < HC; ++c) { co[c] = o1[c]; co[HC + c] = a.o2[(b * HC + c) * hw + s]; }
        }
    }
    __syncthreads();
    // pos_emb partial of this workgroup -> slab[blockIdx.x][h][i][j]
    float* slab = a.dpos_slab + (size_t)blockIdx.x * 2 * 64 * 64;
    for (int idx = threadIdx.x; idx < 2 * 64 * 64; idx += NW * 64) {
        const int h = idx >> 12, i = (idx >> 6) & 63, j = idx & 63;
        slab[idx] = sDpos[(h * 64 + j) * 64 + i];
    }
    // LayerNorm-1 parameter partials: wave reduction, one atomic per wave and parameter
#pragma unroll
    for (int c = 0; c < E; ++c) {
        float g1 = dgam[c], b1 = dbet[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { g1 += __shfl_xor(g1, off); b1 += __shfl_xor(b1, off); }
        if (lane == 0) { atomicAdd(a.d_ln1g + c, g1); atomicAdd(a.d_ln1b + c, b1); }
    }
}

template <int HC, int NW>
static int grid_t(int B, int h, int w) {
    int nwin = B * (h / 8) * (w / 8);
    int ngroups = (nwin + NW - 1) / NW;
    return ngroups < 512 ? ngroups : 512;
}
int attn_bwd_grid(int e, int B, int h, int w) {
    if (e == 16) return grid_t<8, 8>(B, h, w);
    if (e == 32) return grid_t<16, 4>(B, h, w);
    return grid_t<32, 2>(B, h, w);
}

template <int HC, int NW>
static int launch_attn_bwd_t(const AttnBwdArgs& a, hipStream_t s) {
    int nwin = a.B * (a.h / 8) * (a.w / 8);
    int ngroups = (nwin + NW - 1) / NW;
    size_t lds = (size_t)(2 * 64 * 64 + NW * (4 * 64 * HC + 2 * 64 * 3)) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_bwd<HC, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) { lg_set_error("attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_done = true;
    }
    int grid = grid_t<HC, NW>(a.B, a.h, a.w);
    k_attn_bwd<HC, NW><<<grid, NW * 64, lds, s>>>(a, nwin, ngroups);
    LG_CHECK_LAUNCH();
    return 0;
}

int launch_attn_bwd(int e, const AttnBwdArgs& a, hipStream_t s) {
    if ((a.h & 7) || (a.w & 7)) { lg_set_error("attn_bwd: h,w must be multiples of 8"); return -2; }
    if (e == 16) return launch_attn_bwd_t<8, 8>(a, s);
    if (e == 32) return launch_attn_bwd_t<16, 4>(a, s);
    if (e == 64) return launch_attn_bwd_t<32, 2>(a, s);
    lg_set_error("attn_bwd: e=%d unsupported", e);
    return -1;
}

// ------------------------------------------------------------------------------------------------
// proj backward towards the global-mixer half: do2[b,c,y,x] = sum_n projw[n][e/2+c] * dy[p][n] * mask
// ------------------------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(256) void k_proj_o2_bwd(ProjO2BwdArgs a) {
    constexpr int HC = E / 2;
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
        for (int n = 0; n < E; ++n) acc += a.projw[n * E + HC + c] * dy[n];
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
