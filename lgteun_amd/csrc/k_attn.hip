// Local (window) mixer + LGMixer projection + residual for gfx950.
// Reference: models/common/LGT.py:112-146 (local_mixer), 183-219 (LGMixer), 45-61,231-248 (pre_norm/residual).
//
// One wavefront = one 8x8 window: lane i is token i (64 tokens = 64 lanes, the CDNA wave width).  Each lane
// LayerNorms its own pixel, makes its q/k/v (1x1 conv = per-pixel matvec with wave-uniform weights), parks k/v
// in LDS, then walks the 64 keys: scores, softmax and the A.V sum stay in the lane's registers (no score
// tensor ever reaches memory; the reference materialises B*nW*2*64*64 floats per block).  The same lane then
// concatenates the global-mixer output of its pixel, applies proj (+dropout) and the residual.
// A workgroup is 4 waves = 4 horizontally adjacent windows and loops over window quads (pos_emb^T stays in LDS).
#include "kernels.h"

template <int HC>
__global__ __launch_bounds__(256) void k_attn(AttnArgs a, int nwin, int nquads) {
    constexpr int E = 2 * HC, D = HC / 2;
    constexpr float LOG2E = 1.44269504088896340736f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sPos = smem;                       // [2][64][64]  posT[h][j][i]
    float* sK = smem + 2 * 64 * 64;           // [4][64][HC]
    float* sV = sK + 4 * 64 * HC;             // [4][64][HC]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // weights of the block, once per (persistent) workgroup: read as LDS broadcasts in the window loop (as dependent vector
    // loads they were ~60 load instructions with waits per window)
    __shared__ __attribute__((aligned(16))) float sWqkv[3 * HC * HC];
    __shared__ __attribute__((aligned(16))) float sWproj[E * E];
    __shared__ float sBias[3 * HC + E + 2 * HC];   // qkv bias | proj bias | ln1 gamma, beta (local half)
    {
        // Staging: EVERY value of a thread is requested before the first one is stored (unconditional loads from clamped indices).  As five
        // loops with a load, a wait and a store per trip -- the small ones inside exec-masked branches, whose joins wait with vmcnt(0) -- the
        // workgroups of the launch, which all start together, sat through ~8 dependent L2 round trips before their first window (round 4)
        constexpr int NQ = (3 * HC * HC + 255) / 256, NPJ = (E * E + 255) / 256, NB = 3 * HC + E + 2 * HC;
        float pv[32], wq[NQ], wp[NPJ], bv[4];
#pragma unroll
        for (int k = 0; k < 32; ++k) pv[k] = a.posT[k * 256 + threadIdx.x];
#pragma unroll
        for (int k = 0; k < NQ; ++k) wq[k] = a.qkvw[min(k * 256 + (int)threadIdx.x, 3 * HC * HC - 1)];
#pragma unroll
        for (int k = 0; k < NPJ; ++k) wp[k] = a.projw[min(k * 256 + (int)threadIdx.x, E * E - 1)];
        {
            const int i = threadIdx.x;
            bv[0] = a.qkvb[min(i, 3 * HC - 1)];
            bv[1] = a.projb[clampi(i - 3 * HC, 0, E - 1)];
            bv[2] = a.ln1g[clampi(i - 3 * HC - E, 0, HC - 1)];
            bv[3] = a.ln1b[clampi(i - 4 * HC - E, 0, HC - 1)];
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) sPos[k * 256 + threadIdx.x] = pv[k] * LOG2E;   // scores live in the log2 domain: softmax = exp2(s - max)
#pragma unroll
        for (int k = 0; k < NQ; ++k) if (k * 256 + (int)threadIdx.x < 3 * HC * HC) sWqkv[k * 256 + threadIdx.x] = wq[k];
#pragma unroll
        for (int k = 0; k < NPJ; ++k) if (k * 256 + (int)threadIdx.x < E * E) sWproj[k * 256 + threadIdx.x] = wp[k];
        static_assert(NB <= 256, "one bias value per thread");
        if ((int)threadIdx.x < NB) {
            const int i = threadIdx.x;
            sBias[i] = i < 3 * HC ? bv[0] : (i < 3 * HC + E ? bv[1] : (i < 4 * HC + E ? bv[2] : bv[3]));
        }
    }
    const int nwx = a.w >> 3, nwy = a.h >> 3;
    const float scale = (float)(1.0 / sqrt((double)D)) * LOG2E;
    float* myK = sK + wave * 64 * HC;
    float* myV = sV + wave * 64 * HC;
    // pixel of this lane in window `win` (row-major windows, row-major tokens)
    auto pixel_of = [&](int win, long& b, int& y, int& x) -> long {
        const int wx = win % nwx;
        const int r = win / nwx;
        const int wy = r % nwy;
        b = r / nwy;
        y = wy * 8 + (lane >> 3);
        x = wx * 8 + (lane & 7);
        return (b * a.h + y) * (long)a.w + x;
    };
    // software prefetch: the x row of the NEXT quad's window is requested while the current window is processed.  Only where the
    // registers allow it: at HC = 32 the prefetched row, the residual row and o2 (160 registers) pushed the kernel into AGPR spills.
    constexpr bool PRE = HC <= 16;
    float4 xpre[PRE ? E / 4 : 1];
    if (PRE) {
        const int win0 = blockIdx.x * 4 + wave;
        if (blockIdx.x < nquads && win0 < nwin) {
            long b0; int y0, x0;
            const long p0 = pixel_of(win0, b0, y0, x0);
#pragma unroll
            for (int k = 0; k < E / 4; ++k) xpre[k] = reinterpret_cast<const float4*>(a.x + p0 * E)[k];
        }
    }
    for (int quad = blockIdx.x; quad < nquads; quad += gridDim.x) {
        const int win = quad * 4 + wave;
        const bool active = win < nwin;
        long p = 0, b = 0;
        int y = 0, x = 0;
        float q[HC];
        __syncthreads();  // previous iteration's readers of sK/sV are done; sPos is loaded
        if (active) {
            p = pixel_of(win, b, y, x);
            float xv[E];
            if (PRE) {
#pragma unroll
                for (int k = 0; k < E / 4; ++k) { xv[4 * k] = xpre[k].x; xv[4 * k + 1] = xpre[k].y; xv[4 * k + 2] = xpre[k].z; xv[4 * k + 3] = xpre[k].w; }
            } else {
#pragma unroll
                for (int k = 0; k < E / 4; ++k) {
                    const float4 t = reinterpret_cast<const float4*>(a.x + p * E)[k];
                    xv[4 * k] = t.x; xv[4 * k + 1] = t.y; xv[4 * k + 2] = t.z; xv[4 * k + 3] = t.w;
                }
            }
            if (PRE) {
                const int wnext = (quad + (int)gridDim.x) * 4 + wave;
                if (quad + (int)gridDim.x < nquads && wnext < nwin) {
                    long bn; int yn, xn;
                    const long pn = pixel_of(wnext, bn, yn, xn);
#pragma unroll
                    for (int k = 0; k < E / 4; ++k) xpre[k] = reinterpret_cast<const float4*>(a.x + pn * E)[k];
                }
            }
            float mu, rstd;
            ln_stats<E>(xv, mu, rstd);
            float y1[HC];
#pragma unroll
            for (int c = 0; c < HC; ++c) y1[c] = (xv[c] - mu) * rstd * sBias[3 * HC + E + c] + sBias[4 * HC + E + c];
            // to_qkv: rows [0,HC) q, [HC,2HC) k, [2HC,3HC) v   (LGT.py:136 chunk order)
#pragma unroll
            for (int c = 0; c < HC; ++c) {
                float vq = 0.f, vk = 0.f, vv = 0.f;
#pragma unroll
                for (int k = 0; k < HC; ++k) {
                    vq += sWqkv[c * HC + k] * y1[k];
                    vk += sWqkv[(HC + c) * HC + k] * y1[k];
                    vv += sWqkv[(2 * HC + c) * HC + k] * y1[k];
                }
                q[c] = (vq + sBias[c]) * scale;
                myK[c * 64 + lane] = vk + sBias[HC + c];   // channel-major [c][token]: a 16-byte broadcast read = one channel of FOUR keys
                myV[c * 64 + lane] = vv + sBias[2 * HC + c];
            }
        }
        __syncthreads();
        if (active) {
            // operands of the epilogue (FFT-mixer half, residual x) are requested now so their HBM latency hides under the softmax
            float o2[HC];
            const long hw = (long)a.h * a.w;
            const long s = (long)y * a.w + x;
            float4 xres[PRE ? E / 4 : 1];
            if (PRE) {
#pragma unroll
                for (int c = 0; c < HC; ++c) o2[c] = a.o2[(b * HC + c) * hw + s];
#pragma unroll
                for (int k = 0; k < E / 4; ++k) xres[k] = reinterpret_cast<const float4*>(a.x + p * E)[k];
            }
            // dropout keep-bits of this pixel's E outputs, drawn here (one register across the softmax) rather than in the epilogue, where
            // the hash temporaries pushed the kernel over the three-waves-per-SIMD register budget
            uint32_t keep[(E + 31) / 32];
#pragma unroll
            for (int n = 0; n < (E + 31) / 32; ++n) keep[n] = 0xffffffffu;
            if (a.dropout) {
#pragma unroll
                for (int n = 0; n < E; ++n)
                    if (dropout_scale(a.seed, (uint64_t)(p * E + n)) == 0.0f) keep[n >> 5] &= ~(1u << (n & 31));
            }
            float o1[HC];
#pragma unroll
            for (int hd = 0; hd < 2; ++hd) {
                // Packed fp32 (v_pk_fma_f32 / v_pk_add_f32): two KEYS per instruction.  Channel c of keys 4g .. 4g+3 is one 16-byte LDS
                // broadcast; the scores of a key pair sit in an aligned register pair, q_c is splat by op_sel; the A.V sum keeps an
                // (even keys, odd keys) partial per channel.  Per key: 2 + 2 packed FMAs, half a subtract, half an add, one exp2.
                const float4* kh = reinterpret_cast<const float4*>(myK + hd * D * 64);
                const float4* vh = reinterpret_cast<const float4*>(myV + hd * D * 64);
                const float* ph = sPos + hd * 64 * 64 + lane;
                lg_v2f sc[32];
                float mx = -3.0e38f;
                // The reads of key group g land in one of two register sets by the parity of g (written as an explicit load step: the
                // scheduler then keeps the next group's reads apart from this group's arithmetic -- 49.3 -> 47.5 us against reading at the
                // point of use; a fully hand-pipelined form with sched_barriers needs 184 registers = two waves per SIMD: 52.9 us)
                float4 kq[2][D];
                lg_v2f pq[2][2];
                auto load_k = [&](int g, int b) {
                    pq[b][0] = (lg_v2f){ph[(4 * g) * 64], ph[(4 * g + 1) * 64]};
                    pq[b][1] = (lg_v2f){ph[(4 * g + 2) * 64], ph[(4 * g + 3) * 64]};
#pragma unroll
                    for (int c = 0; c < D; ++c) kq[b][c] = kh[c * 16 + g];
                };
                load_k(0, 0);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    if (g) load_k(g, g & 1);
                    lg_v2f s01 = pq[g & 1][0], s23 = pq[g & 1][1];
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        const float4 kv = kq[g & 1][c];
                        const lg_v2f qq = (lg_v2f){q[hd * D + c], q[hd * D + c]};
                        s01 = qq * (lg_v2f){kv.x, kv.y} + s01;
                        s23 = qq * (lg_v2f){kv.z, kv.w} + s23;
                    }
                    sc[2 * g] = s01; sc[2 * g + 1] = s23;
                    mx = fmaxf(mx, fmaxf(fmaxf(s01.x, s01.y), fmaxf(s23.x, s23.y)));
                }
                lg_v2f l2 = (lg_v2f){0.f, 0.f};
                lg_v2f acc2[D];
#pragma unroll
                for (int c = 0; c < D; ++c) acc2[c] = (lg_v2f){0.f, 0.f};
                const lg_v2f mx2 = (lg_v2f){mx, mx};
                auto load_v = [&](int g, int b) {
#pragma unroll
                    for (int c = 0; c < D; ++c) kq[b][c] = vh[c * 16 + g];
                };
                load_v(0, 0);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    if (g) load_v(g, g & 1);
                    const lg_v2f a01 = sc[2 * g] - mx2, a23 = sc[2 * g + 1] - mx2;
                    const lg_v2f p01 = (lg_v2f){__builtin_amdgcn_exp2f(a01.x), __builtin_amdgcn_exp2f(a01.y)};
                    const lg_v2f p23 = (lg_v2f){__builtin_amdgcn_exp2f(a23.x), __builtin_amdgcn_exp2f(a23.y)};
                    l2 += p01; l2 += p23;
#pragma unroll
                    for (int c = 0; c < D; ++c) {
                        const float4 vv = kq[g & 1][c];
                        acc2[c] = p01 * (lg_v2f){vv.x, vv.y} + acc2[c];
                        acc2[c] = p23 * (lg_v2f){vv.z, vv.w} + acc2[c];
                    }
                }
                const float l = l2.x + l2.y;
                float acc[D];
#pragma unroll
                for (int c = 0; c < D; ++c) acc[c] = acc2[c].x + acc2[c].y;
                const float inv = __builtin_amdgcn_rcpf(l);
#pragma unroll
                for (int c = 0; c < D; ++c) o1[hd * D + c] = acc[c] * inv;
            }
            if (!PRE) {
#pragma unroll
                for (int c = 0; c < HC; ++c) o2[c] = a.o2[(b * HC + c) * hw + s];
            }
            // proj (E x E), dropout, residual -- in chunks of 4 outputs
            float4* yo = reinterpret_cast<float4*>(a.y + p * E);
#pragma unroll
            for (int n4 = 0; n4 < E / 4; ++n4) {
                float o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int n = n4 * 4 + u;
                    float v = 0.f;
#pragma unroll
                    for (int k = 0; k < HC; ++k) v += sWproj[n * E + k] * o1[k];
#pragma unroll
                    for (int k = 0; k < HC; ++k) v += sWproj[n * E + HC + k] * o2[k];
                    v += sBias[3 * HC + n];
                    v = (keep[n >> 5] >> (n & 31)) & 1u ? (a.dropout ? v * (1.0f / 0.9f) : v) : 0.0f;
                    o[u] = v;
                }
                const float4 xr = PRE ? xres[n4] : reinterpret_cast<const float4*>(a.x + p * E)[n4];
                yo[n4] = make_float4(xr.x + o[0], xr.y + o[1], xr.z + o[2], xr.w + o[3]);
            }
        }
    }
}

#include "workspace.h"
#define LG_MAX_POS_JOBS (5 * LG_MAX_K)
struct PosTArgs { const float* src[LG_MAX_POS_JOBS]; float* dst[LG_MAX_POS_JOBS]; };
__global__ void k_pos_transpose(PosTArgs a) {
    // pos [2][64][64] (h,i,j) -> posT [2][64][64] (h,j,i); blockIdx.y = block of the stage
    const float* __restrict__ pos = a.src[blockIdx.y];
    float* __restrict__ posT = a.dst[blockIdx.y];
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * 64 * 64) {
        int h = i >> 12, r = (i >> 6) & 63, c = i & 63;
        posT[(h * 64 + c) * 64 + r] = pos[i];
    }
}
int launch_pos_transpose_n(int n, const float* const* pos, float* const* posT, hipStream_t s) {
    if (n < 1 || n > LG_MAX_POS_JOBS) { lg_set_error("pos_transpose: n=%d", n); return -2; }
    PosTArgs a;
    for (int j = 0; j < LG_MAX_POS_JOBS; ++j) { a.src[j] = pos[j < n ? j : 0]; a.dst[j] = posT[j < n ? j : 0]; }
    k_pos_transpose<<<dim3(32, n), 256, 0, s>>>(a);
    LG_CHECK_LAUNCH();
    return 0;
}
int launch_pos_transpose(const float* pos, float* posT, hipStream_t s) { return launch_pos_transpose_n(1, &pos, &posT, s); }

template <int HC>
static int launch_attn_t(const AttnArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_ATTN, s);
    int nwin = a.B * (a.h / 8) * (a.w / 8);
    int nquads = (nwin + 3) / 4;
    size_t lds = (2 * 64 * 64 + 2 * 4 * 64 * HC) * sizeof(float);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn<HC>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) { lg_set_error("attn: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    // persistent grid = the workgroups that are resident at once (LDS: 50 KB / 72 KB / 125 KB per workgroup -> 3 / 2 / 1 per CU;
    // the VGPR counts allow the same), each walking its window quads with pos_emb^T in LDS.  A larger grid runs in rounds and the
    // last round is ragged (HC = 16 at C = 8, 2048 quads: 768 workgroups on 512 slots took 6 quad-times instead of 4).
    constexpr int cap = 256 * (HC <= 8 ? 3 : (HC <= 16 ? 2 : 1));
    const int rounds = (nquads + cap - 1) / cap;
    int grid = nquads < cap ? nquads : (nquads + rounds - 1) / rounds;
    k_attn<HC><<<grid, 256, lds, s>>>(a, nwin, nquads);
    LG_CHECK_LAUNCH();
    return 0;
}

int launch_attn(int e, const AttnArgs& a, hipStream_t s) {
    if ((a.h & 7) || (a.w & 7)) { lg_set_error("attn: h,w must be multiples of 8"); return -2; }
    if (e == 16) return launch_attn_t<8>(a, s);
    if (e == 32) return launch_attn_t<16>(a, s);
    if (e == 64) return launch_attn_t<32>(a, s);
    lg_set_error("attn: e=%d unsupported", e);
    return -1;
}
