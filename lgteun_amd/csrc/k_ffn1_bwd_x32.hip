// k_ffn1_bwd_x32: backward of the first half of feed_forward (reference models/common/LGT.py:96-98 + pre_norm / residual :45-61) at
// e = 32 (hidden width 128: level 1 of the 4-band net, level 0 of the 8-band net), same contract as k_ffn1_bwd<32> (k_ffn_bwd.hip):
//     dh1 = (dh2 W2) * gelu'(h1) ;  dx = dy + LN2^T(dh1 W1) ;  dW1 += dh1^T LN2(x) ;  db1 += sum dh1 ;  LN2 parameter gradients
// The kernel it replaces ran one wave per SIMD on v_mfma_f32_16x16x4_f32 with W2^T streamed from L2 in the GEMM loop (463 us per
// launch at 32 x 128 x 128 pixels, 36 % of the fp32 matrix peak and nothing in flight from HBM meanwhile).  Here, as in k_ffn_x32:
//   * the two pixel GEMMs run on the bf16 matrix pipe in the fp32-equivalent three-piece form (split_bf16.h), weights on the A side:
//     wave w of 8 owns rows [16 w, 16 w + 16) of W2^T (GEMM1, K = 128) and the (16-channel block w & 1, pixel block w >> 1) tile of
//     GEMM2 (W1^T, K = 128); all its weight fragments are PRE-SPLIT (k_split_w) and register-resident (96 VGPRs);
//   * a tile is 64 pixels; its dh2 / gelu'(h1) / x / dy rows are requested one tile ahead into registers (the only vector-memory
//     loads of the loop, so the in-order s_waitcnt never waits on anything but them);
//   * LDS 154 KB: dh2 pieces [3][64][128] bf16 | dh1 pieces [3][64][128] bf16 (16-byte chunks XOR-swizzled by the pixel index) |
//     dh1 fp32 per wave [8][64][20] (A operand of the dW1 tiles: pixels are that GEMM's K, it stays on v_mfma_f32_16x16x4_f32 --
//     64 pixels per tile, 32 instructions per wave) | LN2(x) [64][36] | d LN2-out [64][36].
#include "kernels.h"
#include "bwd_kernels.h"

#include "hstore.h"
#include "split_bf16.h"

namespace {

constexpr int E = 32, N1 = 128, TP = 64, LDT = 20, LDY = 36;
constexpr int PIECE = TP * N1;                       // halves per piece plane
constexpr size_t OFF_D1 = (size_t)3 * PIECE * 2;     // bytes
constexpr size_t OFF_T = 2 * OFF_D1;
constexpr size_t OFF_Y = OFF_T + (size_t)8 * TP * LDT * 4;
constexpr size_t OFF_O = OFF_Y + (size_t)TP * LDY * 4;
constexpr size_t LDS_BYTES = OFF_O + (size_t)TP * LDY * 4;
constexpr int NF_W1 = 8, NF_W2 = 32;                 // fragment counts of k_split_w's first two matrices at e = 32

__device__ __forceinline__ bf16x8_t lds_x8(const uint16_t* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(p)); }

template <int NP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ffn1_bwd_x32(Ffn1BwdArgs a, const u32x4_t* __restrict__ wsp, long ntiles) {
    constexpr bool BF = (NP == 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint16_t* DH = reinterpret_cast<uint16_t*>(smem_raw);
    uint16_t* D1 = reinterpret_cast<uint16_t*>(smem_raw + OFF_D1);
    float* Tw = reinterpret_cast<float*>(smem_raw + OFF_T);
    float* Y = reinterpret_cast<float*>(smem_raw + OFF_Y);
    float* O = reinterpret_cast<float*>(smem_raw + OFF_O);
    __shared__ __attribute__((aligned(16))) float lnp[2 * E];
    __shared__ float red[8 * 2 * E];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    if (threadIdx.x < E) { lnp[threadIdx.x] = a.ln2g[threadIdx.x]; lnp[E + threadIdx.x] = a.ln2b[threadIdx.x]; }
    // weight fragments (k_split_w order: [W1 slot, unused][W2^T (mb, kb)][W1^T (mb, kb)])
    WFrag32 w2f[4], w1f[4];
    const int mb2 = wave & 1, pb2 = wave >> 1;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        w2f[kb] = ld_wfrag(wsp, NF_W1 + wave * 4 + kb);
        w1f[kb] = ld_wfrag(wsp, NF_W1 + NF_W2 + mb2 * 4 + kb);
    }
    float* T = Tw + wave * (TP * LDT);
    // LayerNorm phases: thread = (pixel t / 8, channel quad t % 8)
    const int lpx = threadIdx.x >> 3, lq = threadIdx.x & 7;
    const int c0 = wave * 16 + 4 * g;             // first of the four dh1 channels this lane holds after GEMM1
    __syncthreads();

    f32x4_t acc1[2];                               // dW1 rows 16 w + 4 g + v, columns 16 ct + r
    acc1[0] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    acc1[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float bs[4] = {0.f, 0.f, 0.f, 0.f};            // db1 partial of channels c0 .. c0 + 3 (this lane's pixels)
    float pl[8];                                   // d gamma (4) | d beta (4) of channels 4 lq ..
#pragma unroll
    for (int i = 0; i < 8; ++i) pl[i] = 0.f;

    typename HS<BF>::raw4 dh2n[4], g1n[4];
    float4 xn, dyn;
    auto issue_a = [&](long tile) {                // dh2 rows (coalesced) and the LayerNorm rows
        const long p0 = tile * TP;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = threadIdx.x + 512 * it;
            const int m = idx >> 5, k4 = idx & 31;
            dh2n[it] = (p0 + m < a.P) ? HS<BF>::ldraw(a.dh2, (p0 + m) * N1 + 4 * k4) : HS<BF>::zero();
        }
        const bool pv = p0 + lpx < a.P;
        xn = pv ? *reinterpret_cast<const float4*>(a.x + (p0 + lpx) * E + 4 * lq) : make_float4(0.f, 0.f, 0.f, 0.f);
        dyn = pv ? *reinterpret_cast<const float4*>(a.dy + (p0 + lpx) * E + 4 * lq) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto issue_g = [&](long tile) {                // gelu'(h1) in the GEMM1 result layout: pixel 16 pb + r, channels c0 ..
        const long p0 = tile * TP;
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const long p = p0 + pb * 16 + r;
            g1n[pb] = (p < a.P) ? HS<BF>::ldraw(a.g1, p * N1 + c0) : HS<BF>::zero();
        }
    };
    if ((long)blockIdx.x < ntiles) { issue_a(blockIdx.x); issue_g(blockIdx.x); }

#pragma unroll 1
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long p0 = tile * TP;
        const long next = tile + gridDim.x;
        // ---- P0: dh2 -> pieces ; LN2(x) -> Y ; keep x^, rstd, dy for the LayerNorm backward
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = threadIdx.x + 512 * it;
            const int m = idx >> 5, k4 = idx & 31;
            const float4 v = HS<BF>::widen(dh2n[it]);
            const float vv[4] = {v.x, v.y, v.z, v.w};
            u32x2_t q1, q2, q3;
            split_x4<NP>(vv, q1, q2, q3);
            uint16_t* dst = DH + m * N1 + (((k4 >> 1) ^ (m & 15)) << 3) + 4 * (k4 & 1);
            *reinterpret_cast<u32x2_t*>(dst) = q1;
            if (NP == 3) {
                *reinterpret_cast<u32x2_t*>(dst + PIECE) = q2;
                *reinterpret_cast<u32x2_t*>(dst + 2 * PIECE) = q3;
            }
        }
        const bool pvl = p0 + lpx < a.P;
        const float4 dyr = dyn;
        float xh[4], rstd;
        {
            const float4 xv = xn;
            float s = (xv.x + xv.y) + (xv.z + xv.w);
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
            const float mu = s * (1.0f / E);
            const float d0 = xv.x - mu, d1 = xv.y - mu, d2 = xv.z - mu, d3 = xv.w - mu;
            float v = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
            rstd = __builtin_amdgcn_rsqf(v * (1.0f / E) + LG_EPS);
            xh[0] = d0 * rstd; xh[1] = d1 * rstd; xh[2] = d2 * rstd; xh[3] = d3 * rstd;
            const float m_ = pvl ? 1.0f : 0.0f;    // rows past the end contribute nothing to dW1
            // LN2 gamma / beta from LDS per tile (8 registers the resident weight fragments need)
            const float4 lng = *reinterpret_cast<const float4*>(lnp + 4 * lq), lnb = *reinterpret_cast<const float4*>(lnp + E + 4 * lq);
            *reinterpret_cast<float4*>(Y + lpx * LDY + 4 * lq) =
                make_float4((xh[0] * lng.x + lnb.x) * m_, (xh[1] * lng.y + lnb.y) * m_, (xh[2] * lng.z + lnb.z) * m_, (xh[3] * lng.w + lnb.w) * m_);
        }
        if (next < ntiles) issue_a(next);
        __syncthreads();
        // ---- P1: GEMM1 (K = 128): t[16 w .. +15][64 pixels] = W2^T dh2 ; dh1 = t * gelu'(h1) -> pieces, fp32 copy, db1
        {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                const int px = pb * 16 + r;
                const uint16_t* row = DH + px * N1;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const uint16_t* p = row + (((4 * kb + g) ^ (px & 15)) << 3);
                    mfma_np32<NP>(acc, w2f[kb], lds_x8(p), lds_x8(p + PIECE), lds_x8(p + 2 * PIECE));
                }
                const float4 gv = HS<BF>::widen(g1n[pb]);
                const float d[4] = {acc[0] * gv.x, acc[1] * gv.y, acc[2] * gv.z, acc[3] * gv.w};
                bs[0] += d[0]; bs[1] += d[1]; bs[2] += d[2]; bs[3] += d[3];
                u32x2_t q1, q2, q3;
                split_x4<NP>(d, q1, q2, q3);
                uint16_t* dst = D1 + px * N1 + (((2 * wave + (g >> 1)) ^ (px & 15)) << 3) + 4 * (g & 1);
                *reinterpret_cast<u32x2_t*>(dst) = q1;
                if (NP == 3) {
                    *reinterpret_cast<u32x2_t*>(dst + PIECE) = q2;
                    *reinterpret_cast<u32x2_t*>(dst + 2 * PIECE) = q3;
                }
                *reinterpret_cast<float4*>(T + px * LDT + 4 * g) = make_float4(d[0], d[1], d[2], d[3]);
                __builtin_amdgcn_sched_barrier(0);   // keep the next pixel block's 12 operand reads from being hoisted over this one (registers)
            }
            if (next < ntiles) issue_g(next);
        }
        __syncthreads();
        // ---- P2: GEMM2 (K = 128): d LN2-out[16 mb2 .. +15][pixel block pb2] = W1^T dh1 ; dW1 tiles (K = the tile's 64 pixels)
        {
            f32x4_t o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            const int px = pb2 * 16 + r;
            const uint16_t* row = D1 + px * N1;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const uint16_t* p = row + (((4 * kb + g) ^ (px & 15)) << 3);
                mfma_np32<NP>(o, w1f[kb], lds_x8(p), lds_x8(p + PIECE), lds_x8(p + 2 * PIECE));
            }
            *reinterpret_cast<float4*>(O + px * LDY + 16 * mb2 + 4 * g) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll 4
            for (int ks = 0; ks < TP / 4; ++ks) {
                const float av = T[(4 * ks + g) * LDT + r];
                const float b0 = Y[(4 * ks + g) * LDY + r], b1 = Y[(4 * ks + g) * LDY + 16 + r];
                acc1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc1[1], 0, 0, 0);
            }
        }
        __syncthreads();
        // ---- P3: LayerNorm backward + residual, LN2 parameter gradients
        {
            const float4 ov = *reinterpret_cast<const float4*>(O + lpx * LDY + 4 * lq);
            const float dyl[4] = {pvl ? ov.x : 0.f, pvl ? ov.y : 0.f, pvl ? ov.z : 0.f, pvl ? ov.w : 0.f};
            const float4 lng = *reinterpret_cast<const float4*>(lnp + 4 * lq);
            const float gam[4] = {lng.x, lng.y, lng.z, lng.w};
            float dxh[4], m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                pl[u] += dyl[u] * xh[u];
                pl[4 + u] += dyl[u];
                dxh[u] = dyl[u] * gam[u];
                m1 += dxh[u];
                m2 += dxh[u] * xh[u];
            }
            m1 += __shfl_xor(m1, 1); m1 += __shfl_xor(m1, 2); m1 += __shfl_xor(m1, 4);
            m2 += __shfl_xor(m2, 1); m2 += __shfl_xor(m2, 2); m2 += __shfl_xor(m2, 4);
            m1 *= (1.0f / E);
            m2 *= (1.0f / E);
            if (pvl)
                *reinterpret_cast<float4*>(a.dx + (p0 + lpx) * E + 4 * lq) =
                    make_float4(dyr.x + rstd * (dxh[0] - m1 - xh[0] * m2), dyr.y + rstd * (dxh[1] - m1 - xh[1] * m2),
                                dyr.z + rstd * (dxh[2] - m1 - xh[2] * m2), dyr.w + rstd * (dxh[3] - m1 - xh[3] * m2));
        }
        // no barrier here: the next tile's P0 writes DH (last read before this tile's second barrier) and Y (last read before the
        // third); O is rewritten only after two more barriers
    }
    // ---- partial rows of this workgroup
    // LN2: threads with the same channel quad (t % 8): lanes 8 apart in a wave, then the 8 waves in a fixed order
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float sv = pl[i];
        sv += __shfl_xor(sv, 8); sv += __shfl_xor(sv, 16); sv += __shfl_xor(sv, 32);
        if (lane < 8) red[wave * 2 * E + (i >> 2) * E + 4 * lane + (i & 3)] = sv;
    }
    __syncthreads();
    if (threadIdx.x < 2 * E) {
        float sv = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) sv += red[w * 2 * E + threadIdx.x];
        if (threadIdx.x < E) a.part[blockIdx.x * (size_t)E + threadIdx.x] = sv;
        else a.part[(size_t)gridDim.x * E + blockIdx.x * (size_t)E + threadIdx.x - E] = sv;
    }
    // dW1 / db1: every wave owns its 16 rows
    float* wrow = a.w1slab + (size_t)blockIdx.x * (N1 * E);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int v = 0; v < 4; ++v) wrow[(wave * 16 + 4 * g + v) * E + 16 * ct + r] = acc1[ct][v];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        float sv = bs[v];
        sv += __shfl_xor(sv, 1); sv += __shfl_xor(sv, 2); sv += __shfl_xor(sv, 4); sv += __shfl_xor(sv, 8);
        bs[v] = sv;
    }
    if (r == 0) {
        float* brow = a.w1slab + (size_t)gridDim.x * (N1 * E) + (size_t)blockIdx.x * N1;
        *reinterpret_cast<float4*>(brow + c0) = make_float4(bs[0], bs[1], bs[2], bs[3]);
    }
}

}   // namespace

// wsplit: ffn_wsplit_bytes(32) bytes of scratch for the pre-split W2^T / W1^T fragments (written here, in front of the kernel)
int launch_ffn1_bwd_x32(const Ffn1BwdArgs& a, const float* w1, void* wsplit, hipStream_t s) {
    ProfScope prof__(LG_K_FFN1_BWD, s);
    if (!wsplit) { lg_set_error("ffn1_bwd_x32: no weight-fragment scratch"); return -3; }
    if (!a.part || !a.w1slab || !a.d_w1 || !a.d_b1) { lg_set_error("ffn1_bwd_x32: partial-sum scratch / dW1 destinations missing"); return -2; }
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn1_bwd_x32<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn1_bwd_x32<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) { lg_set_error("ffn1_bwd_x32: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    // A operands: W2^T [128 rows k][K = 128 n] = a.w2t, W1^T [32 rows c][K = 128 k] = a.w1t; the first slot of k_split_w (a [128][32]
    // matrix) is not used by this kernel and is fed the forward W1
    {
        const int rc = launch_split_w(w1, a.w2t, a.w1t, wsplit, E, a.hbf ? 1 : 3, s);
        if (rc) return rc;
    }
    const long ntiles = (a.P + TP - 1) / TP;
    const int grid = (int)(ntiles < 256 ? ntiles : 256);        // one 512-thread workgroup per CU
    if (a.hbf) k_ffn1_bwd_x32<1><<<grid, 512, LDS_BYTES, s>>>(a, reinterpret_cast<const u32x4_t*>(wsplit), ntiles);
    else k_ffn1_bwd_x32<3><<<grid, 512, LDS_BYTES, s>>>(a, reinterpret_cast<const u32x4_t*>(wsplit), ntiles);
    LG_CHECK_LAUNCH();
    int rc = launch_reduce_slab_pair(a.part, a.part + (size_t)grid * E, grid, E, a.d_ln2g, a.d_ln2b, s);
    if (rc) return rc;
    return launch_reduce_slab_wb(a.w1slab, a.w1slab + (size_t)grid * N1 * E, grid, N1, E, a.d_w1, E, a.d_b1, s);
}
