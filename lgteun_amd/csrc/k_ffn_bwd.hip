// Backward of feed_forward (k_ffn.hip) for gfx950 -- autograd of reference models/common/LGT.py:91-109 with the
// pre_norm/residual wrappers (LGT.py:45-61).  Data gradients here; the three 1x1-conv weight gradients are
// pixel-reduction GEMMs in k_wgrad.hip fed by the tensors this file materialises (dh3, dh2, dh1, LN2(x)).
//   k_ffn_dw_bwd : dh3 = (dy W3) * g3 recomputed on a halo tile (never stored) ; dh2 = dw3x3^T dh3 ; dw3x3 weight/bias
//                  gradient partials.  Depthwise work is per channel, so a workgroup takes 32 channels of an 8x16 tile.
//   k_ffn1_bwd   : dh1 = (dh2 W2) * g1 ; dx = dy + LN2^T(dh1 W1) ; LN2 param grads.  Pixelwise: a wave owns its rows.
// g1 = gelu'(h1), g3 = gelu'(h3) were saved by the forward (with a1, a3 for the weight gradients): no GELU here.
// GEMMs on v_mfma_f32_16x16x4_f32 (weights pre-transposed once per step; B fragments register-resident at e = 16).
#include "kernels.h"
#include "bwd_kernels.h"
#include "mfma.h"
#include "hstore.h"

#define LG_MAX_TRANSPOSE_JOBS 15   // W3 / W2 / W1 of the five blocks of an LGT in one launch
struct TransposeJobs { const float* src[LG_MAX_TRANSPOSE_JOBS]; float* dst[LG_MAX_TRANSPOSE_JOBS]; int rows[LG_MAX_TRANSPOSE_JOBS], cols[LG_MAX_TRANSPOSE_JOBS]; };
__global__ __launch_bounds__(256) void k_transpose(TransposeJobs t) {
    const int q = blockIdx.y;
    const float* __restrict__ src = t.src[q];
    float* __restrict__ dst = t.dst[q];
    const int rows = t.rows[q], cols = t.cols[q];
    long n = (long)rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        int r = (int)(i / cols), c = (int)(i - (long)r * cols);
        dst[(long)c * rows + r] = src[i];
    }
}
int launch_transpose3(const float* const* src, float* const* dst, const int* rows, const int* cols, int njobs, hipStream_t s) {
    if (njobs < 1 || njobs > LG_MAX_TRANSPOSE_JOBS) { lg_set_error("transpose: njobs=%d", njobs); return -2; }
    TransposeJobs t;
    long nmax = 0;
    for (int q = 0; q < LG_MAX_TRANSPOSE_JOBS; ++q) {
        const int u = q < njobs ? q : 0;
        t.src[q] = src[u]; t.dst[q] = dst[u]; t.rows[q] = rows[u]; t.cols[q] = cols[u];
        if ((long)rows[u] * cols[u] > nmax) nmax = (long)rows[u] * cols[u];
    }
    int grid = (int)((nmax + 255) / 256);
    if (grid > 1024) grid = 1024;
    k_transpose<<<dim3(grid, njobs), 256, 0, s>>>(t);
    LG_CHECK_LAUNCH();
    return 0;
}
int launch_transpose(const float* src, float* dst, int rows, int cols, hipStream_t s) { return launch_transpose3(&src, &dst, &rows, &cols, 1, s); }

// ------------------------------------------------------------------------------------------------
#define DW_CG 32   // channels per workgroup (depthwise work is per channel); 16 (3 workgroups per CU) measured no faster
#define DW_TPW 8   // tiles walked by one workgroup (weight-gradient partials stay in registers across them)
#ifndef LG_DW_INTERLEAVE
#define LG_DW_INTERLEAVE 1   // tiles dealt round-robin to the workgroups instead of DW_TPW consecutive ones each (A/B: step 7.463 -> 7.437 ms)
#endif
// PRE: a.g3 holds the pre-activation h3 (the forward saved h1 / h2 / h3 only): gelu'(h3) is evaluated here, on the halo tile
template <int E, bool BF, int CG, bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((E == 16 || (E == 32 && CG == 32)) ? 2 : 1))) void k_ffn_dw_bwd(FfnDwBwdArgs a, int tiles_x, int tiles_y) {
    constexpr int N1 = 4 * E, TY = 8, TX = 16, HX = TX + 2, HY = TY + 2, NH = HX * HY, MH = 192, CQ = CG / 4, NTG = CG / 16;
    constexpr int LDY = E + 4, LDG = CG + 4;
    extern __shared__ float smem[];
    float* bufY = smem;                  // [MH][LDY] dy on the halo tile
    float* bufG = bufY + MH * LDY;       // [NH][LDG] dh3 on the halo tile, this workgroup's 32 channels (0 outside the image)
    float* bufH = bufG + NH * LDG;       // [NH][LDG] h2 on the halo tile
    // [4][CQ][40] partial rows of the final reduction: behind the tiles, or (e = 32 in 32-channel groups: two workgroups per CU need
    // <= 80 KB each) on top of bufY, which is dead after the tile loop
    constexpr bool RED_ALIAS = (E == 32 && CG == 32);
    float* red = RED_ALIAS ? bufY : bufH + NH * LDG;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int c0 = blockIdx.y * CG;
    const int h = a.h, w = a.w;
    const int q = threadIdx.x % CQ;
    float wq[4][9];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 9; ++k) wq[u][k] = a.dww[(c0 + 4 * q + u) * 9 + k];
    // taps and gradient partials as channel PAIRS: phase P2 runs on v_pk_fma_f32 (38 instead of 76 vector instructions per pixel quad)
    lg_v2f wq01[9], wq23[9], pw01[10], pw23[10];
#pragma unroll
    for (int k = 0; k < 9; ++k) { wq01[k] = (lg_v2f){wq[0][k], wq[1][k]}; wq23[k] = (lg_v2f){wq[2][k], wq[3][k]}; }
#pragma unroll
    for (int k = 0; k < 10; ++k) { pw01[k] = (lg_v2f){0.f, 0.f}; pw23[k] = (lg_v2f){0.f, 0.f}; }
    const int ntiles = a.B * tiles_x * tiles_y;
    constexpr int NDY = MH * (E / 4) / 256;             // dy halo tile: float4 items per thread
    constexpr int NG3 = (NH * CQ + 255) / 256;          // h2 / g3 halo tile: float4 items per thread
    static_assert(MH * (E / 4) % 256 == 0, "dy tile items");
    // PF (e = 16): the halo-tile loads were 60 % of a tile's time with nothing else in flight (in-kernel clock stamps), so the
    // NEXT tile's dy / h2 / g3 are requested into registers right after the current tile's copies went to LDS, and land under
    // the GEMM + depthwise phases.  The W3 fragments are register-resident for that: a weight load in the compute phases
    // would make its s_waitcnt (one in-order counter) wait for the prefetch as well.
    // (At e = 32 -- one workgroup per CU, fragments 32 registers -- the same prefetch measured SLOWER: 369 -> 397 us.)
    constexpr bool PF = (E == 16);
    float4 w3r[PF ? NTG : 1][1];
    if (PF) load_bfrag<NTG, 1>(reinterpret_cast<float4(&)[NTG][1]>(w3r), a.w3t + (size_t)c0 * E, E);
    float4 dyr[NDY];
    typename HS<BF>::raw4 h2r[NG3], g3n[NG3];   // raw bits: widened when they leave the prefetch registers
    auto issue = [&](int tile_) {
        int t_ = tile_;
        const int tx_ = t_ % tiles_x;
        t_ /= tiles_x;
        const int ty_ = t_ % tiles_y;
        const long b_ = t_ / tiles_y;
        const int yb = ty_ * TY, xb = tx_ * TX;
#pragma unroll
        for (int it = 0; it < NDY; ++it) {
            const int i = threadIdx.x + it * 256;
            const int m = i / (E / 4), k4 = i - m * (E / 4);
            const int hy = m / HX, hx = m - hy * HX;
            const int y = yb + hy - 1, x = xb + hx - 1;
            dyr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < NH && y >= 0 && y < h && x >= 0 && x < w) dyr[it] = *reinterpret_cast<const float4*>(a.dy + ((b_ * h + y) * (long)w + x) * E + 4 * k4);
        }
#pragma unroll
        for (int it = 0; it < NG3; ++it) {
            const int i = threadIdx.x + it * 256;
            h2r[it] = HS<BF>::zero();
            g3n[it] = HS<BF>::zero();
            if (i < NH * CQ) {
                const int m = i / CQ, qq = i - m * CQ;
                const int hy = m / HX, hx = m - hy * HX;
                const int y = yb + hy - 1, x = xb + hx - 1;
                if (y >= 0 && y < h && x >= 0 && x < w) {
                    const long o = ((b_ * h + y) * (long)w + x) * N1 + c0 + 4 * qq;
                    h2r[it] = HS<BF>::ldraw(a.h2, o);
                    g3n[it] = HS<BF>::ldraw(a.g3, o);
                }
            }
        }
    };
#if LG_DW_INTERLEAVE
    const int tile_first = blockIdx.x, tile_step = gridDim.x;
    const int tile_end = ntiles;
#else
    const int tile_first = blockIdx.x * DW_TPW, tile_step = 1;
    const int tile_end = min((int)(blockIdx.x + 1) * DW_TPW, ntiles);
#endif
    if (PF && tile_first < tile_end) issue(tile_first);
    for (int tile = tile_first; tile < tile_end; tile += tile_step) {
    int t = tile;
    const int tx_i = t % tiles_x;
    t /= tiles_x;
    const int ty_i = t % tiles_y;
    const long b = t / tiles_y;
    const int y0 = ty_i * TY, x0 = tx_i * TX;
    __syncthreads();
    // ---- P0: dy halo tile (all e channels) and h2 halo tile (this channel group) -> LDS; g3 stays in registers until the GEMM
    // result is in LDS
    if (!PF) issue(tile);
    float4 g3r[NG3];
#pragma unroll
    for (int it = 0; it < NDY; ++it) {
        const int i = threadIdx.x + it * 256;
        const int m = i / (E / 4), k4 = i - m * (E / 4);
        *reinterpret_cast<float4*>(bufY + m * LDY + 4 * k4) = dyr[it];
    }
#pragma unroll
    for (int it = 0; it < NG3; ++it) {
        const int i = threadIdx.x + it * 256;
        g3r[it] = HS<BF>::widen(g3n[it]);
        if (PRE) {   // gelu'(h3) from the saved pre-activation (dy is 0 outside the image, so is dh3)
            lg_v2f a01, a23, g01, g23;
            gelu2_both_f((lg_v2f){g3r[it].x, g3r[it].y}, a01, g01);
            gelu2_both_f((lg_v2f){g3r[it].z, g3r[it].w}, a23, g23);
            g3r[it] = make_float4(g01.x, g01.y, g23.x, g23.y);
        }
        if (i < NH * CQ) {
            const int m = i / CQ, qq = i - m * CQ;
            *reinterpret_cast<float4*>(bufH + m * LDG + 4 * qq) = HS<BF>::widen(h2r[it]);
        }
    }
    if (PF && tile + tile_step < tile_end) issue(tile + tile_step);
    __syncthreads();
    // ---- P1: dh3 = (dy W3)[:, c0:c0+32] * g3 on the halo tile; wave owns 48 rows (3 m-tiles) x 2 n-tiles
    {
        f32x4 acc[3][NTG];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTG; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (PF) wave_gemm_rb<3, NTG, 1>(acc, bufY + wave * 48 * LDY, LDY, reinterpret_cast<const float4(&)[NTG][1]>(w3r));
        else wave_gemm<3, NTG, E>(acc, bufY + wave * 48 * LDY, LDY, a.w3t + (size_t)c0 * E);
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int m = wave * 48 + mt * 16 + 4 * g + v;
#pragma unroll
                for (int nt = 0; nt < NTG; ++nt)
                    if (m < NH) bufG[m * LDG + nt * 16 + r] = acc[mt][nt][v];
            }
    }
    __syncthreads();
    // dh3 *= g3 (g3 is 0 outside the image)
#pragma unroll
    for (int it = 0; it < NG3; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < NH * CQ) {
            const int m = i / CQ, qq = i - m * CQ;
            const float4 t = *reinterpret_cast<const float4*>(bufG + m * LDG + 4 * qq);
            *reinterpret_cast<float4*>(bufG + m * LDG + 4 * qq) = make_float4(t.x * g3r[it].x, t.y * g3r[it].y, t.z * g3r[it].z, t.w * g3r[it].w);
        }
    }
    __syncthreads();
    // ---- P2: dh2 = dw^T dh3 and the depthwise weight/bias gradient partials; thread <-> (pixel, channel quad)
    for (int m = threadIdx.x / CQ; m < TY * TX; m += 256 / CQ) {
        const int ty = m / TX, tx = m - ty * TX;
        const int y = y0 + ty, x = x0 + tx;
        if (y >= h || x >= w) continue;
        const float4 gc = *reinterpret_cast<const float4*>(bufG + ((ty + 1) * HX + tx + 1) * LDG + 4 * q);
        const lg_v2f gc01 = (lg_v2f){gc.x, gc.y}, gc23 = (lg_v2f){gc.z, gc.w};
        pw01[9] += gc01; pw23[9] += gc23;
        lg_v2f acc01 = (lg_v2f){0.f, 0.f}, acc23 = (lg_v2f){0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                // forward: h3(y',x') += w[dy][dx] * h2(y'+dy-1, x'+dx-1)  ->  h2(y,x) feeds h3(y-dy+1, x-dx+1)
                const float4 gv = *reinterpret_cast<const float4*>(bufG + ((ty + 2 - dy) * HX + tx + 2 - dx) * LDG + 4 * q);
                acc01 = wq01[dy * 3 + dx] * (lg_v2f){gv.x, gv.y} + acc01;
                acc23 = wq23[dy * 3 + dx] * (lg_v2f){gv.z, gv.w} + acc23;
                const float4 hv = *reinterpret_cast<const float4*>(bufH + ((ty + dy) * HX + tx + dx) * LDG + 4 * q);
                pw01[dy * 3 + dx] = gc01 * (lg_v2f){hv.x, hv.y} + pw01[dy * 3 + dx];
                pw23[dy * 3 + dx] = gc23 * (lg_v2f){hv.z, hv.w} + pw23[dy * 3 + dx];
            }
        HS<BF>::st4(a.dh2, ((b * h + y) * (long)w + x) * N1 + c0 + 4 * q, make_float4(acc01.x, acc01.y, acc23.x, acc23.y));
    }
    }   // tiles of this workgroup
    const int tile_id = blockIdx.x;
    if (RED_ALIAS) __syncthreads();   // the last tile's GEMM has read bufY
    // ---- P3: partials -> one slab row per workgroup (fixed-order reduction later; no float atomics)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            float v = u == 0 ? pw01[k].x : (u == 1 ? pw01[k].y : (u == 2 ? pw23[k].x : pw23[k].y));
#pragma unroll
            for (int off = CQ; off < 64; off <<= 1) v += __shfl_xor(v, off);
            if (lane < CQ) red[(wave * CQ + q) * 40 + u * 10 + k] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < CQ * 40; i += 256) {
        const float v = red[i] + red[CQ * 40 + i] + red[2 * CQ * 40 + i] + red[3 * CQ * 40 + i];
        const int qq = i / 40, rem = i - qq * 40, u = rem / 10, k = rem - u * 10;
        const int c = c0 + 4 * qq + u;
        if (k < 9) a.slab_w[(size_t)tile_id * N1 * 9 + c * 9 + k] = v;
        else a.slab_b[(size_t)tile_id * N1 + c] = v;
    }
}

template <int E>
static int launch_ffn_dw_bwd_t(const FfnDwBwdArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_FFN2_BWD, s);
    int tiles_x = (a.w + 15) / 16, tiles_y = (a.h + 7) / 8;
    const long nwg = ((long)a.B * tiles_x * tiles_y + DW_TPW - 1) / DW_TPW;
    // channel group per workgroup: 32 at e = 16 (two groups); 64 from e = 32 up, where every extra group re-reads the dy halo tile
    // and repeats the dy x W3 GEMM rows
    constexpr int CG = (E >= 64 ? 64 : DW_CG);
    dim3 grid((unsigned)nwg, 4 * E / CG);
    const size_t lds = (size_t)(192 * (E + 4) + 2 * 180 * (CG + 4) + ((E == 32 && CG == 32) ? 0 : 4 * (CG / 4) * 40)) * sizeof(float);
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn_dw_bwd<E, false, CG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_dw_bwd<E, true, CG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if constexpr (E == 16) { if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn_dw_bwd<E, false, CG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); }
        if (e != hipSuccess) { lg_set_error("ffn_dw_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    if (a.pre && (a.hbf || E != 16)) { lg_set_error("ffn_dw_bwd: pre-activation saves are fp32, e = 16"); return -2; }
    if (a.hbf) k_ffn_dw_bwd<E, true, CG, false><<<grid, 256, lds, s>>>(a, tiles_x, tiles_y);
    else if (a.pre) { if constexpr (E == 16) k_ffn_dw_bwd<E, false, CG, true><<<grid, 256, lds, s>>>(a, tiles_x, tiles_y); }
    else k_ffn_dw_bwd<E, false, CG, false><<<grid, 256, lds, s>>>(a, tiles_x, tiles_y);
    LG_CHECK_LAUNCH();
    int rc = launch_reduce_slab(a.slab_w, nwg, 4 * E, 9, a.d_dww, 9, 4 * E, 9, s);
    if (rc) return rc;
    return launch_reduce_slab(a.slab_b, nwg, 1, 4 * E, a.d_dwb, 4 * E, 1, 4 * E, s);
}
size_t ffn_dw_bwd_slab_floats(int e, int B, int h, int w) { return (size_t)B * ((w + 15) / 16) * ((h + 7) / 8) * 4 * e * 10; }
int launch_ffn_dw_bwd(int e, const FfnDwBwdArgs& a, hipStream_t s) {
    if (e == 16) return launch_ffn_dw_bwd_t<16>(a, s);
    if (e == 32) return launch_ffn_dw_bwd_t<32>(a, s);
    if (e == 64) return launch_ffn_dw_bwd_t<64>(a, s);
    lg_set_error("ffn_dw_bwd: e=%d unsupported", e);
    return -1;
}

// ------------------------------------------------------------------------------------------------
// PRE: a.g1 holds the pre-activation h1: gelu'(h1) is evaluated here
template <int E, int MT, bool BF, bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(E == 16 ? 2 : 1))) void k_ffn1_bwd(Ffn1BwdArgs a, long nchunks) {
    constexpr int N1 = 4 * E, MW = 16 * MT, LDH = N1 + 4, LDO = E + 1, NTE = E / 16, LDY = LDO;
    constexpr bool RB = (E == 16);
    constexpr bool FW1 = (E <= 32);   // dW1 / db1 accumulated here (bwd_kernels.h: ffn1_bwd_fuses_w1)
    constexpr int NB1 = N1 / 64;      // 64-wide blocks of dh1 columns: element t of a float4 at column 4r feeds tile 4*nb + t
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[4 * 2 * E];
    __shared__ float lnp[2 * E];   // LN2 gamma | beta: read from LDS in the loop (a global load there would stall on the prefetch)
    if (threadIdx.x < E) { lnp[threadIdx.x] = a.ln2g[threadIdx.x]; lnp[E + threadIdx.x] = a.ln2b[threadIdx.x]; }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    float* bufD = smem + wave * (MW * (2 * LDH + LDO));   // [MW][LDH] dh2 rows
    float* bufD1 = bufD + MW * LDH;                        // [MW][LDH] dh1 rows
    float* bufO = bufD1 + MW * LDH;                        // [MW][LDO] d(LN2 output)
    float* bufY = bufO;                                    // [MW][LDY] LN2(x), B operand of the dW1 tiles: dead before bufO is written
    f32x4 acc1[FW1 ? 4 * NB1 : 1][FW1 ? NTE : 1];
    float4 bs1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FW1) {
#pragma unroll
        for (int i = 0; i < 4 * NB1; ++i)
#pragma unroll
            for (int j = 0; j < NTE; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // PF (e = 16): the chunk's loads (dh2, g1, x, dy) were ~58 % of a chunk's time with nothing else in flight (in-kernel
    // clock stamps), so the NEXT chunk's operands are requested into registers as soon as the current ones are committed.
    // For that no other vector-memory load may sit in the compute phases (one in-order s_waitcnt counter): W2^T lives in LDS
    // (it would take 64 VGPRs as fragments, the prefetch needs them) and the W1^T fragments in 16 registers.
    constexpr bool PF = RB;
    float* w2l = smem + 4 * (MW * (2 * LDH + LDO));   // [N1][LDH] W2^T rows (PF)
    if (PF) {
        for (int i2 = threadIdx.x; i2 < N1 * (N1 / 4); i2 += 256) {
            const int row = i2 / (N1 / 4), k4 = i2 - row * (N1 / 4);
            *reinterpret_cast<float4*>(w2l + row * LDH + 4 * k4) = *reinterpret_cast<const float4*>(a.w2t + (size_t)row * N1 + 4 * k4);
        }
    }
    // FW2 (pre-activation saves, e = 16): gelu(h1) is evaluated here for gelu'(h1) anyway, so dW2 = sum_p dh2 (x) gelu(h1) and db2 are
    // accumulated in this kernel too: the rows of all four waves (dh2 in bufD, gelu(h1) in bufA) are the operands, wave w owns the
    // output rows n = 4 i + w (tile t of a float4 at column 4r = channel 4r + t, as in k_wgrad_t's VEC layout)
    constexpr bool FW2 = PRE && E == 16 && LG_FW2;
    float* bufA_all = w2l + N1 * LDH;            // [4][MW][LDH] gelu(h1) rows (FW2)
    float* bufA = bufA_all + wave * (MW * LDH);
    f32x4 acc2[FW2 ? 4 : 1];
    float4 bs2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FW2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float4 w1r[PF ? NTE : 1][PF ? N1 / 16 : 1];
    if (PF) load_bfrag<NTE, N1 / 16>(reinterpret_cast<float4(&)[NTE][N1 / 16]>(w1r), a.w1t, N1);
    // LN-gradient partials: PF: this lane's channel quarter (4 gamma + 4 beta); otherwise all 2E of the lane's pixel
    constexpr int NPL = (E == 16) ? 8 : 2 * E;
    float pl[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) pl[i] = 0.f;
    constexpr int NR = MW * (N1 / 4) / 64;
    // PF: x / dy of the wave's 16 pixels are spread over all 64 lanes (lane = 4 * pixel + channel quarter: one float4 each,
    // one coalesced 1 KB row per load) and the LayerNorm phases work 4 lanes per pixel; otherwise lane < MW owns a whole pixel
    constexpr int NX = PF ? 1 : E / 4;
    float4 xn[NX], dyn[NX];
    typename HS<BF>::raw4 dh2n[NR], g1n[NR];    // raw bits: widened when they leave the prefetch registers
    auto issue = [&](long chunk_) {
        const long q0 = (chunk_ * 4 + wave) * MW;
#pragma unroll
        for (int it = 0; it < NR; ++it) {
            const int i = lane + it * 64;
            const int m = i / (N1 / 4), k4 = i - m * (N1 / 4);
            dh2n[it] = HS<BF>::zero();
            g1n[it] = HS<BF>::zero();
            if (q0 + m < a.P) {
                dh2n[it] = HS<BF>::ldraw(a.dh2, (q0 + m) * N1 + 4 * k4);
                g1n[it] = HS<BF>::ldraw(a.g1, (q0 + m) * N1 + 4 * k4);
            }
        }
        if (PF) {
            const bool pv = q0 + (lane >> 2) < a.P;
            xn[0] = pv ? reinterpret_cast<const float4*>(a.x + (q0 + (lane >> 2)) * E)[lane & 3] : make_float4(0.f, 0.f, 0.f, 0.f);
            dyn[0] = pv ? reinterpret_cast<const float4*>(a.dy + (q0 + (lane >> 2)) * E)[lane & 3] : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const bool pv = lane < MW && q0 + lane < a.P;
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                xn[k] = pv ? reinterpret_cast<const float4*>(a.x + (q0 + lane) * E)[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                dyn[k] = pv ? reinterpret_cast<const float4*>(a.dy + (q0 + lane) * E)[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    if (PF && (long)blockIdx.x < nchunks) issue(blockIdx.x);
    // persistent: the workgroup walks row chunks (weights and LN-gradient partials stay in registers / LDS)
#pragma unroll 1
    for (long chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const long p0 = (chunk * 4 + wave) * MW;
    __syncthreads();
    // operands of the chunk: dh2 -> LDS, g1 and the LayerNorm-phase rows (x, dy of pixel `lane`) wait in registers
    if (!PF) issue(chunk);
    float4 g1r[NR];
#pragma unroll
    for (int it = 0; it < NR; ++it) {
        const int i = lane + it * 64;
        const int m = i / (N1 / 4), k4 = i - m * (N1 / 4);
        g1r[it] = HS<BF>::widen(g1n[it]);
        const float4 d2 = HS<BF>::widen(dh2n[it]);
        if (PRE) {   // gelu'(h1) and gelu(h1) from the saved pre-activation, one exponential for both
            lg_v2f a01, a23, g01, g23;
            gelu2_both_f((lg_v2f){g1r[it].x, g1r[it].y}, a01, g01);
            gelu2_both_f((lg_v2f){g1r[it].z, g1r[it].w}, a23, g23);
            g1r[it] = make_float4(g01.x, g01.y, g23.x, g23.y);
            if (FW2) {
                *reinterpret_cast<float4*>(bufA + m * LDH + 4 * k4) = make_float4(a01.x, a01.y, a23.x, a23.y);
                bs2.x += d2.x; bs2.y += d2.y; bs2.z += d2.z; bs2.w += d2.w;   // k4 = lane % (N1/4) is the same for every `it`
            }
        }
        *reinterpret_cast<float4*>(bufD + m * LDH + 4 * k4) = d2;
    }
    float4 xr[NX], dyr[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) { xr[k] = xn[k]; dyr[k] = dyn[k]; }
    if (PF && chunk + (long)gridDim.x < nchunks) issue(chunk + (long)gridDim.x);
    float mu_q = 0.f, rstd_q = 0.f;   // PF: LayerNorm statistics of this lane's pixel (kept for the backward phase)
    if constexpr (PF) {
        // 4 lanes per pixel: quarter sums joined by two xor-shuffles
        const int px = lane >> 2, qd = lane & 3;
        const bool pv = p0 + px < a.P;
        const float4 xq = xr[0];
        float sm = (xq.x + xq.y) + (xq.z + xq.w);
        sm += __shfl_xor(sm, 1);
        sm += __shfl_xor(sm, 2);
        mu_q = sm * (1.0f / E);
        const float d0 = xq.x - mu_q, d1 = xq.y - mu_q, d2 = xq.z - mu_q, d3 = xq.w - mu_q;
        float vs = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        vs += __shfl_xor(vs, 1);
        vs += __shfl_xor(vs, 2);
        rstd_q = __builtin_amdgcn_rsqf(vs * (1.0f / E) + LG_EPS);
        float* yrow = bufY + px * LDY + 4 * qd;
        yrow[0] = pv ? d0 * rstd_q * lnp[4 * qd] + lnp[E + 4 * qd] : 0.f;
        yrow[1] = pv ? d1 * rstd_q * lnp[4 * qd + 1] + lnp[E + 4 * qd + 1] : 0.f;
        yrow[2] = pv ? d2 * rstd_q * lnp[4 * qd + 2] + lnp[E + 4 * qd + 2] : 0.f;
        yrow[3] = pv ? d3 * rstd_q * lnp[4 * qd + 3] + lnp[E + 4 * qd + 3] : 0.f;
    } else
    if (FW1 && lane < MW) {   // LN2(x) of this lane's pixel -> bufY (rows of pixels past the end are zero)
        const bool pv = p0 + lane < a.P;
        float xv[E];
#pragma unroll
        for (int k = 0; k < E / 4; ++k) { xv[4 * k] = xr[k].x; xv[4 * k + 1] = xr[k].y; xv[4 * k + 2] = xr[k].z; xv[4 * k + 3] = xr[k].w; }
        float mu, rstd;
        ln_stats<E>(xv, mu, rstd);
#pragma unroll
        for (int c = 0; c < E; ++c) bufY[lane * LDY + c] = pv ? (xv[c] - mu) * rstd * lnp[c] + lnp[E + c] : 0.f;
    }
    __syncthreads();
    if constexpr (FW2) {
        // ---- dW2[4 i + wave][4 j + t] += sum over the workgroup's 64 pixels: A[i = r][k = g] = dh2[pixel g][4 r + wave],
        // B[k = g][j = r] = gelu(h1)[pixel g][4 r + t]
#pragma unroll 1
        for (int sw = 0; sw < 4; ++sw) {
            const float* dsrc = smem + sw * (MW * (2 * LDH + LDO));
            const float* asrc = bufA_all + sw * (MW * LDH);
#pragma unroll
            for (int ks = 0; ks < MW / 4; ++ks) {
                const float yv = dsrc[(4 * ks + g) * LDH + 4 * r + wave];
                const float4 xv = *reinterpret_cast<const float4*>(asrc + (4 * ks + g) * LDH + 4 * r);
                acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv, xv.x, acc2[0], 0, 0, 0);
                acc2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv, xv.y, acc2[1], 0, 0, 0);
                acc2[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv, xv.z, acc2[2], 0, 0, 0);
                acc2[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv, xv.w, acc2[3], 0, 0, 0);
            }
        }
    }
    // ---- dh1 = (dh2 W2) * g1
    for (int nc = 0; nc < N1; nc += 64) {
        f32x4 acc[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (PF) wave_gemm_ld<MT, 4, N1>(acc, bufD, LDH, w2l + nc * LDH, LDH);
        else wave_gemm<MT, 4, N1>(acc, bufD, LDH, a.w2t + (size_t)nc * N1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = nc + nt * 16 + r;
#pragma unroll
                for (int v = 0; v < 4; ++v) bufD1[(mt * 16 + 4 * g + v) * LDH + col] = acc[mt][nt][v];
            }
    }
    __syncthreads();
    // dh1 = (dh2 W2) * g1 : coalesced 16-byte pass over the wave's rows (dh1 out), result kept in LDS for the next GEMM
#pragma unroll
    for (int it = 0; it < NR; ++it) {
        const int i = lane + it * 64;
        const int m = i / (N1 / 4), k4 = i - m * (N1 / 4);
        const float4 t = *reinterpret_cast<const float4*>(bufD1 + m * LDH + 4 * k4);
        const float4 d = make_float4(t.x * g1r[it].x, t.y * g1r[it].y, t.z * g1r[it].z, t.w * g1r[it].w);
        if (FW1) { bs1.x += d.x; bs1.y += d.y; bs1.z += d.z; bs1.w += d.w; }   // k4 = lane % (N1/4) is the same for every `it`
        else if (p0 + m < a.P) HS<BF>::st4(a.dh1, (p0 + m) * N1 + 4 * k4, d);
        *reinterpret_cast<float4*>(bufD1 + m * LDH + 4 * k4) = d;
    }
    __syncthreads();
    if (FW1) {
        // ---- dW1[n][k] += sum_pixels dh1[p][n] * LN2(x)[p][k]: pixels are the MFMA K dimension (4 per step)
#pragma unroll
        for (int ks = 0; ks < MW / 4; ++ks) {
            float4 af[NB1];
            float bv[NTE];
#pragma unroll
            for (int nb = 0; nb < NB1; ++nb) af[nb] = *reinterpret_cast<const float4*>(bufD1 + (4 * ks + g) * LDH + nb * 64 + 4 * r);
#pragma unroll
            for (int kt = 0; kt < NTE; ++kt) bv[kt] = bufY[(4 * ks + g) * LDY + kt * 16 + r];
#pragma unroll
            for (int nb = 0; nb < NB1; ++nb) {
                const float av[4] = {af[nb].x, af[nb].y, af[nb].z, af[nb].w};
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int kt = 0; kt < NTE; ++kt) acc1[4 * nb + t][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[kt], acc1[4 * nb + t][kt], 0, 0, 0);
            }
        }
    }
    // ---- d(LN2 out) = dh1 W1
    {
        f32x4 acc[MT][NTE];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTE; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (PF) wave_gemm_rb<MT, NTE, N1 / 16>(acc, bufD1, LDH, reinterpret_cast<const float4(&)[NTE][N1 / 16]>(w1r));
        else wave_gemm<MT, NTE, N1>(acc, bufD1, LDH, a.w1t);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTE; ++nt)
#pragma unroll
                for (int v = 0; v < 4; ++v) bufO[(mt * 16 + 4 * g + v) * LDO + nt * 16 + r] = acc[mt][nt][v];
    }
    __syncthreads();
    // ---- LayerNorm backward + residual, LN2 parameter gradients
    if constexpr (PF) {
        const int px = lane >> 2, qd = lane & 3;
        const bool pv = p0 + px < a.P;
        const float4 xq = xr[0];
        const float xh[4] = {(xq.x - mu_q) * rstd_q, (xq.y - mu_q) * rstd_q, (xq.z - mu_q) * rstd_q, (xq.w - mu_q) * rstd_q};
        float dxh[4], m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float dyl = pv ? bufO[px * LDO + 4 * qd + u] : 0.f;
            pl[u] += dyl * xh[u];
            pl[4 + u] += dyl;
            dxh[u] = dyl * lnp[4 * qd + u];
            m1 += dxh[u];
            m2 += dxh[u] * xh[u];
        }
        m1 += __shfl_xor(m1, 1); m1 += __shfl_xor(m1, 2);
        m2 += __shfl_xor(m2, 1); m2 += __shfl_xor(m2, 2);
        m1 *= (1.0f / E);
        m2 *= (1.0f / E);
        if (pv) {
            const float4 dv = dyr[0];
            reinterpret_cast<float4*>(a.dx + (p0 + px) * E)[qd] =
                make_float4(dv.x + rstd_q * (dxh[0] - m1 - xh[0] * m2), dv.y + rstd_q * (dxh[1] - m1 - xh[1] * m2),
                            dv.z + rstd_q * (dxh[2] - m1 - xh[2] * m2), dv.w + rstd_q * (dxh[3] - m1 - xh[3] * m2));
        }
    } else
    if (lane < MW && p0 + lane < a.P) {
        const int m = lane;
        const long p = p0 + m;
        float xv[E];
#pragma unroll
        for (int k = 0; k < E / 4; ++k) { xv[4 * k] = xr[k].x; xv[4 * k + 1] = xr[k].y; xv[4 * k + 2] = xr[k].z; xv[4 * k + 3] = xr[k].w; }
        float mu, rstd;
        ln_stats<E>(xv, mu, rstd);
        float m1 = 0.f, m2 = 0.f;
        float dxh[E];
#pragma unroll
        for (int c = 0; c < E; ++c) {
            const float xh = (xv[c] - mu) * rstd;
            const float dyl = bufO[m * LDO + c];
            pl[c] += dyl * xh;
            pl[E + c] += dyl;
            dxh[c] = dyl * lnp[c];
            m1 += dxh[c];
            m2 += dxh[c] * xh;
            xv[c] = xh;
        }
        m1 *= (1.0f / E);
        m2 *= (1.0f / E);
        float4* dxo = reinterpret_cast<float4*>(a.dx + p * E);
        float4* y2o = reinterpret_cast<float4*>(a.y2 + p * E);
#pragma unroll
        for (int k = 0; k < E / 4; ++k) {
            const float4 dv = dyr[k];
            float o[4], yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = 4 * k + u;
                o[u] = rstd * (dxh[c] - m1 - xv[c] * m2);
                yv[u] = xv[c] * lnp[c] + lnp[E + c];
            }
            dxo[k] = make_float4(dv.x + o[0], dv.y + o[1], dv.z + o[2], dv.w + o[3]);
            if (!FW1) y2o[k] = make_float4(yv[0], yv[1], yv[2], yv[3]);
        }
    }
    }   // chunks of this workgroup
    if constexpr (PF) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // lanes with the same channel quarter (lane & 3) hold the same channels
            float sv = pl[i];
#pragma unroll
            for (int off = 32; off >= 4; off >>= 1) sv += __shfl_xor(sv, off);
            if (lane < 4) red[wave * 2 * E + (i >> 2) * E + 4 * lane + (i & 3)] = sv;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            float sv = pl[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sv += __shfl_xor(sv, off);
            if (lane == 0) red[wave * 2 * E + i] = sv;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * E) {
        const float sv = red[threadIdx.x] + red[2 * E + threadIdx.x] + red[4 * E + threadIdx.x] + red[6 * E + threadIdx.x];
        // partial rows [grid][E] d gamma | [grid][E] d beta, summed by launch_reduce_slab_pair
        if (threadIdx.x < E) a.part[blockIdx.x * (size_t)E + threadIdx.x] = sv;
        else a.part[(size_t)gridDim.x * E + blockIdx.x * (size_t)E + threadIdx.x - E] = sv;
    }
    if (FW1) {
        // the 4 waves' dW1 / db1 partials, summed in LDS in a fixed order -> one slab row per workgroup
        float* rw = smem;   // [N1][E] + [N1]
        __syncthreads();
        // db1: lanes with equal lane % (N1/4) hold the same 4 columns
        float4 b4 = bs1;
#pragma unroll
        for (int off = N1 / 4; off < 64; off <<= 1) {
            b4.x += __shfl_xor(b4.x, off); b4.y += __shfl_xor(b4.y, off); b4.z += __shfl_xor(b4.z, off); b4.w += __shfl_xor(b4.w, off);
        }
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int i = 0; i < 4 * NB1; ++i)
#pragma unroll
                    for (int kt = 0; kt < NTE; ++kt)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int n = (i >> 2) * 64 + 4 * (4 * g + v) + (i & 3), k = kt * 16 + r;
                            rw[n * E + k] = (w == 0 ? 0.f : rw[n * E + k]) + acc1[i][kt][v];
                        }
                if (lane < N1 / 4) {
                    float* rb = rw + N1 * E + 4 * lane;
                    rb[0] = (w == 0 ? 0.f : rb[0]) + b4.x; rb[1] = (w == 0 ? 0.f : rb[1]) + b4.y;
                    rb[2] = (w == 0 ? 0.f : rb[2]) + b4.z; rb[3] = (w == 0 ? 0.f : rb[3]) + b4.w;
                }
            }
            __syncthreads();
        }
        float* wrow = a.w1slab + (size_t)blockIdx.x * (N1 * E);
        for (int i = threadIdx.x; i < N1 * E; i += 256) wrow[i] = rw[i];
        float* brow = a.w1slab + (size_t)gridDim.x * (N1 * E) + (size_t)blockIdx.x * N1;
        for (int i = threadIdx.x; i < N1; i += 256) brow[i] = rw[N1 * E + i];
    }
    if constexpr (FW2) {
        // dW2: the waves own disjoint output rows -> straight to the slab row of this workgroup; db2 through LDS (4 waves, fixed order)
        float* wrow = a.w2slab + (size_t)blockIdx.x * (N1 * N1);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int n = 4 * (4 * g + v) + wave;
            *reinterpret_cast<float4*>(wrow + n * N1 + 4 * r) = make_float4(acc2[0][v], acc2[1][v], acc2[2][v], acc2[3][v]);
        }
        float4 b4 = bs2;
#pragma unroll
        for (int off = N1 / 4; off < 64; off <<= 1) {
            b4.x += __shfl_xor(b4.x, off); b4.y += __shfl_xor(b4.y, off); b4.z += __shfl_xor(b4.z, off); b4.w += __shfl_xor(b4.w, off);
        }
        float* rb2 = smem;   // [N1]
        __syncthreads();
        for (int w = 0; w < 4; ++w) {
            if (wave == w && lane < N1 / 4) {
                float* rb = rb2 + 4 * lane;
                rb[0] = (w == 0 ? 0.f : rb[0]) + b4.x; rb[1] = (w == 0 ? 0.f : rb[1]) + b4.y;
                rb[2] = (w == 0 ? 0.f : rb[2]) + b4.z; rb[3] = (w == 0 ? 0.f : rb[3]) + b4.w;
            }
            __syncthreads();
        }
        float* brow2 = a.w2slab + (size_t)gridDim.x * (N1 * N1) + (size_t)blockIdx.x * N1;
        for (int i = threadIdx.x; i < N1; i += 256) brow2[i] = rb2[i];
    }
}

template <int E, int MT>
static int launch_ffn1_bwd_t(const Ffn1BwdArgs& a, hipStream_t s) {
    ProfScope prof__(LG_K_FFN1_BWD, s);
    constexpr int N1 = 4 * E, MW = 16 * MT;
    size_t lds = (size_t)(4 * MW * (2 * (N1 + 4) + E + 1) + (E == 16 ? N1 * (N1 + 4) : 0)) * sizeof(float);
    if (ffn1_bwd_fuses_w2(E, a.pre)) lds += (size_t)4 * MW * (N1 + 4) * sizeof(float);   // gelu(h1) rows
    static DeviceOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)k_ffn1_bwd<E, MT, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn1_bwd<E, MT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if constexpr (E == 16) { if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_ffn1_bwd<E, MT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); }
        if (e != hipSuccess) { lg_set_error("ffn1_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
        attr_once.done();
    }
    long per_wg = 4L * MW;
    const long nchunks = (a.P + per_wg - 1) / per_wg;
    const int grid = (int)(nchunks < FFN1_BWD_WGS ? nchunks : FFN1_BWD_WGS);   // persistent workgroups
    if (!a.part) { lg_set_error("ffn1_bwd: partial-sum scratch missing"); return -2; }
    if (ffn1_bwd_fuses_w1(E) && (!a.w1slab || !a.d_w1 || !a.d_b1)) { lg_set_error("ffn1_bwd: dW1 slab / destinations missing"); return -2; }
    if (!ffn1_bwd_fuses_w1(E) && (!a.dh1 || !a.y2)) { lg_set_error("ffn1_bwd: dh1 / y2 outputs missing"); return -2; }
    if (ffn1_bwd_fuses_w2(E, a.pre) && (!a.w2slab || !a.d_w2 || !a.d_b2)) { lg_set_error("ffn1_bwd: dW2 slab / destinations missing"); return -2; }
    if (a.pre && (a.hbf || E != 16)) { lg_set_error("ffn1_bwd: pre-activation saves are fp32, e = 16"); return -2; }
    if (a.hbf) k_ffn1_bwd<E, MT, true, false><<<grid, 256, lds, s>>>(a, nchunks);
    else if (a.pre) { if constexpr (E == 16) k_ffn1_bwd<E, MT, false, true><<<grid, 256, lds, s>>>(a, nchunks); }
    else k_ffn1_bwd<E, MT, false, false><<<grid, 256, lds, s>>>(a, nchunks);
    LG_CHECK_LAUNCH();
    int rc = launch_reduce_slab_pair(a.part, a.part + (size_t)grid * E, grid, E, a.d_ln2g, a.d_ln2b, s);
    if (rc || !ffn1_bwd_fuses_w1(E)) return rc;
    rc = launch_reduce_slab_wb(a.w1slab, a.w1slab + (size_t)grid * N1 * E, grid, N1, E, a.d_w1, E, a.d_b1, s);
    if (rc || !ffn1_bwd_fuses_w2(E, a.pre)) return rc;
    return launch_reduce_slab_wb(a.w2slab, a.w2slab + (size_t)grid * N1 * N1, grid, N1, N1, a.d_w2, N1, a.d_b2, s);
}
int launch_ffn1_bwd(int e, const Ffn1BwdArgs& a, hipStream_t s) {
    if (e == 16) return launch_ffn1_bwd_t<16, 1>(a, s);
    if (e == 32) return (a.w1 && a.wsplit) ? launch_ffn1_bwd_x32(a, a.w1, a.wsplit, s) : launch_ffn1_bwd_t<32, 2>(a, s);
    if (e == 64) return launch_ffn1_bwd_t<64, 1>(a, s);
    lg_set_error("ffn1_bwd: e=%d unsupported", e);
    return -1;
}
