// Weight-gradient GEMMs of the 1x1 convs for gfx950:  dW[n][k] = sum_p Y[p][n] * X[p][k],  db[n] = sum_p Y[p][n]
// (autograd of bmu.point_conv, reference models/common/basic_module_unformer_v2.py:13-14).
// The reduction runs over PIXELS (up to B*H*W = 524 288 at bs=32), the output is tiny (16..256 squared), so the
// pixel axis is the MFMA K dimension: v_mfma_f32_16x16x4_f32 with lane (r,g) feeding A[i=r][k=g] = Y[p+g][n0+r] and
// B[k=g][j=r] = X[p+g][k0+r] -- both operands are read in their natural [pixel][channel] layout, 64-byte segments,
// no transposes.  Every wave owns a 64x64 block of dW over a slice of the pixels; the 4 waves of a workgroup are summed in
// LDS and the workgroup's partial goes to a slab that a second kernel sums in a fixed slice order.
#include "kernels.h"
#include "bwd_kernels.h"
#include "hstore.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));


// Block shape: NT x KT tiles of 16 (compile time, so the pixel loop is branch-free and software-pipelined: the operands of
// the next 16 pixels are in flight while the MFMAs of the current 16 issue).  VEC (64x64 blocks only): lane (r,g) loads ONE
// float4 of Y and ONE of X per 4-pixel step, Y[p+g][n0 + 4r .. 4r+3] / X[p+g][k0 + 4r ..]; element t of the float4 feeds MFMA
// tile t, so tile t owns the strided channel set {4r + t}: a 64-channel row is one fully coalesced 256-byte segment per 16
// lanes and 2 loads feed 16 MFMAs.  Output acc[i][j][v] (MFMA row 4g+v, col r) is then dW[n0 + 4(4g+v) + i][k0 + 4r + j].
// Non-VEC: scalar loads in the natural tile order (tile t = channels 16t .. 16t+15), columns >= n_valid / k_valid read as 0.
// 4-pixel MFMA steps per batch: narrow blocks have few MFMAs per pixel, so they keep more pixels in flight
#ifndef LG_WGRAD_VEC_U
#define LG_WGRAD_VEC_U 4
#endif
#ifndef LG_WGRAD_WGS
#define LG_WGRAD_WGS 512
#endif
#ifndef LG_WGRAD_INTERLEAVE
#define LG_WGRAD_INTERLEAVE 1   // 0: one contiguous pixel range per wave (A/B: the 16-wide blocks run 35 -> 31.5 us interleaved, the 64 x 64 ones the same)
#endif
__host__ __device__ constexpr int wgrad_batch(int nt, int kt, bool vec) { return vec ? LG_WGRAD_VEC_U : (nt + kt <= 2 ? 16 : (nt + kt <= 4 ? 8 : 4)); }

// XGELU: the stored X is a pre-activation and the conv input is gelu(X) (feed_forward's second and third 1x1 convs when the forward
// saved h1 / h3 instead of gelu(h1) / gelu(h3)): evaluated on the operand registers, under the loads of the next batch
template <int NT, int KT, bool VEC, bool YBF, bool XBF, bool XGELU = false>
__global__ __launch_bounds__(256) void k_wgrad_t(WgradArgs a, int k_blocks, long px_per_wave, float* slab, float* bslab) {
    constexpr int U = wgrad_batch(NT, KT, VEC);
    constexpr int NB = 16 * NT, KB = 16 * KT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int nb = blockIdx.y / k_blocks, kb = blockIdx.y - nb * k_blocks;
    const int n0 = nb * NB, k0 = kb * KB;
    const long slice = (long)blockIdx.x * 4 + wave;
#if LG_WGRAD_INTERLEAVE
    // batches of 4 U pixels dealt round-robin to the waves of the grid: at any moment the whole grid reads ONE contiguous window of the
    // operands (all HBM channels), instead of 2 048 streams that all sit at the same offset of their own 64 KiB-aligned chunk
    const long p_stride = (long)gridDim.x * 4 * (4 * U);
    const long p_begin = slice * (4 * U);
    const long p_end = a.P;
#else
    const long p_stride = 4 * U;
    const long p_begin = slice * px_per_wave;
    long p_end = p_begin + px_per_wave;
    if (p_end > a.P) p_end = a.P;
#endif
    f32x4 acc[NT][KT];
    float bsum[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // column offsets / masks of this lane (non-VEC)
    int ycol[NT], xcol[KT];
    bool yok[NT], xok[KT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { const int c = n0 + t * 16 + r; yok[t] = c < a.n_valid; ycol[t] = yok[t] ? c : 0; }
#pragma unroll
    for (int t = 0; t < KT; ++t) { const int c = k0 + t * 16 + r; xok[t] = c < a.k_valid; xcol[t] = xok[t] ? c : 0; }
    // masks are applied by multiplication (operands are finite tensors): a select would let the compiler sink the loads into
    // branches with a full wait each, which is what made the first version of this kernel pure load latency
    float ym[NT], xm[KT];
#pragma unroll
    for (int t = 0; t < NT; ++t) ym[t] = yok[t] ? 1.f : 0.f;
#pragma unroll
    for (int t = 0; t < KT; ++t) xm[t] = xok[t] ? 1.f : 0.f;
    auto fetch = [&](long p, float (&y)[U][NT], float (&x)[U][KT]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = p + 4 * u + g;
            const long rc = row < p_end ? row : p_begin;   // a row that exists; masked in `consume`
            if (VEC) {
                const float4 yv = HS<YBF>::ld4(a.Y, rc * a.ldy + n0 + 4 * r);
                const float4 xv = HS<XBF>::ld4(a.X, rc * a.ldx + k0 + 4 * r);
                const float yy[4] = {yv.x, yv.y, yv.z, yv.w}, xx[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int t = 0; t < NT; ++t) y[u][t] = yy[t & 3];
#pragma unroll
                for (int t = 0; t < KT; ++t) x[u][t] = xx[t & 3];
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) y[u][t] = HS<YBF>::ld1(a.Y, rc * a.ldy + ycol[t]);
#pragma unroll
                for (int t = 0; t < KT; ++t) x[u][t] = HS<XBF>::ld1(a.X, rc * a.ldx + xcol[t]);
            }
        }
    };
    if (p_begin < p_end) {
        float yc[U][NT], xc[U][KT];
        fetch(p_begin, yc, xc);
#pragma unroll 1
        for (long p = p_begin; p < p_end; p += p_stride) {
            float yn[U][NT], xn[U][KT];
            fetch(p + p_stride, yn, xn);   // next batch in flight under this batch's MFMAs (past the slice end: cached, unused)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float rm = (p + 4 * u + g) < p_end ? 1.f : 0.f;
                float yv[NT], xv[KT];
#pragma unroll
                for (int i = 0; i < NT; ++i) { yv[i] = yc[u][i] * (VEC ? rm : rm * ym[i]); bsum[i] += yv[i]; }
                if constexpr (XGELU && KT % 2 == 0) {   // gelu on operand PAIRS (packed fp32, the forward's own gelu2_f): half the vector instructions
#pragma unroll
                    for (int j = 0; j < KT; j += 2) {
                        const lg_v2f gp = gelu2_f((lg_v2f){xc[u][j], xc[u][j + 1]});
                        xv[j] = VEC ? gp.x : gp.x * xm[j];
                        xv[j + 1] = VEC ? gp.y : gp.y * xm[j + 1];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < KT; ++j) {
                        const float xg = XGELU ? gelu_f(xc[u][j]) : xc[u][j];
                        xv[j] = VEC ? xg : xg * xm[j];
                    }
                }
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < KT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[i], xv[j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int i = 0; i < NT; ++i) yc[u][i] = yn[u][i];
#pragma unroll
                for (int j = 0; j < KT; ++j) xc[u][j] = xn[u][j];
            }
        }
    }
    // the 4 waves of the workgroup hold partials of the same block: summed in LDS in a fixed order (no float atomics ->
    // bitwise reproducible), one slab slice per workgroup
    __shared__ float red[NB * KB + NB];
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < KT; ++j)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int idx = VEC ? (4 * (4 * g + v) + i) * KB + 4 * r + j : (i * 16 + 4 * g + v) * KB + j * 16 + r;
                        red[idx] = (w == 0 ? 0.f : red[idx]) + acc[i][j][v];
                    }
            // bias: lanes with equal r hold the same channels for different pixels
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                float sv = bsum[i];
                sv += __shfl_xor(sv, 16);
                sv += __shfl_xor(sv, 32);
                const int idx = NB * KB + (VEC ? 4 * r + i : i * 16 + r);
                if (g == 0) red[idx] = (w == 0 ? 0.f : red[idx]) + sv;
            }
        }
        __syncthreads();
    }
    float* my = slab + (long)blockIdx.x * ((long)a.N * a.K);
    for (int i = threadIdx.x; i < NB * KB; i += 256) {
        const int rr = i / KB, cc = i - rr * KB;
        my[(long)(n0 + rr) * a.K + k0 + cc] = red[i];
    }
    if (a.db && kb == 0 && threadIdx.x < NB) bslab[(long)blockIdx.x * a.N + n0 + threadIdx.x] = red[NB * KB + threadIdx.x];
}

// dst[row*ld + col] += sum_s slab[s][row*cols + col]   for row < rows_valid, col < cols_valid
// block = 64 consecutive outputs x 4 slice phases (fixed summation order: bitwise reproducible)
// block = 64 consecutive outputs x 4 slice phases (fixed summation order: bitwise reproducible)
__global__ __launch_bounds__(256) void k_reduce_slab(const float* __restrict__ slab, long nslices, int rows, int cols, float* dst,
                                                     int ld, int rows_valid, int cols_valid, const float* __restrict__ slab2, int n2,
                                                     float* dst2, int n2_valid) {
    // 16 outputs x 16 slice phases per workgroup: every thread has 8 independent loads in flight (the old 64 x 4 shape walked 128
    // slices per thread and was pure load latency, ~12 us per call); fixed summation order -> deterministic.
    __shared__ float part[16][17];
    // job 0 (blockIdx.y == 0): the [rows][cols] slab; job 1: an optional [n2] vector slab (bias) in the same launch
    if (blockIdx.y == 1) { slab = slab2; rows = 1; cols = n2; dst = dst2; ld = n2; rows_valid = 1; cols_valid = n2_valid; }
    const long n = (long)rows * cols;
    const int o = threadIdx.x & 15, ph = threadIdx.x >> 4;
    const long i = blockIdx.x * 16L + o;
    float sv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) sv[u] = 0.f;
    if (i < n) {
        long k = ph;
        for (; k + 16 * 7 < nslices; k += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) sv[u] += slab[(k + 16 * u) * n + i];
        }
        for (; k < nslices; k += 16) sv[0] += slab[k * n + i];
    }
    part[ph][o] = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
    __syncthreads();
    if (ph == 0 && i < n) {
        const int row = (int)(i / cols), col = (int)(i - (long)row * cols);
        if (row < rows_valid && col < cols_valid) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += part[q][o];
            dst[(long)row * ld + col] += t;
        }
    }
}

// all queued jobs in one launch: blockIdx.y = job, same 16 outputs x 16 slice phases shape as k_reduce_slab
__global__ __launch_bounds__(256) void k_reduce_jobs(ReduceJobTable t) {
    __shared__ float4 part[16][17];
    const ReduceJob& jb = t.j[blockIdx.y];
    const long n = (long)jb.rows * jb.cols;
    const int o = threadIdx.x & 15, ph = threadIdx.x >> 4;
    // round 6: jobs whose rows are whole, 16-byte aligned quads (every large one: pos_emb, the FFN weight slabs) move FOUR outputs per thread -- a
    // 16-lane group reads 256 contiguous bytes of a slice instead of 64, a quarter of the workgroups.  Same order of additions per output.
    const bool v4 = ((jb.cols | jb.row_stride | jb.ld) & 3) == 0 && (jb.slice_stride & 3) == 0 && jb.rows_valid == jb.rows && jb.cols_valid == jb.cols &&
                    ((reinterpret_cast<uintptr_t>(jb.slab) | reinterpret_cast<uintptr_t>(jb.dst) | reinterpret_cast<uintptr_t>(jb.dst2)) & 15) == 0;
    if (v4) {
        const long n4 = n >> 2;
        if (blockIdx.x * 16L >= n4) return;
        const long i4 = blockIdx.x * 16L + o;
        const long i = i4 << 2;
        const int row = (int)(i / jb.cols), col = (int)(i - (long)row * jb.cols);
        float4 sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) sv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i4 < n4) {
            const float* __restrict__ src = jb.slab + (long)row * jb.row_stride + col;
            const long ss = jb.slice_stride, ns = jb.nslices;
            long k = ph;
            for (; k + 16 * 7 < ns; k += 16 * 8) {
                float4 ld[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) ld[u] = *reinterpret_cast<const float4*>(src + (k + 16 * u) * ss);
#pragma unroll
                for (int u = 0; u < 8; ++u) { sv[u].x += ld[u].x; sv[u].y += ld[u].y; sv[u].z += ld[u].z; sv[u].w += ld[u].w; }
            }
            for (; k < ns; k += 16) {
                const float4 l = *reinterpret_cast<const float4*>(src + k * ss);
                sv[0].x += l.x; sv[0].y += l.y; sv[0].z += l.z; sv[0].w += l.w;
            }
        }
        auto add4 = [](const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
        part[ph][o] = add4(add4(add4(sv[0], sv[1]), add4(sv[2], sv[3])), add4(add4(sv[4], sv[5]), add4(sv[6], sv[7])));
        __syncthreads();
        if (ph == 0 && i4 < n4) {
            float4 ts = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 16; ++q) ts = add4(ts, part[q][o]);
            float4* d = reinterpret_cast<float4*>(jb.dst + (long)row * jb.ld + col);
            *d = add4(*d, ts);
            if (jb.dst2) {
                float4* d2 = reinterpret_cast<float4*>(jb.dst2 + (long)row * jb.ld + col);
                *d2 = add4(*d2, ts);
            }
        }
        return;
    }
    if (blockIdx.x * 16L >= n) return;
    const long i = blockIdx.x * 16L + o;
    const int row = (int)(i / jb.cols), col = (int)(i - (long)row * jb.cols);
    float sv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) sv[u] = 0.f;
    if (i < n) {
        const float* __restrict__ src = jb.slab + (long)row * jb.row_stride + col;
        const long ss = jb.slice_stride, ns = jb.nslices;
        long k = ph;
        for (; k + 16 * 7 < ns; k += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) sv[u] += src[(k + 16 * u) * ss];
        }
        for (; k < ns; k += 16) sv[0] += src[k * ss];
    }
    float* parts = reinterpret_cast<float*>(part);
    parts[ph * 17 + o] = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
    __syncthreads();
    if (ph == 0 && i < n && row < jb.rows_valid && col < jb.cols_valid) {
        float tsum = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) tsum += parts[q * 17 + o];
        jb.dst[(long)row * jb.ld + col] += tsum;
        if (jb.dst2) jb.dst2[(long)row * jb.ld + col] += tsum;
    }
}

static thread_local ReduceQueue* tl_rq = nullptr;
void reduce_queue_begin(ReduceQueue* q) { tl_rq = q; }
int reduce_queue_end() {
    ReduceQueue* q = tl_rq;
    tl_rq = nullptr;
    return q ? q->flush() : 0;
}
int ReduceQueue::flush() {
    off = 0;
    if (tab.n == 0) return 0;
    long nmax = 0;
    for (int k = 0; k < tab.n; ++k) {
        const long n = (long)tab.j[k].rows * tab.j[k].cols;
        if (n > nmax) nmax = n;
    }
    dim3 grid((unsigned)((nmax + 15) / 16), (unsigned)tab.n);
#ifdef LG_RQ_DEBUG
    fprintf(stderr, "[rq] flush: %d jobs, grid.x %u:", tab.n, grid.x);
    for (int k = 0; k < tab.n; ++k) fprintf(stderr, " (%ld sl x %d x %d)", tab.j[k].nslices, tab.j[k].rows, tab.j[k].cols);
    fprintf(stderr, "\n");
#endif
    k_reduce_jobs<<<grid, 256, 0, stream>>>(tab);
    tab.n = 0;
    LG_CHECK_LAUNCH();
    return 0;
}
int ReduceQueue::push(const ReduceJob& j) {
    if (tab.n == LG_MAX_REDUCE_JOBS) {
        // the slabs already handed out stay valid: only the table is drained (off is restored)
        const size_t keep = off;
        int rc = flush();
        off = keep;
        if (rc) return rc;
    }
    tab.j[tab.n++] = j;
    return 0;
}
float* ReduceQueue::take(size_t nfloats) {
    nfloats = (nfloats + 63) & ~(size_t)63;
    if (nfloats > cap) { lg_set_error("reduce queue: slab of %zu floats exceeds the arena (%zu)", nfloats, cap); return nullptr; }
    if (off + nfloats > cap && flush()) return nullptr;
    float* p = arena + off;
    off += nfloats;
    return p;
}

static int launch_reduce_slab2(const float* slab, long nslices, int rows, int cols, float* dst, int ld, int rows_valid, int cols_valid,
                               const float* slab2, int n2, float* dst2, int n2_valid, hipStream_t s) {
    if (tl_rq) {
        ReduceJob j;
        j.slab = slab; j.dst = dst; j.dst2 = nullptr; j.nslices = nslices; j.slice_stride = (long)rows * cols;
        j.rows = rows; j.cols = cols; j.row_stride = cols; j.ld = ld; j.rows_valid = rows_valid; j.cols_valid = cols_valid;
        int rc = tl_rq->push(j);
        if (rc || !slab2) return rc;
        j.slab = slab2; j.dst = dst2; j.slice_stride = n2; j.rows = 1; j.cols = n2; j.row_stride = n2; j.ld = n2; j.rows_valid = 1;
        j.cols_valid = n2_valid;
        return tl_rq->push(j);
    }
    long n = (long)rows * cols;
    dim3 grid((unsigned)((n + 15) / 16), slab2 ? 2 : 1);
    k_reduce_slab<<<grid, 256, 0, s>>>(slab, nslices, rows, cols, dst, ld, rows_valid, cols_valid, slab2, n2, dst2, n2_valid);
    LG_CHECK_LAUNCH();
    return 0;
}
int launch_reduce_job(const ReduceJob& j, hipStream_t s) {
    if (tl_rq) return tl_rq->push(j);
    ReduceJobTable t;
    t.j[0] = j;
    t.n = 1;
    const long n = (long)j.rows * j.cols;
    k_reduce_jobs<<<dim3((unsigned)((n + 15) / 16), 1), 256, 0, s>>>(t);
    LG_CHECK_LAUNCH();
    return 0;
}
// per-channel partial rows (k_dw_bwd / k_dstep_top_bwd) as queue jobs; false when no queue is active
bool reduce_chan_enqueue(const float* part, const ChanReduce& m, int* rc) {
    if (!tl_rq) return false;
    *rc = 0;
    for (int k = 0; k < m.NK && !*rc; ++k) {
        ReduceJob j;
        j.slab = part + k; j.dst = m.dst[k]; j.dst2 = m.dst2[k]; j.cols = 1; j.cols_valid = 1;
        if ((m.allc_mask >> k) & 1u) {
            j.nslices = (long)m.nslices * m.C; j.slice_stride = m.NK; j.rows = 1; j.row_stride = 0; j.ld = 1; j.rows_valid = 1;
        } else {
            j.nslices = m.nslices; j.slice_stride = (long)m.C * m.NK; j.rows = m.C; j.row_stride = m.NK; j.ld = m.stride[k]; j.rows_valid = m.C;
        }
        *rc = tl_rq->push(j);
    }
    return true;
}
// [nslices][rows*cols] matrix slab + [nslices][rows] bias slab -> dW[rows][ld] , db[rows] (+=)
int launch_reduce_slab_wb(const float* wslab, const float* bslab, long nslices, int rows, int cols, float* dW, int ld, float* db, hipStream_t s) {
    return launch_reduce_slab2(wslab, nslices, rows, cols, dW, ld, rows, cols, bslab, rows, db, rows, s);
}
// two [nslices][n] vector slabs -> dst_a[n], dst_b[n] (+=) in one launch
int launch_reduce_slab_pair(const float* slab_a, const float* slab_b, long nslices, int n, float* dst_a, float* dst_b, hipStream_t s) {
    return launch_reduce_slab2(slab_a, nslices, 1, n, dst_a, n, 1, n, slab_b, n, dst_b, n, s);
}
int launch_reduce_slab(const float* slab, long nslices, int rows, int cols, float* dst, int ld, int rows_valid, int cols_valid,
                       hipStream_t s) {
    return launch_reduce_slab2(slab, nslices, rows, cols, dst, ld, rows_valid, cols_valid, nullptr, 0, nullptr, 0, s);
}

static int tiles_per_block(int n16) {   // largest of 4,3,2,1 dividing the tile count
    for (int t = 4; t > 1; --t) if (n16 % t == 0) return t;
    return 1;
}
static long wgrad_splits(int N, int K) {
    const int blocks = (N / 16 / tiles_per_block(N / 16)) * (K / 16 / tiles_per_block(K / 16));
    long splits = LG_WGRAD_WGS / blocks;   // two waves per SIMD
    return splits < 1 ? 1 : splits;
}
size_t wgrad_slab_floats(int N, int K, long P) {
    // sized for the launch geometry below (upper bound)
    (void)P;
    return (size_t)wgrad_splits(N, K) * ((size_t)N * K + N);
}

template <int NT, int KT, bool VEC>
static void wgrad_dispatch_bf(const WgradArgs& a, dim3 grid, int k_blocks, long px, float* slab, float* bslab, hipStream_t s) {
    if constexpr (KT == 4) {
        if (a.xgelu) { k_wgrad_t<NT, KT, VEC, false, false, true><<<grid, 256, 0, s>>>(a, k_blocks, px, slab, bslab); return; }
    }
    if (a.ybf && a.xbf) k_wgrad_t<NT, KT, VEC, true, true><<<grid, 256, 0, s>>>(a, k_blocks, px, slab, bslab);
    else if (a.ybf) k_wgrad_t<NT, KT, VEC, true, false><<<grid, 256, 0, s>>>(a, k_blocks, px, slab, bslab);
    else if (a.xbf) k_wgrad_t<NT, KT, VEC, false, true><<<grid, 256, 0, s>>>(a, k_blocks, px, slab, bslab);
    else k_wgrad_t<NT, KT, VEC, false, false><<<grid, 256, 0, s>>>(a, k_blocks, px, slab, bslab);
}
template <int NT>
static void wgrad_dispatch_kt(int KT, const WgradArgs& a, dim3 grid, int k_blocks, long px, float* slab, float* bslab, hipStream_t s) {
    switch (KT) {
        case 1: wgrad_dispatch_bf<NT, 1, false>(a, grid, k_blocks, px, slab, bslab, s); break;
        case 2: wgrad_dispatch_bf<NT, 2, false>(a, grid, k_blocks, px, slab, bslab, s); break;
        case 3: wgrad_dispatch_bf<NT, 3, false>(a, grid, k_blocks, px, slab, bslab, s); break;
        default: wgrad_dispatch_bf<NT, 4, false>(a, grid, k_blocks, px, slab, bslab, s); break;
    }
}

int launch_wgrad(const WgradArgs& a, float* slab, hipStream_t s) {
    ProfScope prof__(LG_K_WGRAD, s);
    if ((a.N & 15) || (a.K & 15) || a.N <= 0 || a.K <= 0 || a.P <= 0) { lg_set_error("wgrad: N,K must be positive multiples of 16"); return -2; }
    if (a.n_valid > a.N || a.k_valid > a.K || a.n_valid > a.ldy || a.k_valid > a.ldx) { lg_set_error("wgrad: valid extents exceed the operands"); return -2; }
    const int NT = tiles_per_block(a.N / 16), KT = tiles_per_block(a.K / 16);
    if (a.xgelu && (KT != 4 || a.ybf || a.xbf)) { lg_set_error("wgrad: gelu(X) operands need fp32 storage and K a multiple of 64"); return -2; }
    const int n_blocks = a.N / (16 * NT), k_blocks = a.K / (16 * KT);
    const int blocks = n_blocks * k_blocks;
    long splits = wgrad_splits(a.N, a.K);
    long nslices = splits * 4;
    long px = (a.P + nslices - 1) / nslices;
    const bool vec = NT == 4 && KT == 4 && a.n_valid == a.N && a.k_valid == a.K && (a.ldy & 3) == 0 && (a.ldx & 3) == 0;
    const long batch = 4L * wgrad_batch(NT, KT, vec);
    px = (px + batch - 1) / batch * batch;   // whole batches
    // shrink the slice count if the tensor is small
    nslices = (a.P + px - 1) / px;
    splits = (nslices + 3) / 4;
    float* bslab = slab + splits * (long)a.N * a.K;
    dim3 grid((unsigned)splits, (unsigned)blocks);
    if (vec) wgrad_dispatch_bf<4, 4, true>(a, grid, k_blocks, px, slab, bslab, s);
    else switch (NT) {
        case 1: wgrad_dispatch_kt<1>(KT, a, grid, k_blocks, px, slab, bslab, s); break;
        case 2: wgrad_dispatch_kt<2>(KT, a, grid, k_blocks, px, slab, bslab, s); break;
        case 3: wgrad_dispatch_kt<3>(KT, a, grid, k_blocks, px, slab, bslab, s); break;
        default: wgrad_dispatch_kt<4>(KT, a, grid, k_blocks, px, slab, bslab, s); break;
    }
    LG_CHECK_LAUNCH();
    return launch_reduce_slab2(slab, splits, a.N, a.K, a.dW, a.ldw, a.n_valid, a.k_valid, a.db ? bslab : nullptr, a.N, a.db, a.n_valid, s);
}
