// Launcher declarations of the backward kernels (autograd of the forward kernels in kernels.h).
#pragma once
#include "common.h"

// ---------------- generic 1x1-conv weight gradient ----------------
struct WgradArgs {
    const void* Y;   // [P, ldy]  gradient of the conv output (fp32, or bf16 hidden storage when ybf)
    const void* X;   // [P, ldx]  conv input (fp32, or bf16 hidden storage when xbf)
    float* dW;       // [N][ldw] accumulated (+=)
    float* db;       // [N] accumulated (+=), nullable
    long P;
    int ldy, ldx, ldw;
    int N, K;              // multiples of 16 (padded)
    int n_valid, k_valid;  // rows / cols of dW that exist
    int ybf, xbf;          // operand storage: 0 fp32, 1 bf16 (hstore.h)
    int xgelu;             // 1: the conv input is gelu(X) of the stored pre-activation X (fp32 storage, 64-wide K blocks)
};
size_t wgrad_slab_floats(int N, int K, long P);
// Parameter-gradient partial sums: every workgroup writes its partial to its own row of a scratch slab and a second tiny
// kernel sums the rows in a fixed order.  (Float atomics onto the handful of parameter addresses serialise in L2 at
// ~40 ns each: 2048 workgroups x one address was an 80 us tail on a 20 us kernel -- and the result was order-dependent.)
#define PIXEL_PART_WGS 2048   // grid cap of the per-pixel backward kernels that emit partial rows
size_t chan_partial_floats(int C, int B, int H, int W);   // k_dw_bwd / k_dstep_top_bwd scratch (upper bound)
// dst_k[c * stride_k] (+)= sum over slices of part[(slice * C + c) * NK + k]; k in allc_mask: summed over c as well (dst_k[0])
struct ChanReduce {
    float* dst[14];
    float* dst2[14];   // optional second destination of the same sum
    int stride[14];
    int NK, C, nslices;
    unsigned allc_mask;
};
int launch_reduce_chan(const float* part, const ChanReduce& m, hipStream_t s);
int launch_reduce_slab_pair(const float* slab_a, const float* slab_b, long nslices, int n, float* dst_a, float* dst_b, hipStream_t s);

// Deferred reductions: while a ReduceQueue is active on the calling thread (reduce_queue_begin), launch_reduce_slab* /
// launch_reduce_chan only RECORD their job; flush() sums all recorded slabs in ONE launch (blockIdx.y = job).  A backward
// pass has ~60 of these tiny reductions (5 us of pure latency each).  Slabs must then stay untouched until the flush, so
// producers take them from the queue's arena (take() flushes by itself when the arena or the job table is full).
struct ReduceJob {
    const float* slab;   // element (slice s, row r, col c) at slab[s * slice_stride + r * row_stride + c]
    float* dst;          // dst[r * ld + c] += sum_s ...   for r < rows_valid, c < cols_valid
    float* dst2;         // optional second destination
    long nslices, slice_stride;
    int rows, cols, row_stride, ld, rows_valid, cols_valid;
};
#define LG_MAX_REDUCE_JOBS 56
struct ReduceJobTable {
    ReduceJob j[LG_MAX_REDUCE_JOBS];
    int n;
};
struct ReduceQueue {
    float* arena;
    size_t cap, off;   // floats
    hipStream_t stream;
    ReduceJobTable tab;
    void init(float* base, size_t cap_floats, hipStream_t s) { arena = base; cap = cap_floats; off = 0; stream = s; tab.n = 0; }
    float* take(size_t nfloats);   // nullptr (error set) when nfloats > cap
    int push(const ReduceJob& j);
    int flush();
};
int launch_reduce_slab_wb(const float* wslab, const float* bslab, long nslices, int rows, int cols, float* dW, int ld, float* db, hipStream_t s);
// one job: recorded when a queue is active on this thread, otherwise summed right away on `s`
int launch_reduce_job(const ReduceJob& j, hipStream_t s);
bool reduce_chan_enqueue(const float* part, const ChanReduce& m, int* rc);
void reduce_queue_begin(ReduceQueue* q);
int reduce_queue_end();   // flushes and deactivates
int launch_wgrad(const WgradArgs& a, float* slab, hipStream_t s);
int launch_reduce_slab(const float* slab, long nslices, int rows, int cols, float* dst, int ld, int rows_valid, int cols_valid,
                       hipStream_t s);

// ---------------- data module backward ----------------
struct DwBwdArgs {
    const float* gout;  // [planes, n, n] gradient wrt the dw-conv output
    const float* in;    // conv input before resampling: [planes, hi, wi]; conv input = resample<MODE>(in)
    float* gin;         // [planes, n, n] gradient wrt the resampled conv input (dw^T gout)
    const float* w9;    // [C,1,3,3]
    float* dw9;         // grads (+=)
    float* dbias;
    float* part;        // scratch [workgroups][10] per-workgroup partial sums (chan_partial_floats)
    int C, planes, hi, wi, n_h, n_w;  // n_h x n_w = conv resolution
};
int launch_dw_bwd(int mode, const DwBwdArgs& a, hipStream_t s);
struct DstepTopArgs {
    const float* g;    // dZ' [B,C,H,W]
    const float* s1;   // [B,C,H/2,W/2]
    const float* z;    // Z_i [B,C,H,W]
    const float* pan;  // [B,1,H,W]
    float* gu;         // out: grad wrt up(s1)  [B,C,H,W]
    float* dz;         // out: direct part of dZ_i = g + Rw[c]*dpr
    const float *w9, *b9, *rw, *rb, *rtw, *rtb, *eta;
    float *dw9, *dbias, *drw, *drb, *drtw, *drtb, *deta;
    float* part;       // scratch [workgroups][14] per-workgroup partial sums (chan_partial_floats)
    int C, B, H, W;
};
int launch_dstep_top_bwd(const DstepTopArgs& a, hipStream_t s);
// the data step's backward as one pixelwise launch + one plane-in-LDS launch (k_dstep.hip; planes dstep_fused_ok accepts)
struct DstepPreBwdArgs {
    const float *z, *g, *pan;
    float* dz;
    const float *rw, *rb, *rtw, *rtb, *eta;
    float* part;   // [workgroups][C][5]
    int hw4;
};
struct DstepBwdArgs {
    const float* g;      // dZ' [B,C,N,N]
    const float* z;      // Z_i [B,C,N,N]
    const float* pan;    // [B,1,N,N]
    const float *t1, *r, *s1;   // the forward's intermediates
    float* dz;           // out: dZ_i [B,C,N,N]  (must not alias g)
    const float *d1w, *d3w, *dt1w, *dt3w, *dt3b, *rw, *rb, *rtw, *rtb, *eta;
    float *part_top, *part_dt1, *part_d3, *part_d1, *part_pre;   // partial rows: [B*C][14], 3 x [B*C][10], [workgroups of the pixelwise kernel][C][5]
    int B, C, N;
};
struct DstepBwdGrads { float *d1w, *d1b, *d3w, *d3b, *dt1w, *dt1b, *dt3w, *dt3b, *rw, *rb, *rtw, *rtb, *eta; };   // (+=)
size_t dstep_bwd_part_floats(int C, int B, int N);
int launch_dstep_bwd(const DstepBwdArgs& a, const DstepBwdGrads& g, hipStream_t s);
// adjoint of the bicubic resampler: gin[planes, hi, wi] (= or +=) R^T gout[planes, ho, wo]; mode 0: x0.5, 1: x2
int launch_resample_adj(int mode, const float* gout, float* gin, int planes, int hi, int wi, int accumulate, hipStream_t s);

// ---------------- LGT pixelwise backward ----------------
struct TailBwdArgs {
    const float* dout;  // [B,C,H,W]
    const float* x;     // [P,E] the last block's output (the conv input)
    float* dx;          // [P,E] = Wt^T dout
    float* dz;          // [B,C,H,W] = dout (residual path)
    const float* w;
    float *d_w, *d_b;   // the conv's weight [C][E] / bias [C] gradients, accumulated in the kernel (+=)
    float* part;        // scratch: per-workgroup partial sums, tail_bwd_part_floats(C) floats
    int HW;
    long total;
};
size_t tail_bwd_part_floats(int C);
int launch_tail_bwd(int C, const TailBwdArgs& a, hipStream_t s);

struct EmbedBwdArgs {
    const float* dx;  // [P,E] grad wrt embed output
    const float* z;   // [B,C,H,W]
    float* dz;        // [B,C,H,W]  += dt * dww
    const float *dww, *dwb, *w, *b, *lng;
    float *d_dww, *d_dwb, *d_lng, *d_lnb;
    float *d_w, *d_b; // the 1x1 conv's weight [E][C] / bias [E] gradients, accumulated in the kernel (+=)
    float* part;      // scratch: per-workgroup partial sums, embed_bwd_part_floats(C) floats
    int HW;
    long total;
};
size_t embed_bwd_part_floats(int C);
int launch_embed_bwd(int C, const EmbedBwdArgs& a, hipStream_t s);

struct DownBwdArgs {
    const float* dy;     // [B,H/2,W/2,2E] grad wrt down output
    float* du;           // [B,H/2,W/2,E]  = Wd^T dy
    const float* dskip;  // [B,H,W,E] contribution from the fusion conv
    float* dx;           // [B,H,W,E] = dskip + down2^T(du)
    const float* w;
    int B, H, W;
};
int launch_down_bwd_a(int E, const DownBwdArgs& a, hipStream_t s);
int launch_down_bwd_b(int E, const DownBwdArgs& a, hipStream_t s);

struct UpFuseBwdArgs {
    const float* dy;  // [B,H,W,E] grad wrt fusion output
    float* dt;        // [B,H,W,E] grad wrt the up-path (post 1x1) tensor
    float* dskip;     // [B,H,W,E]
    float* v;         // [B,H/2,W/2,E] = up2^T(dt)
    float* dxb;       // [B,H/2,W/2,2E] = Wu^T v
    float* tmp;       // scratch [B,H,W/2,E]: dt contracted along x (k_upadj_h)
    const float *fw, *upw;
    int B, H, W;
};
int launch_upfuse_bwd_a(int E, const UpFuseBwdArgs& a, hipStream_t s);
int launch_upfuse_bwd_b(int E, const UpFuseBwdArgs& a, hipStream_t s);

// ---------------- feed_forward backward ----------------
struct FfnDwBwdArgs {
    const float* dy;   // [B,h,w,e]  grad wrt block output
    const void* g3;    // [B,h,w,4e] saved gelu'(h3)   (hidden storage)
    const void* h2;    // [B,h,w,4e] saved              (hidden storage)
    void* dh2;         // [B,h,w,4e] out: dw3x3^T ((dy W3) * g3)   (hidden storage)
    int hbf;           // hidden storage: 0 fp32, 1 bf16
    int pre;           // 1: g3 holds the PRE-ACTIVATION h3 (forward saved h1 / h2 / h3 only); gelu'(h3) is evaluated in the kernel
    const float *w3t;  // [4e][e] transposed W3
    const float* dww;  // [4e,1,3,3]
    float *slab_w, *slab_b;  // [tiles][4e*9], [tiles][4e] partials
    float *d_dww, *d_dwb;
    int B, h, w;
};
size_t ffn_dw_bwd_slab_floats(int e, int B, int h, int w);
int launch_ffn_dw_bwd(int e, const FfnDwBwdArgs& a, hipStream_t s);
struct Ffn1BwdArgs {
    const void* dh2;   // [P,4e]  (hidden storage)
    const void* g1;    // [P,4e] saved gelu'(h1)  (hidden storage)
    const float* x;    // [P,e] block mid activation (LN2 input)
    const float* dy;   // [P,e] grad wrt block output (residual path)
    void* dh1;         // [P,4e] (wgrad operand, hidden storage)
    float* y2;         // [P,e] LN2(x) (wgrad operand)
    float* dx;         // [P,e] grad wrt x
    const float *w2t, *w1t, *ln2g, *ln2b;
    float *d_ln2g, *d_ln2b;
    float* part;        // scratch: per-workgroup LN2 partial sums, PIXEL_PART_WGS * 2e floats
    // e <= 32: the first conv's weight / bias gradient (dW1 = sum_p dh1 (x) LN2(x), db1 = sum_p dh1) is accumulated in this
    // kernel (dh1 and LN2(x) are already on chip), so dh1 / y2 never go to HBM: w1slab = FFN1_BWD_WGS * (4e*e + 4e) floats
    float *w1slab, *d_w1, *d_b1;
    // e = 16 with pre-activation saves: dW2 / db2 as well (gelu(h1) is evaluated here): w2slab = FFN1_BWD_WGS * (4e*4e + 4e) floats
    float *w2slab, *d_w2, *d_b2;
    long P;
    int hbf;           // hidden storage: 0 fp32, 1 bf16
    int pre;           // 1: g1 holds the PRE-ACTIVATION h1; gelu'(h1) is evaluated in the kernel
    // e = 32: the split-bf16 kernel k_ffn1_bwd_x32 runs when both are set (w1 = the forward W1 [4e][e], wsplit = ffn_wsplit_bytes(32)
    // bytes of scratch for the pre-split W2^T / W1^T fragments); null -> the f32-MFMA kernel k_ffn1_bwd<32>
    const float* w1;
    void* wsplit;
};
#define FFN1_BWD_WGS 512   // persistent grid cap of k_ffn1_bwd
inline bool ffn1_bwd_fuses_w1(int e) { return e <= 32; }
#ifndef LG_FW2
#define LG_FW2 0   // 1: k_ffn1_bwd accumulates dW2 itself in the pre-activation mode (64 extra f32 MFMAs per chunk: 7.51 ms against 7.43 with dW2 from k_wgrad_t, gelu evaluated on its X operand)
#endif
inline bool ffn1_bwd_fuses_w2(int e, int pre) { return LG_FW2 && pre && e == 16; }
int launch_ffn1_bwd(int e, const Ffn1BwdArgs& a, hipStream_t s);
int launch_ffn1_bwd_x32(const Ffn1BwdArgs& a, const float* w1, void* wsplit, hipStream_t s);   // k_ffn1_bwd_x32.hip
// e = 16 | 32 (k_ffn_bwd_x.hip): everything that hangs off dh2 in ONE pass on the bf16 matrix pipe (split arithmetic) --
// h1 re-computed from x, dx, LayerNorm gradients, dW1 / db1 AND dW2 / db2; replaces k_ffn1_bwd<16> / k_ffn1_bwd_x32 + the 4e x 4e k_wgrad_t launches
// (N1 = 4 e below)
struct Ffn1BwdXArgs {
    const void* dh2;   // [P,N1]  (hidden storage: fp32, or bf16 when hbf)
    int hbf;           // 1: precision = 'bf16' (plain bf16 products, dh2 stored as bf16)
    const float* x;    // [P,e] block mid activation (LN2 input)
    const float* dy;   // [P,e] grad wrt block output (residual path)
    float* dx;         // [P,e]
    const float *w1, *b1;      // forward W1 [N1][e], b1 [N1]
    const float *w2t, *w1t;    // W2^T [N1][N1], W1^T [e][N1]
    const float *ln2g, *ln2b;
    float* slab;       // ffn1_bwd_x_slab_floats(e) floats (per-workgroup partial sums)
    float *d_w1, *d_b1, *d_w2, *d_b2, *d_ln2g, *d_ln2b;   // accumulated (+=) by the deferred reduce launch
    long P;            // multiple of 64
    const float* scales = nullptr;   // the block's row of NetBufs::ffn_scales (k_ffn_prep.hip) + in [6] max |dh2| of this backward (k_ffn_dw_bwd_xs): f16-pair
                                     // products (NP = 2) when set; null: bf16 triples
};
inline int ffn1_bwd_x_wgs(int e) { return e == 16 ? 512 : 256; }   // persistent grid: two workgroups per CU at e = 16 (55 KB of LDS), one at e = 32 (139 KB)
size_t ffn1_bwd_x_slab_floats(int e);                               // floats of Ffn1BwdXArgs::slab
bool ffn1_bwd_x32_built();                                          // the e = 32 instance is compiled (make AB=1)
int launch_ffn1_bwd_xs(int e, const Ffn1BwdXArgs& a, hipStream_t s);   // e = 16 | 32
// e = 16 | 32 (k_ffn_dwbwd_x.hip): the strip-walking spatial half -- dh3 in an LDS ring, dh2 out, depthwise gradients AND
// dW3 / db3 in the same pass; replaces k_ffn_dw_bwd<16> + the 16 x 64 k_wgrad_t launch
struct FfnDwBwdXArgs {
    const float* dy;   // [B,h,w,16]
    const void* h3;    // [B,h,w,64] saved pre-activation of the second GELU   (hidden storage: fp32, or bf16 when hbf)
    const void* h2;    // [B,h,w,64] saved                                      (hidden storage)
    void* dh2;         // [B,h,w,64] out                                        (hidden storage)
    int hbf;           // 1: precision = 'bf16' (plain bf16 products, bf16 storage of h2 / h3 / dh2)
    const float* w3t;  // [4e][e] transposed W3
    const float* dww;  // [4e,1,3,3]
    const float* dwb = nullptr;   // [4e] depthwise bias: k_ffn_dw_bwd_h only (it re-computes h3 = dw3x3(h2) + b; h3 is then nullptr)
    float* slab;       // FFN_DW_BWD_X_WGS rows of FFN_DW_BWD_X_ROW floats (per-workgroup partial sums)
    float *d_dww, *d_dwb, *d_w3, *d_b3;   // accumulated (+=) by the deferred reduce launch
    int B, h, w;
    float* dh2_max = nullptr;   // optional: max |dh2| of the launch is atomically max-ed into this word (float bits; zeroed by k_ffn_scales every forward)
};
#define FFN_DW_BWD_X_WGS 512      // e = 16: two resident workgroups per CU
#define FFN_DW_BWD_X_DB 576
#define FFN_DW_BWD_X_W3 640
#define FFN_DW_BWD_X_B3 1664
#define FFN_DW_BWD_X_ROW 1680
#define FFN_DW_BWD_X32_ROW 2720   // e = 32: [d dww 64 x 9 | d dwb 64 | dW3 32 x 64 | db3 32] per channel half; 256 workgroups per half
inline size_t ffn_dw_bwd_x_slab_floats(int e) { return (size_t)FFN_DW_BWD_X_WGS * (e == 16 ? FFN_DW_BWD_X_ROW : FFN_DW_BWD_X32_ROW); }
int launch_ffn_dw_bwd_xs(int e, const FfnDwBwdXArgs& a, hipStream_t s);   // e = 16 | 32
// round 6, e = 16: the same half WITHOUT a saved h3 (re-computed from an LDS ring of h2; two channel halves of 32): k_ffn_dwbwd_h.hip
#define FFN_DW_BWD_H_WGS 512
#define FFN_DW_BWD_H_ROW 848      // [d dww 32 x 9 | d dwb 32 | dW3 16 x 32 | db3 16] per workgroup and channel half
inline size_t ffn_dw_bwd_h_slab_floats() { return (size_t)2 * FFN_DW_BWD_H_WGS * FFN_DW_BWD_H_ROW; }
int launch_ffn_dw_bwd_h(const FfnDwBwdXArgs& a, hipStream_t s);
int launch_transpose(const float* src, float* dst, int rows, int cols, hipStream_t s);  // dst[cols][rows]
int launch_transpose3(const float* const* src, float* const* dst, const int* rows, const int* cols, int njobs, hipStream_t s);

// ---------------- mixer backward ----------------
struct ProjO2BwdArgs {
    const float* dy;  // [P,e] grad wrt mixer-half output (Xmid)
    float* do2;       // [B,e/2,h,w] planar grad wrt global-mixer output
    float* dym;       // optional [P,e]: dropout-masked dy (wgrad operand); null when no dropout
    const float* projw;
    int HW;
    long total;
    int dropout;
    uint64_t seed;
};
int launch_proj_o2_bwd(int e, const ProjO2BwdArgs& a, hipStream_t s);

struct FftBwdArgs {
    const float* do2;  // [planes,n,n]
    const float* sgn;  // [planes,n,n] sign of the irfft2 output (saved)
    const float* amp;  // [planes,n,n/2+1]
    const float* pha;
    float* dg;         // [planes,n,n]
    float* scratch;    // n > 128 only: half-spectrum scratch
    const float *ampw, *ampb, *phaw, *phab;
    float *d_ampw, *d_ampb, *d_phaw, *d_phab;
    float* part;       // scratch: per-workgroup partial sums of the four parameter gradients, fft_bwd_part_floats()
    int planes, ch, n;
    int h, w;          // plane size (when non-zero; n = side of a square plane otherwise)
    int full = 0;      // 1: the complex-row in-LDS kernels (A/B variant LG_VAR_FFT_FULL); 0: the real-input kernels (k_fftmix_r / k_fftmix_bwd_r)
};
size_t fft_bwd_part_floats(int planes, int h, int w);
int launch_fftmix_bwd(const FftBwdArgs& a, hipStream_t s);

struct AttnBwdArgs {
    const float* x;     // [P,e] block input
    const float* dy;    // [P,e] grad wrt Xmid
    const float* dym;   // masked dy (== dy when no dropout)
    const float* o2;    // planar global-mixer output
    const float* dg;    // planar grad wrt LN1(x)[e/2:]
    float* dx;          // [P,e]
    float* cat;         // [P,e]   proj input (wgrad operand)
    float* y1;          // [P,e/2] to_qkv input (wgrad operand)
    float* dqkv;        // [P,3e/2] grad wrt to_qkv output (wgrad operand)
    const float* pos;   // [2,64,64]
    const float* posT;  // [2,64,64] transposed
    float* dpos_slab;   // [grid][2*64*64] partial pos_emb grads
    const float *ln1g, *ln1b, *qkvw, *qkvb, *projw;
    float *d_ln1g, *d_ln1b;
    float *d_qkvw, *d_qkvb;   // e = 16 (attn_bwd_fuses_qkv): to_qkv's weight / bias gradients are accumulated in the epilogue kernel (+=); y1 may be null
    float* part;        // scratch: per-workgroup partial sums, attn_bwd_part_floats(e) floats
    int B, h, w;
    const float* so = nullptr;   // [P,e/2] the forward's attention output (pre-proj, head-major) and
    const float* sl = nullptr;   // [P,2] log-sum-exp of its score rows (log2 domain), left by k_attn_m's saving launch (round 6); both null: k_attn_bwd_core re-derives them
    float* stats = nullptr;   // k_attn_bwd_core_m only: [P][2 heads][4] row statistics between its two launches
    int core_m = 0;     // 1 (lg_config.variant LG_VAR_ATTN_BWD_CORE_M): the matrix-pipe core k_attn_bwd_core_m at e = 32 instead of the vector-pipe k_attn_bwd_core
};
int launch_attn_bwd_core_m(int e, const AttnBwdArgs& a, int grid, int nwin, int ngroups, hipStream_t s);   // k_attn_bwd_m.hip
#ifndef LG_ATTN_FUSE_QKV
#define LG_ATTN_FUSE_QKV 1
#endif
constexpr bool attn_bwd_fuses_qkv(int e) { return LG_ATTN_FUSE_QKV && e == 16; }
inline size_t attn_bwd_part_floats(int e) { return (size_t)PIXEL_PART_WGS * 2 * e + (attn_bwd_fuses_qkv(e) ? (size_t)1024 * ((3 * e / 2) * (e / 2) + 3 * e / 2) : 0); }
int attn_bwd_grid(int e, int B, int h, int w);
int launch_attn_bwd(int e, const AttnBwdArgs& a, hipStream_t s);

// ---- round 4: the whole local-mixer half-block backward in one kernel (k_attn_bwd_f.hip; e = 16) ----
// k_proj_o2_bwd_k runs first (the FFT-mixer backward needs do2): do2, ONE dropout keep-bit word per pixel instead of the masked copy of
// dy, and the proj bias gradient; then k_attn_bwd_f: flash passes, to_qkv^T, LayerNorm backward, dx and every parameter gradient of the
// half-block (pos_emb, to_qkv, proj, LayerNorm) without a pixel-sized intermediate in HBM.
struct ProjO2BwdKArgs {
    const float* dy;   // [P,e] grad wrt the mixer half-block output
    float* do2;        // [B,e/2,h,w] planar grad wrt the global-mixer output
    uint32_t* keep;    // [P] out: dropout keep bits (bit n = channel n kept); null = no dropout
    const float* projw;
    float* slab;       // PROJ_O2_K_WGS rows of e floats: proj bias gradient partials
    float* d_projb;    // += by the deferred reduce launch
    int HW;
    long total;
    uint64_t seed;
};
#ifndef PROJ_O2_K_WGS
#define PROJ_O2_K_WGS 512   // 1 024 workgroups: the same kernel time, but the bias-gradient reduce job behind it walks twice the slices (step 6.31 -> 6.285 ms)
#endif
int launch_proj_o2_bwd_k(int e, const ProjO2BwdKArgs& a, hipStream_t s);
struct AttnBwdFArgs {
    const float* x;        // [P,e] block input
    const float* dy;       // [P,e] grad wrt the mixer half-block output (unmasked)
    const uint32_t* keep;  // [P] dropout keep bits written by k_proj_o2_bwd_k; null = no dropout
    const float* o2;       // planar global-mixer output (proj input, for the proj weight gradient)
    const float* dg;       // planar grad wrt LN1(x)[e/2:] (FFT-mixer backward)
    float* dx;             // [P,e]
    const float* pos;      // [2,64,64]
    const float *ln1g, *ln1b, *qkvw, *qkvb, *projw;
    const float* so = nullptr;   // [P,e/2] the forward's attention output (pre-proj, head-major) and
    const float* sl = nullptr;   // [P,2] log-sum-exp of its score rows (log2 domain), saved by k_attn_m's saving launch (round 6); both null: recomputed here
    float* slab;           // ATTN_BWD_F_WGS rows of ATTN_BWD_F_ROW floats: per-workgroup partial sums
    float *d_pos, *d_qkvw, *d_qkvb, *d_projw, *d_ln1g, *d_ln1b;   // += by the deferred reduce launch
    int B, h, w;
    unsigned rcp_nwx, rcp_nwy;   // filled by the launcher: 2^32 / (w / 8) + 1, 2^32 / (h / 8) + 1
};
#define ATTN_BWD_F_WGS 256
#define ATTN_BWD_F_WQ 8192            // row: pos_emb [2][64][64] | dWqkv [24][8] | dbqkv [24] | dWproj [16][16] | d gamma [16] | d beta [16]
#define ATTN_BWD_F_ROW (8192 + 192 + 24 + 256 + 32)
inline bool attn_bwd_fused(int e) { return e == 16; }
int attn_bwd_f_grid(int B, int h, int w);
int launch_attn_bwd_f(int e, const AttnBwdFArgs& a, hipStream_t s);
