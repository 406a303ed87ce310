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
};
size_t wgrad_slab_floats(int N, int K, long P);
int launch_wgrad(const WgradArgs& a, float* slab, hipStream_t s);
int launch_reduce_slab(const float* slab, long nslices, int rows, int cols, float* dst, int ld, int rows_valid, int cols_valid,
                       hipStream_t s);

// ---------------- data module backward ----------------
struct DwBwdArgs {
    const float* gout;  // [planes, n, n] gradient wrt the dw-conv output
    const float* in;    // conv input before resampling: [planes, hi, wi]; conv input = resample<MODE>(in)
    float* gin;         // [planes, n, n] gradient wrt the resampled conv input (dw^T gout)
    const float* w9;    // [C,1,3,3]
    float* dw9;         // grads (+= atomics)
    float* dbias;
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
    int C, B, H, W;
};
int launch_dstep_top_bwd(const DstepTopArgs& a, hipStream_t s);
// adjoint of the bicubic resampler: gin[planes, hi, wi] (= or +=) R^T gout[planes, ho, wo]; mode 0: x0.5, 1: x2
int launch_resample_adj(int mode, const float* gout, float* gin, int planes, int hi, int wi, int accumulate, hipStream_t s);

// ---------------- LGT pixelwise backward ----------------
struct TailBwdArgs {
    const float* dout;  // [B,C,H,W]
    float* dx;          // [P,E] = Wt^T dout
    float* doutp;       // [P,16] zero-padded pixel-major copy of dout (wgrad operand)
    float* dz;          // [B,C,H,W] = dout (residual path)
    const float* w;
    int HW;
    long total;
};
int launch_tail_bwd(int C, const TailBwdArgs& a, hipStream_t s);

struct EmbedBwdArgs {
    const float* dx;  // [P,E] grad wrt embed output
    const float* z;   // [B,C,H,W]
    float* dz;        // [B,C,H,W]  += dt * dww
    float* de;        // [P,E] grad wrt the 1x1 conv output (wgrad operand)
    float* tp;        // [P,16] padded conv input (wgrad operand)
    const float *dww, *dwb, *w, *b, *lng;
    float *d_dww, *d_dwb, *d_lng, *d_lnb;
    int HW;
    long total;
};
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
    long P;
    int hbf;           // hidden storage: 0 fp32, 1 bf16
};
int launch_ffn1_bwd(int e, const Ffn1BwdArgs& a, hipStream_t s);
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
    int planes, ch, n;
};
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
    int B, h, w;
};
int attn_bwd_grid(int e, int B, int h, int w);
int launch_attn_bwd(int e, const AttnBwdArgs& a, hipStream_t s);
