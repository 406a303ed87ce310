// Internal launcher declarations (one per fused kernel).  All pointers are device pointers.
#pragma once
#include "common.h"

// ---------------- data module (reference models/unlg_former.py:29-37,58-61) ----------------
struct DwArgs {
    const float* in;    // [planes, hi, wi]
    float* out;         // [planes, ho, wo]
    const float* w9;    // [C,1,3,3]
    const float* bias;  // [C]
    const float* sub;   // EPI 1: subtract [planes, ho, wo]
    // EPI 2 (data-step update): out = z - eta * (val + RT(R(z) - pan))
    const float* z;     // [B, C, ho, wo]
    const float* pan;   // [B, 1, ho, wo]
    const float* rw;    // [1,C,1,1]
    const float* rb;    // [1]
    const float* rtw;   // [C,1,1,1]
    const float* rtb;   // [C]
    const float* eta;   // scalar
    int C, planes, hi, wi, ho, wo;
};
// MODE 0: x0.5, 1: x2 ; EPI 0: none, 1: minus sub, 2: data-step update
int launch_resample_dw(int mode, int epi, const DwArgs& a, hipStream_t s);
int launch_resample(int mode, const float* x, float* y, int planes, int hi, int wi, hipStream_t s);
// the whole proximal-gradient step of a stage in ONE launch (k_dstep.hip): planes that fit the LDS of a CU (square, 128 or 64 wide)
struct DstepFwdArgs {
    const float* z;      // Z_i [B,C,N,N]
    const float* ms;     // [B,C,N/4,N/4]
    const float* pan;    // [B,1,N,N]
    float* zout;         // [B,C,N,N]
    float *t1, *r, *s1;  // the chain's intermediates (what the backward reads): [B,C,N/2,N/2], [B,C,N/4,N/4], [B,C,N/2,N/2]
    float* pr;           // scratch [B,N,N]: R Z - pan of the sample (written by the pixelwise launch in front of the plane kernel)
    const float *d1w, *d1b, *d3w, *d3b, *dt1w, *dt1b, *dt3w, *dt3b, *rw, *rb, *rtw, *rtb, *eta;
    int B, C, N;
};
bool dstep_fused_ok(int C, int H, int W);
int launch_dstep_fwd(const DstepFwdArgs& a, hipStream_t s);

// ---------------- LGT pixelwise pieces (reference models/common/LGT.py) ----------------
struct EmbedArgs {
    const float* z;  // [B,C,H,W]
    float* x;        // [B,H,W,E]
    float* g;        // [B,E/2,H,W] LN1(next block)(x)[..., E/2:]  (nullable)
    const float *dww, *dwb, *w, *b, *lng, *lnb, *n1g, *n1b;
    int HW;
    long total;  // B*H*W
};
int launch_embed(int C, const EmbedArgs& a, hipStream_t s);

struct DownArgs {
    const float* x;  // [B,H,W,E]
    float* y;        // [B,H/2,W/2,2E]
    float* g;        // [B,E,H/2,W/2]
    float* u_save;   // optional [B,H/2,W/2,E]: the resampled conv input (for the weight gradient)
    const float *w, *b, *n1g, *n1b;
    int B, H, W;  // input size
};
int launch_down(int E, const DownArgs& a, hipStream_t s);

struct UpFuseArgs {
    const float* xb;    // [B,H/2,W/2,2E]
    const float* skip;  // [B,H,W,E]
    float* y;           // [B,H,W,E]
    float* g;           // [B,E/2,H,W]
    float* t_save;      // optional [B,H,W,E]: up-path tensor after its 1x1 conv (for the weight gradient)
    const float *upw, *upb, *fw, *fb, *n1g, *n1b;
    int B, H, W;  // output size
};
int launch_upfuse(int E, const UpFuseArgs& a, hipStream_t s);

struct TailArgs {
    const float* x;  // [B,H,W,E]
    const float* z;  // [B,C,H,W]
    float* out;      // [B,C,H,W]
    const float *w, *b;
    int HW;
    long total;
};
int launch_tail(int C, const TailArgs& a, hipStream_t s);

// ---------------- global (FFT) mixer, LGT.py:149-180 ----------------
struct FftArgs {
    const float* g;   // [B,ch,n,n] planar, LayerNorm-ed global half
    float* o;         // [B,ch,n,n] planar: abs(irfft2(...))
    float* amp;       // optional save [B,ch,n,n/2+1]
    float* pha;       // optional save
    float* sgn;       // optional save: sign of the irfft2 output [B,ch,n,n]
    float* scratch;   // n > 128 only: half-spectrum scratch [planes][n][n/2+1] complex
    const float *ampw, *ampb, *phaw, *phab;  // [ch]
    int planes, ch, n;   // n: side of a square plane (legacy callers); h, w (when non-zero) override it
    int h, w;
    int full = 0;      // 1: the complex-row in-LDS kernels (A/B variant LG_VAR_FFT_FULL); 0: the real-input kernels (k_fftmix_r / k_fftmix_bwd_r)
};
int launch_fftmix(const FftArgs& a, hipStream_t s);
size_t fft_scratch_floats(int planes, int n);
size_t fft_scratch_floats_hw(int planes, int h, int w);
bool fft_is_generic(int h, int w);   // true: Bluestein three-kernel path (anything but a square power of two 8..512)

// ---------------- local mixer + proj + residual, LGT.py:112-146,183-219,231-248 ----------------
struct AttnArgs {
    const float* x;     // [B,h,w,e]
    const float* o2;    // [B,e/2,h,w] planar global-mixer output
    float* y;           // [B,h,w,e] = x + dropout(proj(cat(attn, o2)))
    const float* posT;  // [2,64,64] transposed pos_emb: posT[h][j][i] = pos[h][i][j]   (k_attn, the vector-pipe kernel)
    const float* pos;   // [2,64,64] pos_emb as stored: pos[h][i][j]                      (k_attn_m, the matrix-pipe kernel)
    const float *ln1g, *ln1b, *qkvw, *qkvb, *projw, *projb;
    int B, h, w;
    int dropout;
    uint64_t seed;
    int bf16;           // k_attn_m: 1 = one round-to-nearest piece per operand (precision = 'bf16'), 0 = fp32-equivalent split arithmetic
    float* save_o = nullptr;         // k_attn_m, e = 16, saving launch (round 6): [P,e/2] attention output before proj (head-major = the local half of cat) and
    float* save_l = nullptr;         // [P,2] log2-domain log-sum-exp of the score rows, for k_attn_bwd_f (which then skips its reduction pass); null: not written
    const float* scales = nullptr;   // k_attn_m: this block's static operand scales { s_y, s_w, s_q, s_k } (k_ffn_prep.hip, round 6): to_qkv and Q K^T on f16 pairs; nullptr: bf16 triples
};
int launch_attn(int e, const AttnArgs& a, hipStream_t s);     // round 2's kernel: lane = token, every product on the vector pipe (LG_VAR_ATTN_FWD_VALU)
int launch_attn_m(int e, const AttnArgs& a, hipStream_t s);   // round 5: every product on the matrix pipe (k_attn_m.hip)
// posT[blk] for nblk blocks: src pointers via offsets into params
int launch_pos_transpose(const float* pos, float* posT, hipStream_t s);
int launch_pos_transpose_n(int n, const float* const* pos, float* const* posT, hipStream_t s);   // n <= 5 * LG_MAX_K tables in one launch

// ---------------- feed_forward, LGT.py:91-109 ----------------
struct Ffn1Args {
    const float* x;  // [P, e]   (P = B*h*w)
    void* a1s;       // optional save [P,4e]: gelu(h1)        (conv input of W2 for its weight gradient)
    void* g1s;       // optional save [P,4e]: gelu'(h1)       (backward never re-evaluates GELU)
    void* h2;        // [P,4e] = W2 gelu(W1 LN(x) + b1) + b2
    int hbf;         // hidden storage: 0 fp32, 1 bf16 (hstore.h)
    int tile16;      // A/B switch (lg_plan::ffn_tile / ffn_xs): 0 = split-arithmetic kernels (default), 1 = f32-MFMA strip kernel, 2 = f32-MFMA per-tile kernel, 3 = software-pipelined split kernel k_ffn_xp (e = 16), 4 = the channel-split k_ffn_xs where the register-chain k_ffn_xr is the default (e = 16)
    const float *ln2g, *ln2b, *w1, *b1, *w2, *b2;
    long P;
    void* wsplit;    // workspace scratch for pre-split weight fragments (ffn_wsplit_bytes; k_ffn_x32.hip), or nullptr
    int wsplit_ready = 0;   // 1: the fragments are in `wsplit` already (prep launch of the forward call: launch_split_w_jobs); 0: the launcher splits in front of its kernel
    const float* scales;   // this block's operand scales { s_x, s_a1, s_a3, s_w1, s_w2, s_w3 } (k_ffn_prep.hip): the f16-pair arithmetic (NP = 2) of the
                           // fused forward kernels; nullptr = the three-piece bf16 arithmetic (NP = 3)
};
// operand scales of the f16-pair FFN arithmetic: one job per block, all in one launch (k_ffn_prep.hip); out[job][8]
struct FfnPrepJob { const float *ln2g, *ln2b, *w1, *b1, *w2, *b2, *dww, *dwb, *w3; int e; const float *ln1g, *ln1b, *qkvw, *qkvb; };   // ln1 / qkv: the local mixer's static scales (attn_out)
#define LG_MAX_FFN_PREP_JOBS 40
struct FfnPrepTable { FfnPrepJob j[LG_MAX_FFN_PREP_JOBS]; };
int launch_ffn_scales(int n, const FfnPrepJob* jobs, float* out, hipStream_t s, float* attn_out = nullptr);   // attn_out[job][4] = { s_y, s_w, s_q, s_k } (nullptr: not computed)
// pre-split weight fragments of the e >= 32 FFN blocks: one job per block, all blocks of a forward call in ONE launch (k_ffn_x32.hip)
struct SplitWJob { const float *w1, *w2, *w3, *scales; void* out; int e, np; };
struct SplitWTable { SplitWJob j[LG_MAX_FFN_PREP_JOBS]; };
int launch_split_w_jobs(int n, const SplitWJob* jobs, hipStream_t s);
int launch_ffn1(int e, const Ffn1Args& a, hipStream_t s);
struct Ffn2Args {
    const void* h2;   // [B,h,w,4e]
    const float* x;   // [B,h,w,e] residual input
    void* a3s;        // optional save [B,h,w,4e]: gelu(h3)
    void* g3s;        // optional save [B,h,w,4e]: gelu'(h3)
    int hbf;          // hidden storage: 0 fp32, 1 bf16
    float* y;         // [B,h,w,e]
    float* g;         // optional [B,e/2,h,w] LN1(next block)(y) global half
    const float *dww, *dwb, *w3, *b3, *n1g, *n1b;
    int B, h, w;
};
int launch_ffn2(int e, const Ffn2Args& a, hipStream_t s);
// fused feed_forward half-block (h2 stays in LDS); returns LG_FFN_NOT_FUSED when e is not covered -> use launch_ffn1 + launch_ffn2
// (a code of its own: 1 is hipErrorInvalidValue)
#define LG_FFN_NOT_FUSED (-1000)
int launch_ffn_fused(int e, const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s);
int launch_ffn_xs(const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s);   // e = 16, fp32 storage (k_ffn_x.hip)
int launch_ffn_xr(const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s);   // e = 16, f16 pairs, fp32 storage: the register chain (k_ffn_xr.hip, round 6)
int launch_ffn_xp(const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s);   // e = 16, software-pipelined halo pass (k_ffn_xp.hip)
int launch_ffn_x32(const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s);  // e = 32, fp32 storage (k_ffn_x32.hip)
size_t ffn_wsplit_bytes(int e);   // bytes of a1.wsplit for hidden width 4e
// pre-split weight fragments: np = 3 three bf16 pieces, np = 1 one RNE bf16 piece, np = 2 f16 pairs of W * scales[3 + which matrix] (k_ffn_prep.hip)
int launch_split_w(const float* w1, const float* w2, const float* w3, void* out, int e, int np, hipStream_t s, const float* scales = nullptr);
int launch_ffn_x64(const Ffn1Args& a1, const Ffn2Args& a2, hipStream_t s);  // e = 64: two kernels (h2 through HBM), split-bf16 GEMMs (k_ffn_x64.hip)

// test helper: g[B,e/2,HW] = LayerNorm(x)[..., e/2:] (the epilogue the producing kernels fuse)
int launch_ln_split(int e, const float* x, const float* n1g, const float* n1b, float* g, int B, int HW, hipStream_t s);
