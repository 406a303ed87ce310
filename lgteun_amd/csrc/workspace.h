// Workspace carving shared by forward and backward orchestrators.  The caller owns one flat buffer; this
// maps named activation tensors onto it deterministically from (plan, B, train).
#pragma once
#include "common.h"
#include "kernels.h"

struct Carver {
    char* base;
    size_t off;
    float* take(size_t nfloats) {
        size_t o = off;
        off += (nfloats * sizeof(float) + 255) & ~(size_t)255;
        return base ? reinterpret_cast<float*>(base + o) : nullptr;
    }
};

#define LG_MAX_K 16

struct BlockBufs {
    int e, h, w;       // channels / spatial size of this block
    float *xin, *xmid, *xout;
    float *g, *o2;     // planar [B,e/2,h,w]
    float *amp, *pha;  // saved spectrum [B,e/2,h,w/2+1] (train)
    float* sgn;        // saved sign of the irfft2 output [B,e/2,h,w] (train)
    float *att_o, *att_l;  // train: the local mixer's attention output [B,h,w,e/2] and score-row log-sum-exp [B,h,w,2] (k_attn_m -> k_attn_bwd_f)
    float *a1, *g1, *h2, *a3, *g3;  // [B,h,w,4e]: gelu(h1), gelu'(h1), h2, gelu(h3), gelu'(h3)  (a*, g* train only)
};

struct NetBufs {
    float* Z[LG_MAX_K + 1];                          // Z_0 .. Z_K  [B,C,H,W]: Z[i] -> data step i -> Z[i+1] (= input of LGT i)
    float* X[LG_MAX_K];                              // chained mode: input of data step i (X[0] = Z[0], X[i] = output of LGT i-1)
    float *t1[LG_MAX_K], *r[LG_MAX_K], *s1[LG_MAX_K];  // data-step intermediates per stage
    float* pr;                                       // [B,H,W] scratch of the one-launch data step (k_dstep.hip): R Z - pan of the stage in flight
    float* posT;                                     // [K][5][2*64*64]
    BlockBufs blk[5];
    float* x0;        // embed output = blk[0].xin
    float* deadout;   // output of dead-stage LGTs (faithful mode)
    float* u_down;    // saved bicubic-downsampled encoder output [B,H/2,W/2,E] (train)
    float* t_up;      // saved up-path tensor [B,H,W,E] (train)
    float* fft_scratch;  // PAN > 128 only: half-spectrum scratch of the split FFT path
    float* ffn_scales;   // [K][5][8] operand scales of the f16-pair FFN arithmetic (k_ffn_prep.hip), written once per forward call
    float* attn_scales;  // [K][5][4] static operand scales of the local mixer's f16-pair products { s_y, s_w, s_q, s_k } (k_ffn_prep.hip), written once per forward call
    float* wsplit;       // [K][5] slots of pre-split (and pre-scaled) weight fragments of the e >= 32 FFN blocks (k_ffn_x32.hip), written once per forward call
    size_t set_off, set_bytes;  // the per-LGT activation set [set_off, set_off + set_bytes): train == 2 carves K of them back to back
    size_t bytes;
};

// NetBufs of stage `i` when every stage keeps its own activation set (chained training): same layout, shifted by i sets
static inline NetBufs stage_view(const NetBufs& nb, int i) {
    NetBufs v = nb;
    const size_t sh = (size_t)i * nb.set_bytes;
    auto mv = [sh](float*& p) { if (p) p = reinterpret_cast<float*>(reinterpret_cast<char*>(p) + sh); };
    for (int j = 0; j < 5; ++j) {
        BlockBufs& b = v.blk[j];
        mv(b.xin); mv(b.xmid); mv(b.xout); mv(b.g); mv(b.o2); mv(b.amp); mv(b.pha); mv(b.sgn); mv(b.att_o); mv(b.att_l);
        mv(b.a1); mv(b.g1); mv(b.h2); mv(b.a3); mv(b.g3);
    }
    mv(v.x0); mv(v.u_down); mv(v.t_up);
    return v;
}

// train: 0 = inference, 1 = training (one saved activation set: only the last stage's LGT is live),
// 2 = chained training (LG_FLAG_CHAINED: every stage's LGT is live and keeps its own set)
static inline void carve(const lg_plan* plan, int B, int train, void* base, NetBufs& nb) {
    const lg_config& c = plan->cfg;
    Carver cv{reinterpret_cast<char*>(base), 0};
    const size_t P0 = (size_t)c.H * c.W, P1 = P0 / 4, E = 4 * (size_t)c.C;
    for (int i = 0; i <= c.K; ++i) nb.Z[i] = cv.take(B * c.C * P0);
    for (int i = 0; i < c.K; ++i) {
        nb.t1[i] = cv.take(B * c.C * P1);
        nb.r[i] = cv.take(B * c.C * P1 / 4);
        nb.s1[i] = cv.take(B * c.C * P1);
    }
    nb.pr = cv.take(B * P0);
    nb.posT = cv.take((size_t)c.K * 5 * 2 * 64 * 64);
    nb.ffn_scales = cv.take((size_t)c.K * 5 * 8);
    nb.attn_scales = cv.take((size_t)c.K * 5 * 4);
    nb.wsplit = cv.take((size_t)c.K * 5 * (ffn_wsplit_bytes((int)(2 * E)) / sizeof(float)));   // [K][5] slots sized for the widest block (level 1: e = 8 C): written once per forward call (prep_stages)
    nb.deadout = cv.take(B * c.C * P0);
    nb.X[0] = nb.Z[0];
    for (int i = 1; i < c.K; ++i) nb.X[i] = (train == 2) ? cv.take(B * c.C * P0) : nb.deadout;
    float* shared_h2 = nullptr;
    if (!train) shared_h2 = cv.take(B * P0 * 4 * E);  // largest (level-0) hidden tensor, reused by every block
    nb.set_off = cv.off;
    for (int j = 0; j < 5; ++j) {
        BlockBufs& bb = nb.blk[j];
        const bool l1 = (j == 2);
        bb.e = (int)(l1 ? 2 * E : E);
        bb.h = l1 ? c.H / 2 : c.H;
        bb.w = l1 ? c.W / 2 : c.W;
        const size_t P = l1 ? P1 : P0, e = bb.e;
        bb.xin = nullptr;  // wired below
        bb.xmid = cv.take(B * P * e);
        bb.xout = cv.take(B * P * e);
        bb.g = cv.take(B * P * e / 2);
        bb.o2 = cv.take(B * P * e / 2);
        if (train) {
            bb.amp = cv.take(B * (e / 2) * bb.h * (bb.w / 2 + 1));
            bb.pha = cv.take(B * (e / 2) * bb.h * (bb.w / 2 + 1));
            bb.sgn = cv.take(B * P * e / 2);
            bb.att_o = cv.take(B * P * e / 2);
            bb.att_l = cv.take(B * P * 2);
            bb.a1 = cv.take(B * P * 4 * e);
            bb.g1 = cv.take(B * P * 4 * e);
            bb.h2 = cv.take(B * P * 4 * e);
            bb.a3 = cv.take(B * P * 4 * e);
            bb.g3 = cv.take(B * P * 4 * e);
        } else {
            bb.att_o = bb.att_l = nullptr;
            bb.amp = bb.pha = bb.sgn = bb.a1 = bb.g1 = bb.a3 = bb.g3 = nullptr;
            bb.h2 = shared_h2;
        }
    }
    nb.x0 = cv.take(B * P0 * E);
    nb.blk[0].xin = nb.x0;
    nb.blk[1].xin = nb.blk[0].xout;
    nb.blk[2].xin = cv.take(B * P1 * 2 * E);  // down output
    nb.blk[3].xin = cv.take(B * P0 * E);      // up+fusion output
    nb.blk[4].xin = nb.blk[3].xout;
    nb.u_down = train ? cv.take(B * P1 * E) : nullptr;
    nb.t_up = train ? cv.take(B * P0 * E) : nullptr;
    nb.set_bytes = cv.off - nb.set_off;
    if (train == 2) cv.off += (size_t)(c.K - 1) * nb.set_bytes;
    {
        // level 0: B * E/2 planes of H x W; level 1: B * E planes of H/2 x W/2 -- level 0's is the larger whenever both need one
        const size_t f0 = fft_scratch_floats_hw((int)(B * (E / 2)), c.H, c.W), f1 = fft_scratch_floats_hw((int)(B * E), c.H / 2, c.W / 2);
        const size_t fl = f0 > f1 ? f0 : f1;
        nb.fft_scratch = fl ? cv.take(fl) : nullptr;
    }
    nb.bytes = cv.off;
}
