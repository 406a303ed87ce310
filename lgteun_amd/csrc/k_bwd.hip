// Backward orchestration: autograd of Pansharpening.forward (reference models/unlg_former.py:50-67) over the live
// graph only -- the last stage's LGT and the K shared data steps (dead-stage LGTs never get a gradient).
#include <string.h>

#include "kernels.h"
#include "bwd_kernels.h"
#include "backward.h"

struct BwdBufs {
    float *dzA, *dzB, *gu3, *gs1, *gu1, *gr, *gd3, *gt1, *gd1;
    float *dx[3];
    float *dh2, *dh1, *y2, *do2, *dg, *dym, *cat, *y1, *dqkv, *dpos_slab;
    float *w3t[5], *w2t[5], *w1t[5], *wsp, *dt, *dskip, *v, *du, *fft_scratch;   // transposed FFN weights, one set per block of the LGT
    float* slab_arena;    // scratch of the deferred parameter-gradient reductions (ReduceQueue, bwd_kernels.h)
    float* attn_stats;    // [P0][2][4]: row statistics between the two launches of k_attn_bwd_core_m (LG_ATTN_BWD_CORE=m)
    float* ffn_scales;    // NetBufs::ffn_scales of the forward this backward belongs to ([stage][5][8]); null: no f16-pair products in the backward
    size_t slab_cap;      // floats
    ReduceQueue rq;
    size_t bytes;
};

static void carve_bwd(const lg_plan* plan, int B, void* base, BwdBufs& bb) {
    const lg_config& c = plan->cfg;
    Carver cv{reinterpret_cast<char*>(base), 0};
    const size_t P0 = (size_t)c.H * c.W * B, P1 = P0 / 4, E = 4 * (size_t)c.C, C = c.C;
    bb.dzA = cv.take(C * P0); bb.dzB = cv.take(C * P0); bb.gu3 = cv.take(C * P0);
    bb.gs1 = cv.take(C * P1); bb.gu1 = cv.take(C * P1); bb.gt1 = cv.take(C * P1); bb.gd1 = cv.take(C * P1);
    bb.gr = cv.take(C * P1 / 4); bb.gd3 = cv.take(C * P1 / 4);
    for (int i = 0; i < 3; ++i) bb.dx[i] = cv.take(P0 * E);
    bb.dh2 = cv.take(P0 * 4 * E); bb.dh1 = cv.take(P0 * 4 * E);
    bb.y2 = cv.take(P0 * E); bb.do2 = cv.take(P0 * E / 2); bb.dg = cv.take(P0 * E / 2);
    bb.dym = cv.take(P0 * E); bb.cat = cv.take(P0 * E); bb.y1 = cv.take(P0 * 16 > P0 * E / 2 ? P0 * 16 : P0 * E / 2);
    bb.dqkv = cv.take(P0 * 2 * E);
    bb.attn_stats = cv.take(P0 * 8);
    bb.dpos_slab = cv.take((size_t)ATTN_BWD_F_WGS * ATTN_BWD_F_ROW > (size_t)512 * 2 * 64 * 64 ? (size_t)ATTN_BWD_F_WGS * ATTN_BWD_F_ROW : (size_t)512 * 2 * 64 * 64);
    for (int j = 0; j < 5; ++j) { bb.w3t[j] = cv.take(8 * E * 2 * E); bb.w2t[j] = cv.take(8 * E * 8 * E); bb.w1t[j] = cv.take(8 * E * 2 * E); }
    bb.wsp = cv.take(ffn_wsplit_bytes(32) / sizeof(float));   // pre-split W2^T / W1^T fragments of k_ffn1_bwd_x32 (e = 32 blocks)
    size_t sl = wgrad_slab_floats((int)(8 * E), (int)(8 * E), (long)P0);
    size_t sl2 = wgrad_slab_floats(64, 64, (long)P0);
    size_t sl3 = ffn_dw_bwd_slab_floats((int)E, B, c.H, c.W);
    if (sl2 > sl) sl = sl2;
    // per-workgroup partial-sum rows of the parameter-gradient reductions (bwd_kernels.h)
    size_t sl4 = chan_partial_floats(c.C, B, c.H, c.W);
    const size_t sl5 = (size_t)PIXEL_PART_WGS * (2 * 8 * c.C + 2 * c.C);
    if (sl5 > sl4) sl4 = sl5;
    if (sl4 > sl) sl = sl4;
    // arena: one block's worth of slabs (dw-conv partials + the three FFN weight gradients + the small ones) between flushes;
    // take() flushes by itself if a configuration needs more
    if (sl3 > sl) sl = sl3;
    for (size_t e = E; e <= 2 * E; e *= 2)   // slab rows of the fused FFN backward kernels (k_ffn_bwd_x.hip, k_ffn_dwbwd_x.hip): only of the paths this plan runs
        if ((e == 16 && plan->ffn_bwd_x(16)) || (e == 32 && plan->ffn1_bwd_x32(32))) {
            if (ffn1_bwd_x_slab_floats((int)e) > sl) sl = ffn1_bwd_x_slab_floats((int)e);
        }
    if (ffn_dw_bwd_x_slab_floats(16) > sl) sl = ffn_dw_bwd_x_slab_floats(16);
    if (ffn_dw_bwd_x_slab_floats(32) > sl) sl = ffn_dw_bwd_x_slab_floats(32);
    if (ffn_dw_bwd_h_slab_floats() > sl) sl = ffn_dw_bwd_h_slab_floats();
    bb.slab_cap = 4 * sl;
    bb.slab_arena = cv.take(bb.slab_cap);
    bb.ffn_scales = nullptr;
    bb.dt = cv.take(P0 * E); bb.dskip = cv.take(P0 * E); bb.v = cv.take(P1 * E); bb.du = cv.take(P1 * E);
    bb.bytes = cv.off;
}

size_t bwd_workspace_bytes(const lg_plan* plan, int B) {
    BwdBufs bb;
    carve_bwd(plan, B, nullptr, bb);
    return bb.bytes;
}

#define RC(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

// deactivates the calling thread's reduce queue on every exit path (error returns included)
struct ReduceQueueScope {
    explicit ReduceQueueScope(BwdBufs& bb, hipStream_t s) {
        bb.rq.init(bb.slab_arena, bb.slab_cap, s);
        reduce_queue_begin(&bb.rq);
    }
    ~ReduceQueueScope() { (void)reduce_queue_end(); }
};

static int wgrad(const void* Y, int ldy, const void* X, int ldx, float* dW, int ldw, float* db, long P, int N, int K, int nv, int kv,
                 int ybf, int xbf, BwdBufs& bb, hipStream_t s, int xgelu = 0) {
    float* slab = bb.rq.take(wgrad_slab_floats(N, K, P));
    if (!slab) return -3;
    WgradArgs a;
    a.Y = Y; a.X = X; a.dW = dW; a.db = db; a.P = P; a.ldy = ldy; a.ldx = ldx; a.ldw = ldw; a.N = N; a.K = K;
    a.n_valid = nv; a.k_valid = kv; a.ybf = ybf; a.xbf = xbf; a.xgelu = xgelu;
    return launch_wgrad(a, slab, s);
}

// W3^T / W2^T / W1^T of blocks [j0, j1) of stage st into bb.w?t[j], all in ONE launch (backward GEMMs read transposed weights)
static int ffn_transposes(const lg_plan* pl, const float* P, int st, int j0, int j1, const NetBufs& nb, BwdBufs& bb, hipStream_t s) {
    const float* tsrc[15];
    float* tdst[15];
    int trows[15], tcols[15], n = 0;
    for (int j = j0; j < j1; ++j) {
        const int e = nb.blk[j].e, n1 = 4 * e;
        tsrc[n] = P + pl->blk(st, j, B_W3); tdst[n] = bb.w3t[j]; trows[n] = e; tcols[n] = n1; ++n;
        tsrc[n] = P + pl->blk(st, j, B_W2); tdst[n] = bb.w2t[j]; trows[n] = n1; tcols[n] = n1; ++n;
        tsrc[n] = P + pl->blk(st, j, B_W1); tdst[n] = bb.w1t[j]; trows[n] = n1; tcols[n] = e; ++n;
    }
    return launch_transpose3(tsrc, tdst, trows, tcols, n, s);
}

// feed_forward half-block backward: dy (grad wrt block output) -> tmp (grad wrt the mid activation); bb.w?t[j] hold the transposed weights
static int ffn_half_bwd(const lg_plan* pl, const float* P, float* G, int st, int j, const BlockBufs& fb, BwdBufs& bb, const float* dy,
                        float* tmp, int B, hipStream_t s) {
    const int e = fb.e, n1 = 4 * e;
    const int hbf = pl->hidden_bf16(e) ? 1 : 0;       // bf16 storage of the hidden / saved FFN tensors
    const int pre = pl->ffn_saves_preact(e) ? 1 : 0;  // fb.a1 / fb.a3 hold h1 / h3 (fb.g1 / fb.g3 unused): GELU re-evaluated where needed
    const long Pn = (long)B * fb.h * fb.w;
    if (pl->ffn_bwd_x(e) && (!pl->dwbwd_tile || hbf)) {
        // two launches per half-block: the strip-walking spatial half (dh3 in LDS -> dh2, depthwise gradients, dW3 / db3) and the pixelwise
        // half (h1 re-computed, dx, LayerNorm gradients, dW1 / db1, dW2 / db2); dh2 is the only tensor between them
        FfnDwBwdXArgs fk;
        const bool h3re = pl->ffn_h3_recompute(e) && (fb.h & 7) == 0 && (fb.w & 15) == 0;   // round 6: the forward saved h2 only; h3 is re-computed in the kernel
        fk.dy = dy; fk.h3 = h3re ? nullptr : fb.a3; fk.h2 = fb.h2; fk.dh2 = bb.dh2; fk.w3t = bb.w3t[j]; fk.dww = P + pl->blk(st, j, B_DWW); fk.dwb = P + pl->blk(st, j, B_DWB);
        fk.slab = bb.rq.take(h3re ? ffn_dw_bwd_h_slab_floats() : ffn_dw_bwd_x_slab_floats(e));
        if (!fk.slab) return -3;
        fk.d_dww = G + pl->blk(st, j, B_DWW); fk.d_dwb = G + pl->blk(st, j, B_DWB); fk.d_w3 = G + pl->blk(st, j, B_W3); fk.d_b3 = G + pl->blk(st, j, B_B3);
        fk.B = B; fk.h = fb.h; fk.w = fb.w; fk.hbf = hbf;
        // f16-pair products in the pixelwise half (round 5): the forward's operand scales of this block + max |dh2|, which the spatial half leaves in word 6
        float* fsc = (bb.ffn_scales && pl->ffn_f16x2(e) && !pl->ffn_bwd_bf16x3 && !hbf) ? bb.ffn_scales + ((size_t)st * 5 + j) * 8 : nullptr;
        fk.dh2_max = fsc ? fsc + 6 : nullptr;
        RC(h3re ? launch_ffn_dw_bwd_h(fk, s) : launch_ffn_dw_bwd_xs(e, fk, s));
        Ffn1BwdXArgs fx;
        fx.scales = fsc;
        fx.dh2 = bb.dh2; fx.x = fb.xmid; fx.dy = dy; fx.dx = tmp;
        fx.w1 = P + pl->blk(st, j, B_W1); fx.b1 = P + pl->blk(st, j, B_B1); fx.w2t = bb.w2t[j]; fx.w1t = bb.w1t[j];
        fx.ln2g = P + pl->blk(st, j, B_LN2G); fx.ln2b = P + pl->blk(st, j, B_LN2B);
        fx.slab = bb.rq.take(ffn1_bwd_x_slab_floats(e));
        if (!fx.slab) return -3;
        fx.d_w1 = G + pl->blk(st, j, B_W1); fx.d_b1 = G + pl->blk(st, j, B_B1); fx.d_w2 = G + pl->blk(st, j, B_W2); fx.d_b2 = G + pl->blk(st, j, B_B2);
        fx.d_ln2g = G + pl->blk(st, j, B_LN2G); fx.d_ln2b = G + pl->blk(st, j, B_LN2B);
        fx.P = Pn; fx.hbf = hbf;
        return launch_ffn1_bwd_xs(e, fx, s);
    }
    const bool dwx32 = pl->ffn_dw_x32(e, fb.h, fb.w);
    float* fsc2 = nullptr;
    if (dwx32) {
        // e = 32: the strip-walking spatial half (dh3 in an LDS ring -> dh2; depthwise gradients, dW3 / db3 in the same pass) on the saved
        // pre-activation h3; the pixelwise half below is round 2's k_ffn1_bwd_x32 + the 128 x 128 weight-gradient launch
        FfnDwBwdXArgs fk;
        fk.dy = dy; fk.h3 = fb.a3; fk.h2 = fb.h2; fk.dh2 = bb.dh2; fk.w3t = bb.w3t[j]; fk.dww = P + pl->blk(st, j, B_DWW);
        fk.slab = bb.rq.take(ffn_dw_bwd_x_slab_floats(e));
        if (!fk.slab) return -3;
        fk.d_dww = G + pl->blk(st, j, B_DWW); fk.d_dwb = G + pl->blk(st, j, B_DWB); fk.d_w3 = G + pl->blk(st, j, B_W3); fk.d_b3 = G + pl->blk(st, j, B_B3);
        fk.B = B; fk.h = fb.h; fk.w = fb.w; fk.hbf = hbf;
        fsc2 = (bb.ffn_scales && pl->ffn_f16x2(e) && !pl->ffn_bwd_bf16x3 && !hbf && pl->ffn1_bwd_x32(e)) ? bb.ffn_scales + ((size_t)st * 5 + j) * 8 : nullptr;
        fk.dh2_max = fsc2 ? fsc2 + 6 : nullptr;
        RC(launch_ffn_dw_bwd_xs(e, fk, s));
    }
    if (!dwx32) {
        FfnDwBwdArgs fd;
        fd.dy = dy; fd.g3 = pre ? fb.a3 : fb.g3; fd.h2 = fb.h2; fd.dh2 = bb.dh2; fd.w3t = bb.w3t[j]; fd.dww = P + pl->blk(st, j, B_DWW);
        fd.slab_w = bb.rq.take(ffn_dw_bwd_slab_floats(e, B, fb.h, fb.w));
        if (!fd.slab_w) return -3;
        fd.slab_b = fd.slab_w + ffn_dw_bwd_slab_floats(e, B, fb.h, fb.w) / 10 * 9;
        fd.d_dww = G + pl->blk(st, j, B_DWW); fd.d_dwb = G + pl->blk(st, j, B_DWB);
        fd.B = B; fd.h = fb.h; fd.w = fb.w; fd.hbf = hbf; fd.pre = pre;
        RC(launch_ffn_dw_bwd(e, fd, s));
    }
    if (pl->ffn_bwd_x(e) || pl->ffn1_bwd_x32(e)) {
        // one pass over dh2 re-computes h1 and yields dx, the LayerNorm gradients, dW1 / db1 and dW2 / db2 (e = 16: h1 was not saved; e = 32:
        // the saved gelu(h1) / gelu'(h1) are simply not read)
        Ffn1BwdXArgs fx;
        fx.scales = fsc2;      // (set only when dh2 came from k_ffn_dw_bwd_xs, which leaves max |dh2| behind)
        fx.dh2 = bb.dh2; fx.x = fb.xmid; fx.dy = dy; fx.dx = tmp;
        fx.w1 = P + pl->blk(st, j, B_W1); fx.b1 = P + pl->blk(st, j, B_B1); fx.w2t = bb.w2t[j]; fx.w1t = bb.w1t[j];
        fx.ln2g = P + pl->blk(st, j, B_LN2G); fx.ln2b = P + pl->blk(st, j, B_LN2B);
        fx.slab = bb.rq.take(ffn1_bwd_x_slab_floats(e));
        if (!fx.slab) return -3;
        fx.d_w1 = G + pl->blk(st, j, B_W1); fx.d_b1 = G + pl->blk(st, j, B_B1); fx.d_w2 = G + pl->blk(st, j, B_W2); fx.d_b2 = G + pl->blk(st, j, B_B2);
        fx.d_ln2g = G + pl->blk(st, j, B_LN2G); fx.d_ln2b = G + pl->blk(st, j, B_LN2B);
        fx.P = Pn; fx.hbf = hbf;
        RC(launch_ffn1_bwd_xs(e, fx, s));
        if (dwx32) return 0;   // dW3 / db3 came out of k_ffn_dw_bwd_xs<32>
        return wgrad(dy, e, fb.a3, n1, G + pl->blk(st, j, B_W3), n1, G + pl->blk(st, j, B_B3), Pn, e, n1, e, n1, 0, hbf, bb, s, pre);
    }
    Ffn1BwdArgs f1;
    f1.dh2 = bb.dh2; f1.g1 = pre ? fb.a1 : fb.g1; f1.x = fb.xmid; f1.dy = dy; f1.dh1 = bb.dh1; f1.y2 = bb.y2; f1.dx = tmp;
    f1.w2t = bb.w2t[j]; f1.w1t = bb.w1t[j];
    f1.ln2g = P + pl->blk(st, j, B_LN2G); f1.ln2b = P + pl->blk(st, j, B_LN2B);
    f1.d_ln2g = G + pl->blk(st, j, B_LN2G); f1.d_ln2b = G + pl->blk(st, j, B_LN2B); f1.part = bb.rq.take((size_t)PIXEL_PART_WGS * 2 * e);
    if (!f1.part) return -3;
    f1.P = Pn; f1.hbf = hbf; f1.pre = pre;
    f1.w1 = nullptr; f1.wsplit = nullptr;
    if (e == 32 && pl->ffn_tile == 0) { f1.w1 = P + pl->blk(st, j, B_W1); f1.wsplit = bb.wsp; }   // LG_FFN_IMPL=strip|tile: the f32-MFMA kernel (A/B)
    f1.w1slab = nullptr; f1.d_w1 = nullptr; f1.d_b1 = nullptr;
    if (ffn1_bwd_fuses_w1(e)) {   // dW1 / db1 come out of k_ffn1_bwd itself
        f1.w1slab = bb.rq.take((size_t)FFN1_BWD_WGS * ((size_t)n1 * e + n1));
        if (!f1.w1slab) return -3;
        f1.d_w1 = G + pl->blk(st, j, B_W1); f1.d_b1 = G + pl->blk(st, j, B_B1);
    }
    f1.w2slab = nullptr; f1.d_w2 = nullptr; f1.d_b2 = nullptr;
    if (ffn1_bwd_fuses_w2(e, pre)) {   // dW2 / db2 too
        f1.w2slab = bb.rq.take((size_t)FFN1_BWD_WGS * ((size_t)n1 * n1 + n1));
        if (!f1.w2slab) return -3;
        f1.d_w2 = G + pl->blk(st, j, B_W2); f1.d_b2 = G + pl->blk(st, j, B_B2);
    }
    RC(launch_ffn1_bwd(e, f1, s));
    if (!ffn1_bwd_fuses_w2(e, pre))
        RC(wgrad(bb.dh2, n1, fb.a1, n1, G + pl->blk(st, j, B_W2), n1, G + pl->blk(st, j, B_B2), Pn, n1, n1, n1, n1, hbf, hbf, bb, s, pre));
    if (!ffn1_bwd_fuses_w1(e))
        RC(wgrad(bb.dh1, n1, bb.y2, e, G + pl->blk(st, j, B_W1), e, G + pl->blk(st, j, B_B1), Pn, n1, e, n1, e, hbf, 0, bb, s));
    // last: dh2's two readers run right behind its producer (Infinity Cache), this one only needs the saved gelu(h3) and dy
    if (!dwx32) RC(wgrad(dy, e, fb.a3, n1, G + pl->blk(st, j, B_W3), n1, G + pl->blk(st, j, B_B3), Pn, e, n1, e, n1, 0, hbf, bb, s, pre));
    return 0;
}

static int fft_bwd_call(const lg_plan* pl, const float* P, float* G, int st, int j, const BlockBufs& fb, const float* do2, float* dg,
                        int B, hipStream_t s, float* fft_scratch, float* part) {
    const int hc = fb.e / 2;
    FftBwdArgs fa;
    fa.do2 = do2; fa.sgn = fb.sgn; fa.amp = fb.amp; fa.pha = fb.pha; fa.dg = dg; fa.scratch = fft_scratch;
    fa.ampw = P + pl->blk(st, j, B_AMPW); fa.ampb = P + pl->blk(st, j, B_AMPB);
    fa.phaw = P + pl->blk(st, j, B_PHAW); fa.phab = P + pl->blk(st, j, B_PHAB);
    fa.d_ampw = G + pl->blk(st, j, B_AMPW); fa.d_ampb = G + pl->blk(st, j, B_AMPB);
    fa.d_phaw = G + pl->blk(st, j, B_PHAW); fa.d_phab = G + pl->blk(st, j, B_PHAB);
    fa.ch = hc; fa.planes = B * hc; fa.n = fb.h; fa.h = fb.h; fa.w = fb.w; fa.part = part; fa.full = pl->fft_full;
    return launch_fftmix_bwd(fa, s);
}

// mixer half-block backward: tmp (grad wrt the mid activation) -> dx_out (grad wrt block input)
static int mixer_half_bwd(const lg_plan* pl, const float* P, float* G, int st, int j, const BlockBufs& fb, BwdBufs& bb, const float* posT,
                          const float* tmp, float* dx_out, int B, int flags, uint64_t seed, hipStream_t s) {
    const int e = fb.e, hc = e / 2;
    const long Pn = (long)B * fb.h * fb.w;
    const int drop = (flags & LG_FLAG_DROPOUT) ? 1 : 0;
    if (attn_bwd_fused(e) && !pl->attn_bwd_old) {
        // round 4: three launches per half-block -- proj^T towards the global mixer (+ dropout keep bits, proj bias gradient), the FFT-mixer
        // backward, and ONE kernel for everything else (flash passes, to_qkv^T, LayerNorm backward, dx, every parameter gradient)
        uint32_t* keep = drop ? reinterpret_cast<uint32_t*>(bb.dym) : nullptr;
        ProjO2BwdKArgs pk;
        pk.dy = tmp; pk.do2 = bb.do2; pk.keep = keep; pk.projw = P + pl->blk(st, j, B_PROJW);
        pk.slab = bb.rq.take((size_t)PROJ_O2_K_WGS * e);
        if (!pk.slab) return -3;
        pk.d_projb = G + pl->blk(st, j, B_PROJB); pk.HW = fb.h * fb.w; pk.total = Pn; pk.seed = mix_seed(seed, st, j);
        RC(launch_proj_o2_bwd_k(e, pk, s));
        float* fpart = bb.rq.take(fft_bwd_part_floats(B * hc, fb.h, fb.w));
        if (!fpart) return -3;
        RC(fft_bwd_call(pl, P, G, st, j, fb, bb.do2, bb.dg, B, s, bb.fft_scratch, fpart));
        AttnBwdFArgs af;
        af.x = fb.xin; af.dy = tmp; af.keep = keep; af.o2 = fb.o2; af.dg = bb.dg; af.dx = dx_out;
        af.pos = P + pl->blk(st, j, B_POS);
        af.ln1g = P + pl->blk(st, j, B_LN1G); af.ln1b = P + pl->blk(st, j, B_LN1B);
        af.qkvw = P + pl->blk(st, j, B_QKVW); af.qkvb = P + pl->blk(st, j, B_QKVB); af.projw = P + pl->blk(st, j, B_PROJW);
        af.slab = bb.dpos_slab;   // ATTN_BWD_F_WGS rows: untouched until the block's flush
        af.d_pos = G + pl->blk(st, j, B_POS); af.d_qkvw = G + pl->blk(st, j, B_QKVW); af.d_qkvb = G + pl->blk(st, j, B_QKVB);
        af.d_projw = G + pl->blk(st, j, B_PROJW); af.d_ln1g = G + pl->blk(st, j, B_LN1G); af.d_ln1b = G + pl->blk(st, j, B_LN1B);
        af.B = B; af.h = fb.h; af.w = fb.w;
        if (pl->attn_saves_stats(e)) { af.so = fb.att_o; af.sl = fb.att_l; }   // left by the forward's saving launch (block_mixer_fwd)
        return launch_attn_bwd_f(e, af, s);
    }
    ProjO2BwdArgs po;
    po.dy = tmp; po.do2 = bb.do2; po.dym = drop ? bb.dym : nullptr; po.projw = P + pl->blk(st, j, B_PROJW);
    po.HW = fb.h * fb.w; po.total = Pn; po.dropout = drop; po.seed = mix_seed(seed, st, j);
    RC(launch_proj_o2_bwd(e, po, s));
    const float* dym = drop ? bb.dym : tmp;
    float* fpart = bb.rq.take(fft_bwd_part_floats(B * hc, fb.h, fb.w));
    if (!fpart) return -3;
    RC(fft_bwd_call(pl, P, G, st, j, fb, bb.do2, bb.dg, B, s, bb.fft_scratch, fpart));
    AttnBwdArgs at;
    at.x = fb.xin; at.dy = tmp; at.dym = dym; at.o2 = fb.o2; at.dg = bb.dg; at.dx = dx_out;
    at.cat = bb.cat; at.y1 = bb.y1; at.dqkv = bb.dqkv;
    at.pos = P + pl->blk(st, j, B_POS); at.posT = posT; at.dpos_slab = bb.dpos_slab;
    at.ln1g = P + pl->blk(st, j, B_LN1G); at.ln1b = P + pl->blk(st, j, B_LN1B);
    at.qkvw = P + pl->blk(st, j, B_QKVW); at.qkvb = P + pl->blk(st, j, B_QKVB); at.projw = P + pl->blk(st, j, B_PROJW);
    at.d_ln1g = G + pl->blk(st, j, B_LN1G); at.d_ln1b = G + pl->blk(st, j, B_LN1B); at.part = bb.rq.take(attn_bwd_part_floats(e));
    at.d_qkvw = G + pl->blk(st, j, B_QKVW); at.d_qkvb = G + pl->blk(st, j, B_QKVB);
    if (attn_bwd_fuses_qkv(e)) at.y1 = nullptr;   // the epilogue kernel forms y1 itself and accumulates the to_qkv weight gradient
    if (!at.part) return -3;
    at.B = B; at.h = fb.h; at.w = fb.w; at.core_m = pl->attn_bwd_core_m; at.stats = bb.attn_stats;
    if (pl->attn_saves_stats(e) && !(e == 32 && pl->attn_bwd_core_m)) { at.so = fb.att_o; at.sl = fb.att_l; }   // left by the forward's saving launch (block_mixer_fwd)
    RC(launch_attn_bwd(e, at, s));
    const int grid = attn_bwd_grid(e, B, fb.h, fb.w);
    RC(launch_reduce_slab(bb.dpos_slab, grid, 1, 2 * 64 * 64, G + pl->blk(st, j, B_POS), 2 * 64 * 64, 1, 2 * 64 * 64, s));
    RC(wgrad(dym, e, bb.cat, e, G + pl->blk(st, j, B_PROJW), e, G + pl->blk(st, j, B_PROJB), Pn, e, e, e, e, 0, 0, bb, s));
    if (!attn_bwd_fuses_qkv(e)) {
        const int y1ld = (hc + 15) / 16 * 16, dqld = (3 * hc + 15) / 16 * 16;
        RC(wgrad(bb.dqkv, dqld, bb.y1, y1ld, G + pl->blk(st, j, B_QKVW), hc, G + pl->blk(st, j, B_QKVB), Pn, dqld, y1ld, 3 * hc, hc, 0, 0,
                 bb, s));
    }
    return 0;
}

static int block_bwd(const lg_plan* pl, const float* P, float* G, int st, int j, const BlockBufs& fb, BwdBufs& bb, const float* posT,
                     const float* dy, float* tmp, float* dx_out, int B, int flags, uint64_t seed, hipStream_t s) {
    RC(ffn_half_bwd(pl, P, G, st, j, fb, bb, dy, tmp, B, s));
    RC(mixer_half_bwd(pl, P, G, st, j, fb, bb, posT, tmp, dx_out, B, flags, seed, s));
    return bb.rq.flush();   // dpos_slab and the arena are reused by the next block
}

// per-op backward entry (tests): which 0: global mixer (dy, dx planar), 1: mixer half-block, 2: ffn half-block
int op_block_bwd(const lg_plan* pl, const float* P, float* G, int st, int j, int which, const float* dy, float* dx, NetBufs& nb,
                 void* bwd_ws, int B, hipStream_t s) {
    BwdBufs bb;
    carve_bwd(pl, B, bwd_ws, bb);
    bb.fft_scratch = nb.fft_scratch;
    bb.ffn_scales = nb.ffn_scales;
    const BlockBufs& fb = nb.blk[j];
    if (which == 0) return fft_bwd_call(pl, P, G, st, j, fb, dy, dx, B, s, nb.fft_scratch, bb.slab_arena);   // no queue: summed at once
    ReduceQueueScope rqs(bb, s);
    int rc = which == 1 ? 0 : ffn_transposes(pl, P, st, j, j + 1, nb, bb, s);
    if (!rc) rc = which == 1 ? mixer_half_bwd(pl, P, G, st, j, fb, bb, nb.posT, dy, dx, B, 0, 0, s)
                             : ffn_half_bwd(pl, P, G, st, j, fb, bb, dy, dx, B, s);
    const int rc2 = reduce_queue_end();
    return rc ? rc : rc2;
}

// zin: the tensor the forward data step of stage st consumed (nb.Z[st]; nb.X[st] in chained mode)
static int data_step_bwd(const lg_plan* pl, const float* P, float* G, int st, const NetBufs& nb, BwdBufs& bb, const float* zin,
                         const float* pan, const float* g, float* dz, int B, hipStream_t s) {
    const lg_config& c = pl->cfg;
    const int planes = B * c.C, H = c.H, W = c.W;
    if (pl->dstep_fused(H, W)) {
        // one pixelwise + one plane-in-LDS launch (k_dstep.hip) instead of the nine tile launches below
        DstepBwdArgs a;
        a.g = g; a.z = zin; a.pan = pan; a.t1 = nb.t1[st]; a.r = nb.r[st]; a.s1 = nb.s1[st]; a.dz = dz;
        a.d1w = P + pl->shared(S_D1W); a.d3w = P + pl->shared(S_D3W); a.dt1w = P + pl->shared(S_DT1W); a.dt3w = P + pl->shared(S_DT3W);
        a.dt3b = P + pl->shared(S_DT3B);
        a.rw = P + pl->shared(S_RW); a.rb = P + pl->shared(S_RB); a.rtw = P + pl->shared(S_RTW); a.rtb = P + pl->shared(S_RTB);
        a.eta = P + pl->eta(st);
        a.B = B; a.C = c.C; a.N = H;
        float* part = bb.rq.take(dstep_bwd_part_floats(c.C, B, H));
        if (!part) return -3;
        const size_t bc = (size_t)B * c.C;
        a.part_top = part; a.part_dt1 = part + 14 * bc; a.part_d3 = part + 24 * bc; a.part_d1 = part + 34 * bc; a.part_pre = part + 44 * bc;
        DstepBwdGrads gg;
        gg.d1w = G + pl->shared(S_D1W); gg.d1b = G + pl->shared(S_D1B); gg.d3w = G + pl->shared(S_D3W); gg.d3b = G + pl->shared(S_D3B);
        gg.dt1w = G + pl->shared(S_DT1W); gg.dt1b = G + pl->shared(S_DT1B); gg.dt3w = G + pl->shared(S_DT3W); gg.dt3b = G + pl->shared(S_DT3B);
        gg.rw = G + pl->shared(S_RW); gg.rb = G + pl->shared(S_RB); gg.rtw = G + pl->shared(S_RTW); gg.rtb = G + pl->shared(S_RTB);
        gg.eta = G + pl->eta(st);
        RC(launch_dstep_bwd(a, gg, s));
        return bb.rq.flush();   // the K stages share these parameters: two stages' jobs must not meet in one reduce launch
    }
    DstepTopArgs t;
    t.g = g; t.s1 = nb.s1[st]; t.z = zin; t.pan = pan; t.gu = bb.gu3; t.dz = dz;
    t.w9 = P + pl->shared(S_DT3W); t.b9 = P + pl->shared(S_DT3B);
    t.rw = P + pl->shared(S_RW); t.rb = P + pl->shared(S_RB); t.rtw = P + pl->shared(S_RTW); t.rtb = P + pl->shared(S_RTB);
    t.eta = P + pl->eta(st);
    t.dw9 = G + pl->shared(S_DT3W); t.dbias = G + pl->shared(S_DT3B);
    t.drw = G + pl->shared(S_RW); t.drb = G + pl->shared(S_RB); t.drtw = G + pl->shared(S_RTW); t.drtb = G + pl->shared(S_RTB);
    t.deta = G + pl->eta(st);
    t.C = c.C; t.B = B; t.H = H; t.W = W; t.part = bb.rq.take(chan_partial_floats(c.C, B, H, W));
    if (!t.part) return -3;
    RC(launch_dstep_top_bwd(t, s));
    // s1 = dw(up(r)): grad wrt s1, then through the DT.1 conv
    RC(launch_resample_adj(1, bb.gu3, bb.gs1, planes, H / 2, W / 2, 0, s));
    DwBwdArgs d;
    d.C = c.C; d.planes = planes;
    d.gout = bb.gs1; d.in = nb.r[st]; d.gin = bb.gu1; d.w9 = P + pl->shared(S_DT1W);
    d.dw9 = G + pl->shared(S_DT1W); d.dbias = G + pl->shared(S_DT1B);
    d.hi = H / 4; d.wi = W / 4; d.n_h = H / 2; d.n_w = W / 2;
    d.part = bb.rq.take(chan_partial_floats(c.C, B, H, W));
    if (!d.part) return -3;
    RC(launch_dw_bwd(1, d, s));
    RC(launch_resample_adj(1, bb.gu1, bb.gr, planes, H / 4, W / 4, 0, s));
    // r = dw(down(t1)) - ms
    d.gout = bb.gr; d.in = nb.t1[st]; d.gin = bb.gd3; d.w9 = P + pl->shared(S_D3W);
    d.dw9 = G + pl->shared(S_D3W); d.dbias = G + pl->shared(S_D3B);
    d.hi = H / 2; d.wi = W / 2; d.n_h = H / 4; d.n_w = W / 4;
    d.part = bb.rq.take(chan_partial_floats(c.C, B, H, W));
    if (!d.part) return -3;
    RC(launch_dw_bwd(0, d, s));
    RC(launch_resample_adj(0, bb.gd3, bb.gt1, planes, H / 2, W / 2, 0, s));
    // t1 = dw(down(Z))
    d.gout = bb.gt1; d.in = zin; d.gin = bb.gd1; d.w9 = P + pl->shared(S_D1W);
    d.dw9 = G + pl->shared(S_D1W); d.dbias = G + pl->shared(S_D1B);
    d.hi = H; d.wi = W; d.n_h = H / 2; d.n_w = W / 2;
    d.part = bb.rq.take(chan_partial_floats(c.C, B, H, W));
    if (!d.part) return -3;
    RC(launch_dw_bwd(0, d, s));
    RC(launch_resample_adj(0, bb.gd1, dz, planes, H, W, 1, s));
    return bb.rq.flush();   // the K stages share these parameters: two stages' jobs must not meet in one reduce launch
}

// backward of stage st's LGT (LGT.py:314-344, reversed) from the activation set `nb` holds: dout = gradient wrt the LGT's
// output, zin = the LGT's input; leaves the gradient wrt zin in bb.dzA
static int lgt_bwd(const lg_plan* pl, const float* P, float* G, int st, const NetBufs& nb, BwdBufs& bb, const float* dout,
                   const float* zin, int B, int flags, uint64_t seed, hipStream_t s) {
    const lg_config& c = pl->cfg;
    const int E = 4 * c.C;
    const long P0 = (long)B * c.H * c.W, P1 = P0 / 4;
    const float* posT = nb.posT + (size_t)st * 5 * 2 * 64 * 64;
    float *A = bb.dx[0], *Bf = bb.dx[1], *Cf = bb.dx[2];
    RC(ffn_transposes(pl, P, st, 0, 5, nb, bb, s));
    TailBwdArgs tb;
    tb.dout = dout; tb.x = nb.blk[4].xout; tb.dx = A; tb.dz = bb.dzA; tb.w = P + pl->lgt(st, L_TAILW);
    tb.d_w = G + pl->lgt(st, L_TAILW); tb.d_b = G + pl->lgt(st, L_TAILB);   // the conv's own weight gradient comes out of the same kernel
    tb.HW = c.H * c.W; tb.total = P0; tb.part = bb.rq.take(tail_bwd_part_floats(c.C));
    if (!tb.part) return -3;
    RC(launch_tail_bwd(c.C, tb, s));
    RC(block_bwd(pl, P, G, st, 4, nb.blk[4], bb, posT + 4 * 8192, A, Bf, Cf, B, flags, seed, s));
    RC(block_bwd(pl, P, G, st, 3, nb.blk[3], bb, posT + 3 * 8192, Cf, Bf, A, B, flags, seed, s));
    // up + fusion
    UpFuseBwdArgs ub;
    ub.dy = A; ub.dt = bb.dt; ub.dskip = bb.dskip; ub.v = bb.v; ub.dxb = Bf; ub.tmp = Cf;   // Cf is free until the bottleneck block
    ub.fw = P + pl->lgt(st, L_FUSEW); ub.upw = P + pl->lgt(st, L_UPW);
    ub.B = B; ub.H = c.H; ub.W = c.W;
    RC(launch_upfuse_bwd_a(E, ub, s));
    RC(wgrad(A, E, nb.t_up, E, G + pl->lgt(st, L_FUSEW), 2 * E, G + pl->lgt(st, L_FUSEB), P0, E, E, E, E, 0, 0, bb, s));
    RC(wgrad(A, E, nb.blk[1].xout, E, G + pl->lgt(st, L_FUSEW) + E, 2 * E, nullptr, P0, E, E, E, E, 0, 0, bb, s));
    RC(launch_upfuse_bwd_b(E, ub, s));
    RC(wgrad(bb.v, E, nb.blk[2].xout, 2 * E, G + pl->lgt(st, L_UPW), 2 * E, G + pl->lgt(st, L_UPB), P1, E, 2 * E, E, 2 * E, 0, 0, bb, s));
    // bottleneck
    RC(block_bwd(pl, P, G, st, 2, nb.blk[2], bb, posT + 2 * 8192, Bf, Cf, A, B, flags, seed, s));
    // down
    DownBwdArgs db;
    db.dy = A; db.du = bb.du; db.dskip = bb.dskip; db.dx = Bf; db.w = P + pl->lgt(st, L_DOWNW);
    db.B = B; db.H = c.H; db.W = c.W;
    RC(launch_down_bwd_a(E, db, s));
    RC(wgrad(A, 2 * E, nb.u_down, E, G + pl->lgt(st, L_DOWNW), E, G + pl->lgt(st, L_DOWNB), P1, 2 * E, E, 2 * E, E, 0, 0, bb, s));
    RC(launch_down_bwd_b(E, db, s));
    // encoder
    RC(block_bwd(pl, P, G, st, 1, nb.blk[1], bb, posT + 1 * 8192, Bf, Cf, A, B, flags, seed, s));
    RC(block_bwd(pl, P, G, st, 0, nb.blk[0], bb, posT + 0 * 8192, A, Bf, Cf, B, flags, seed, s));
    // patch embed
    EmbedBwdArgs eb;
    eb.dx = Cf; eb.z = zin; eb.dz = bb.dzA;
    eb.dww = P + pl->lgt(st, L_PE_DWW); eb.dwb = P + pl->lgt(st, L_PE_DWB); eb.w = P + pl->lgt(st, L_PE_W); eb.b = P + pl->lgt(st, L_PE_B);
    eb.lng = P + pl->lgt(st, L_PE_LNG);
    eb.d_dww = G + pl->lgt(st, L_PE_DWW); eb.d_dwb = G + pl->lgt(st, L_PE_DWB);
    eb.d_lng = G + pl->lgt(st, L_PE_LNG); eb.d_lnb = G + pl->lgt(st, L_PE_LNB);
    eb.d_w = G + pl->lgt(st, L_PE_W); eb.d_b = G + pl->lgt(st, L_PE_B);   // the conv's own weight gradient comes out of the same kernel
    eb.HW = c.H * c.W; eb.total = P0; eb.part = bb.rq.take(embed_bwd_part_floats(c.C));
    if (!eb.part) return -3;
    RC(launch_embed_bwd(c.C, eb, s));
    return 0;
}

// per-op backward entries (tests): one data step / one LGT in isolation.  The forward of the same piece has just filled `nb`
// (data step: nb.t1 / r / s1 [st]; LGT: the saved activation set).
int op_data_step_bwd(const lg_plan* pl, const float* P, float* G, int st, NetBufs& nb, void* bwd_ws, const float* z_in, const float* pan,
                     const float* g, float* dz, int B, hipStream_t s) {
    BwdBufs bb;
    carve_bwd(pl, B, bwd_ws, bb);
    ReduceQueueScope rqs(bb, s);
    const int rc = data_step_bwd(pl, P, G, st, nb, bb, z_in, pan, g, dz, B, s);
    const int rc2 = reduce_queue_end();
    return rc ? rc : rc2;
}

int op_lgt_bwd(const lg_plan* pl, const float* P, float* G, int st, NetBufs& nb, void* bwd_ws, const float* z, const float* dout, float* dz,
               int B, int flags, uint64_t seed, hipStream_t s) {
    BwdBufs bb;
    carve_bwd(pl, B, bwd_ws, bb);
    bb.fft_scratch = nb.fft_scratch;
    bb.ffn_scales = nb.ffn_scales;
    bb.dzA = dz;                       // lgt_bwd leaves the gradient wrt the LGT's input here
    ReduceQueueScope rqs(bb, s);
    const int rc = lgt_bwd(pl, P, G, st, nb, bb, dout, z, B, flags, seed, s);
    const int rc2 = reduce_queue_end();
    return rc ? rc : rc2;
}

int net_backward(const lg_plan* pl, const float* P, float* G, const float* ms, const float* pan, const float* dout, NetBufs& nb,
                 void* bwd_ws, int B, int flags, uint64_t seed, hipStream_t s) {
    (void)ms;
    const lg_config& c = pl->cfg;
    BwdBufs bb;
    carve_bwd(pl, B, bwd_ws, bb);
    bb.fft_scratch = nb.fft_scratch;
    bb.ffn_scales = nb.ffn_scales;
    ReduceQueueScope rqs(bb, s);
    if (flags & LG_FLAG_CHAINED) {
        // intended unfolding (every stage live): LGT_i then data step i, last stage first; each LGT reads its own activation set
        const float* g = dout;
        for (int i = c.K - 1; i >= 0; --i) {
            const NetBufs sv = stage_view(nb, i);
            RC(lgt_bwd(pl, P, G, i, sv, bb, g, nb.Z[i + 1], B, flags, seed, s));
            RC(data_step_bwd(pl, P, G, i, nb, bb, nb.X[i], pan, bb.dzA, bb.dzB, B, s));
            g = bb.dzB;
        }
        return reduce_queue_end();
    }
    const bool do_lgt = !(flags & (LG_FLAG_BWD_LGT | LG_FLAG_BWD_DATA)) || (flags & LG_FLAG_BWD_LGT);
    const bool do_data = !(flags & (LG_FLAG_BWD_LGT | LG_FLAG_BWD_DATA)) || (flags & LG_FLAG_BWD_DATA);
    // ---------------- LGT of the last stage: the only live one
    if (do_lgt) RC(lgt_bwd(pl, P, G, c.K - 1, nb, bb, dout, nb.Z[c.K], B, flags, seed, s));
    if (!do_data) return reduce_queue_end();
    // ---------------- K shared data steps, last to first (unlg_former.py:56-61); input gradient: bb.dzA
    float* g = bb.dzA;
    float* dz = bb.dzB;
    for (int i = c.K - 1; i >= 0; --i) {
        RC(data_step_bwd(pl, P, G, i, nb, bb, nb.Z[i], pan, g, dz, B, s));
        float* t = g; g = dz; dz = t;
    }
    return reduce_queue_end();
}
