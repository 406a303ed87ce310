// C ABI entry points (include/lgteun_hip.h): plan, forward orchestration, per-op entries, L1 loss, Adam.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <atomic>
#include <mutex>

// only the C ABI of include/lgteun_hip.h is exported from the shared object (everything else: -fvisibility=hidden)
#pragma GCC visibility push(default)
#include "../../include/lgteun_hip.h"
#pragma GCC visibility pop

#include "kernels.h"
#include "workspace.h"
#include "backward.h"

static thread_local char g_err[512] = "";

void lg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------------------
// live per-kernel timing
// ------------------------------------------------------------------------------------------------
// The ONE piece of mutable process-global state of the library (documented in the header): the event table of the timing
// facility.  `kid` is atomic so that the launch path pays one relaxed load while profiling is off; everything else is under the mutex.
static struct {
    std::atomic<int> kid{0};
    std::atomic<int> paused{0};
    int cap = 0, n = 0;
    hipEvent_t* ev = nullptr;  // 2 per launch
    std::mutex mu;
} g_prof;

void lg_prof_begin(int kid, hipStream_t s) {
    if (kid != g_prof.kid.load(std::memory_order_relaxed) || g_prof.paused.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (kid != g_prof.kid.load(std::memory_order_relaxed) || g_prof.n >= g_prof.cap) return;
    hipEventRecord(g_prof.ev[2 * g_prof.n], s);
}
void lg_prof_end(int kid, hipStream_t s) {
    if (kid != g_prof.kid.load(std::memory_order_relaxed) || g_prof.paused.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (kid != g_prof.kid.load(std::memory_order_relaxed) || g_prof.n >= g_prof.cap) return;
    hipEventRecord(g_prof.ev[2 * g_prof.n + 1], s);
    g_prof.n++;
}
static void prof_disable_locked() {
    for (int i = 0; i < 2 * g_prof.cap; ++i) hipEventDestroy(g_prof.ev[i]);
    free(g_prof.ev);
    g_prof.ev = nullptr;
    g_prof.cap = g_prof.n = 0;
    g_prof.kid.store(0);
    g_prof.paused.store(0);
}
extern "C" void lg_prof_disable(void) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    prof_disable_locked();
}
extern "C" int lg_prof_enable(int32_t kernel_id, int32_t max_launches) {
    if (kernel_id <= 0 || kernel_id >= LG_K_COUNT || max_launches <= 0) { lg_set_error("prof_enable: invalid argument"); return -1; }
    std::lock_guard<std::mutex> lk(g_prof.mu);
    prof_disable_locked();
    g_prof.ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * 2 * max_launches);
    for (int i = 0; i < 2 * max_launches; ++i) {
        hipError_t e = hipEventCreate(&g_prof.ev[i]);
        if (e != hipSuccess) { lg_set_error("prof_enable: hipEventCreate: %s", hipGetErrorString(e)); return (int)e; }
    }
    g_prof.cap = max_launches;
    g_prof.kid.store(kernel_id);
    return 0;
}
// sampling: while paused, launches are neither timed nor counted (an event pair costs ~2 us of stream time; a caller that times whole
// steps around the kernels keeps that out of most of them).  Toggle between launches of the profiled kernel only.
extern "C" void lg_prof_pause(int32_t paused) { g_prof.paused.store(paused ? 1 : 0); }
extern "C" int lg_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.n = 0;
    return 0;
}
extern "C" int lg_prof_read(double* total_ms, int64_t* launches) {
    if (!total_ms || !launches) { lg_set_error("prof_read: null argument"); return -1; }
    std::lock_guard<std::mutex> lk(g_prof.mu);
    double tot = 0.0;
    for (int i = 0; i < g_prof.n; ++i) {
        hipError_t e = hipEventSynchronize(g_prof.ev[2 * i + 1]);
        if (e != hipSuccess) { lg_set_error("prof_read: %s", hipGetErrorString(e)); return (int)e; }
        float ms = 0.f;
        hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
        tot += ms;
    }
    *total_ms = tot;
    *launches = g_prof.n;
    return 0;
}
extern "C" const char* lg_kernel_name(int32_t k) {
    static const char* names[LG_K_COUNT] = {"none", "k_ffn1", "fused FFN forward (k_ffn_xr e=16 / k_ffn_x32 e=32 / k_ffn1_x64+k_ffn2_x64 e=64)", "k_fftmix", "k_attn", "k_upfuse", "k_down", "k_embed", "k_tail",
                                            "k_resample_dw", "k_ffn1_bwd", "k_ffn2_bwd", "k_fftmix_bwd", "k_attn_bwd", "k_wgrad"};
    return (k >= 0 && k < LG_K_COUNT) ? names[k] : "?";
}

extern "C" const char* lg_version(void) { return "lgteun_hip 0.2 (gfx950)"; }
extern "C" int32_t lg_abi_version(void) { return LG_ABI_VERSION; }
extern "C" const char* lg_last_error(void) { return g_err; }

extern "C" int lg_plan_create(const lg_config* cfg, const int64_t* offsets, int32_t n_offsets, lg_plan** out) {
    if (!cfg || !offsets || !out) { lg_set_error("plan_create: null argument"); return -1; }
    if (cfg->C != 4 && cfg->C != 8) { lg_set_error("plan_create: C must be 4 or 8 (got %d)", cfg->C); return -2; }
    if (cfg->precision != 0 && cfg->precision != 1) { lg_set_error("plan_create: precision must be 0 (fp32) or 1 (bf16 hidden storage)"); return -2; }
    if (cfg->K < 1 || cfg->K > LG_MAX_K) { lg_set_error("plan_create: K out of range (%d)", cfg->K); return -2; }
    if (cfg->H % 16 || cfg->W % 16 || cfg->H <= 0 || cfg->W <= 0) { lg_set_error("plan_create: H,W must be positive multiples of 16"); return -2; }
    if (cfg->H > 1024 || cfg->W > 1024) {   // FFT mixer: Bluestein lines of up to 1024 points (square powers of two <= 512: radix-2 paths)
        lg_set_error("plan_create: PAN sizes up to 1024x1024 are supported (got %dx%d)", cfg->H, cfg->W);
        return -2;
    }
    const int expect = S_NSHARED + cfg->K + L_NSLOT * cfg->K;
    if (n_offsets != expect) { lg_set_error("plan_create: expected %d offsets, got %d", expect, n_offsets); return -2; }
    for (int i = 0; i < n_offsets; ++i)
        if (offsets[i] < 0 || (offsets[i] & 3)) { lg_set_error("plan_create: offset %d (=%lld) must be a non-negative multiple of 4 floats", i, (long long)offsets[i]); return -2; }
    if (cfg->variant & ~LG_VAR_ALL) { lg_set_error("plan_create: unknown variant bits 0x%x", cfg->variant & ~LG_VAR_ALL); return -2; }
    if ((cfg->variant & LG_VAR_FFN_SAVE_MASK) == LG_VAR_FFN_SAVE_MASK) { lg_set_error("plan_create: invalid FFN save variant"); return -2; }
#ifndef LG_BUILD_AB
    if ((cfg->variant & LG_VAR_FFN_IMPL_MASK) >= LG_VAR_FFN_TILE) { lg_set_error("plan_create: FFN variants 2 / 3 exist in `make AB=1` builds only"); return -2; }
    if (cfg->precision == 1 && (cfg->variant & LG_VAR_FFN_IMPL_MASK)) { lg_set_error("plan_create: precision = 1 with an FFN variant exists in `make AB=1` builds only"); return -2; }
#endif
    lg_plan* p = new lg_plan;
    p->cfg = *cfg;
    p->n_offsets = n_offsets;
    {   // A/B switches come in through lg_config.variant (the library reads no environment variable)
        const uint32_t v = cfg->variant;
        p->ffn_tile = (int)(v & LG_VAR_FFN_IMPL_MASK);
        const uint32_t sv = v & LG_VAR_FFN_SAVE_MASK;
        p->save_mode = sv == LG_VAR_FFN_SAVE5 ? 5 : (sv == LG_VAR_FFN_SAVE3 ? 3 : 2);   // common.h: lg_plan::save_mode
        p->bwd32_old = (v & LG_VAR_FFN_BWD32_PAIR) ? 1 : 0;   // default (round 5): k_ffn1_bwd_xs<32> behind the strip-walking spatial half -- 14.36 -> 14.20 ms per c3 step, and the forward no longer saves gelu(h1) / gelu'(h1)
        p->dwbwd_tile = (v & LG_VAR_FFN_DWBWD_TILE) ? 1 : 0;
        p->attn_bwd_old = (v & LG_VAR_ATTN_BWD_R3) ? 1 : 0;
        p->dstep_tiles = (v & LG_VAR_DSTEP_TILES) ? 1 : 0;
        p->attn_fwd_valu = (v & LG_VAR_ATTN_FWD_VALU) ? 1 : 0;
        p->ffn_bf16x3 = (v & LG_VAR_FFN_BF16X3) ? 1 : 0;
        p->fft_full = (v & LG_VAR_FFT_FULL) ? 1 : 0;
        p->attn_bwd_core_m = (v & LG_VAR_ATTN_BWD_CORE_M) ? 1 : 0;
        p->ffn_bwd_bf16x3 = (v & LG_VAR_FFN_BWD_BF16X3) ? 1 : 0;
        p->ffn_xs = (v & LG_VAR_FFN_XS) ? 1 : 0;
        p->attn_bf16x3 = (v & LG_VAR_ATTN_BF16X3) ? 1 : 0;
        p->ffn_h3_re = (v & LG_VAR_FFN_H3_RECOMPUTE) ? 1 : 0;
        p->attn_restats = (v & LG_VAR_ATTN_BWD_RESTATS) ? 1 : 0;
    }
    p->off = (int64_t*)malloc(sizeof(int64_t) * n_offsets);
    memcpy(p->off, offsets, sizeof(int64_t) * n_offsets);
    *out = p;
    return 0;
}

extern "C" void lg_plan_destroy(lg_plan* plan) {
    if (!plan) return;
    free(plan->off);
    delete plan;
}

extern "C" size_t lg_workspace_bytes(const lg_plan* plan, int32_t B, int32_t train) {
    if (!plan || B <= 0 || train < 0 || train > 2) return 0;
    NetBufs nb;
    carve(plan, B, train, nullptr, nb);
    size_t fwd = nb.bytes;
    if (train) fwd += bwd_workspace_bytes(plan, B);
    return fwd;
}

// ------------------------------------------------------------------------------------------------
// forward pieces
// ------------------------------------------------------------------------------------------------
static int data_step_fwd(const lg_plan* pl, const float* P, int stage, const float* z_in, const float* ms, const float* pan,
                         float* z_out, float* t1, float* r, float* s1, float* pr, int B, hipStream_t s) {
    const lg_config& c = pl->cfg;
    const int planes = B * c.C, H = c.H, W = c.W;
    if (pl->dstep_fused(H, W)) {
        DstepFwdArgs f;
        f.z = z_in; f.ms = ms; f.pan = pan; f.zout = z_out; f.t1 = t1; f.r = r; f.s1 = s1; f.pr = pr;
        f.d1w = P + pl->shared(S_D1W); f.d1b = P + pl->shared(S_D1B); f.d3w = P + pl->shared(S_D3W); f.d3b = P + pl->shared(S_D3B);
        f.dt1w = P + pl->shared(S_DT1W); f.dt1b = P + pl->shared(S_DT1B); f.dt3w = P + pl->shared(S_DT3W); f.dt3b = P + pl->shared(S_DT3B);
        f.rw = P + pl->shared(S_RW); f.rb = P + pl->shared(S_RB); f.rtw = P + pl->shared(S_RTW); f.rtb = P + pl->shared(S_RTB);
        f.eta = P + pl->eta(stage);
        f.B = B; f.C = c.C; f.N = H;
        return launch_dstep_fwd(f, s);
    }
    DwArgs a;
    memset(&a, 0, sizeof(a));
    a.C = c.C; a.planes = planes;
    int rc;
    // D: x0.5, dw3, x0.5, dw3  (unlg_former.py:29-30) ; then "- ms" (unlg_former.py:58)
    a.in = z_in; a.out = t1; a.w9 = P + pl->shared(S_D1W); a.bias = P + pl->shared(S_D1B);
    a.hi = H; a.wi = W; a.ho = H / 2; a.wo = W / 2;
    if ((rc = launch_resample_dw(0, 0, a, s))) return rc;
    a.in = t1; a.out = r; a.w9 = P + pl->shared(S_D3W); a.bias = P + pl->shared(S_D3B); a.sub = ms;
    a.hi = H / 2; a.wi = W / 2; a.ho = H / 4; a.wo = W / 4;
    if ((rc = launch_resample_dw(0, 1, a, s))) return rc;
    // DT: x2, dw3, x2, dw3 (unlg_former.py:32-33)
    a.in = r; a.out = s1; a.w9 = P + pl->shared(S_DT1W); a.bias = P + pl->shared(S_DT1B); a.sub = nullptr;
    a.hi = H / 4; a.wi = W / 4; a.ho = H / 2; a.wo = W / 2;
    if ((rc = launch_resample_dw(1, 0, a, s))) return rc;
    // last DT stage fused with pan term and the update (unlg_former.py:59-61)
    a.in = s1; a.out = z_out; a.w9 = P + pl->shared(S_DT3W); a.bias = P + pl->shared(S_DT3B);
    a.z = z_in; a.pan = pan; a.rw = P + pl->shared(S_RW); a.rb = P + pl->shared(S_RB);
    a.rtw = P + pl->shared(S_RTW); a.rtb = P + pl->shared(S_RTB); a.eta = P + pl->eta(stage);
    a.hi = H / 2; a.wi = W / 2; a.ho = H; a.wo = W;
    return launch_resample_dw(1, 2, a, s);
}

static int block_mixer_fwd(const lg_plan* pl, const float* P, int stage, int j, const BlockBufs& bb, const float* posT, int B,
                           int flags, uint64_t seed, hipStream_t s, float* fft_scratch = nullptr, const float* attn_scales = nullptr) {
    int rc;
    FftArgs f;
    f.g = bb.g; f.o = bb.o2;
    f.amp = (flags & LG_FLAG_SAVE) ? bb.amp : nullptr;
    f.pha = (flags & LG_FLAG_SAVE) ? bb.pha : nullptr;
    f.sgn = (flags & LG_FLAG_SAVE) ? bb.sgn : nullptr;
    f.scratch = fft_scratch;
    f.ampw = P + pl->blk(stage, j, B_AMPW); f.ampb = P + pl->blk(stage, j, B_AMPB);
    f.phaw = P + pl->blk(stage, j, B_PHAW); f.phab = P + pl->blk(stage, j, B_PHAB);
    f.ch = bb.e / 2; f.planes = B * f.ch; f.n = bb.h; f.h = bb.h; f.w = bb.w; f.full = pl->fft_full;
    if ((rc = launch_fftmix(f, s))) return rc;
    AttnArgs t;
    t.x = bb.xin; t.o2 = bb.o2; t.y = bb.xmid; t.posT = posT; t.pos = P + pl->blk(stage, j, B_POS);
    t.bf16 = pl->cfg.precision == 1 ? 1 : 0;
    t.ln1g = P + pl->blk(stage, j, B_LN1G); t.ln1b = P + pl->blk(stage, j, B_LN1B);
    t.qkvw = P + pl->blk(stage, j, B_QKVW); t.qkvb = P + pl->blk(stage, j, B_QKVB);
    t.projw = P + pl->blk(stage, j, B_PROJW); t.projb = P + pl->blk(stage, j, B_PROJB);
    t.B = B; t.h = bb.h; t.w = bb.w;
    t.dropout = (flags & LG_FLAG_DROPOUT) ? 1 : 0;
    t.seed = mix_seed(seed, stage, j);
    t.scales = (attn_scales && pl->attn_f16x2()) ? attn_scales + ((size_t)stage * 5 + j) * 4 : nullptr;   // written by prep_stages for the stages of this call
    if ((flags & LG_FLAG_SAVE) && pl->attn_saves_stats(bb.e)) { t.save_o = bb.att_o; t.save_l = bb.att_l; }
    return pl->attn_fwd_valu ? launch_attn(bb.e, t, s) : launch_attn_m(bb.e, t, s);
}

static int block_ffn_fwd(const lg_plan* pl, const float* P, int stage, int j, const BlockBufs& bb, float* g_next, int next_blk,
                         int B, int flags, hipStream_t s, float* wsplit, const float* ffn_scales) {
    int rc;
    Ffn1Args a1;
    const bool pre = pl->ffn_saves_preact(bb.e);   // h1 / h3 go to the a1 / a3 slots, nothing to g1 / g3
    const bool noh1 = pl->ffn_bwd_x(bb.e) || pl->ffn1_bwd_x32(bb.e);   // the backward re-computes h1 from x: nothing of it is saved
    a1.x = bb.xmid; a1.a1s = ((flags & LG_FLAG_SAVE) && !noh1) ? bb.a1 : nullptr; a1.g1s = ((flags & LG_FLAG_SAVE) && !pre && !noh1) ? bb.g1 : nullptr; a1.h2 = bb.h2;
    a1.ln2g = P + pl->blk(stage, j, B_LN2G); a1.ln2b = P + pl->blk(stage, j, B_LN2B);
    a1.w1 = P + pl->blk(stage, j, B_W1); a1.b1 = P + pl->blk(stage, j, B_B1);
    a1.w2 = P + pl->blk(stage, j, B_W2); a1.b2 = P + pl->blk(stage, j, B_B2);
    a1.P = (long)B * bb.h * bb.w;
    a1.hbf = pl->hidden_bf16(bb.e) ? 1 : 0;
    a1.tile16 = pl->ffn_tile ? pl->ffn_tile : ((pl->ffn_xs && bb.e == 16) ? 4 : 0);
    a1.wsplit = wsplit ? wsplit + ((size_t)stage * 5 + j) * (ffn_wsplit_bytes(8 * pl->cfg.C) / sizeof(float)) : nullptr;   // this block's slot, filled by prep_stages
    a1.wsplit_ready = (wsplit && bb.e >= 32) ? 1 : 0;
    a1.scales = pl->ffn_f16x2(bb.e) ? ffn_scales + ((size_t)stage * 5 + j) * 8 : nullptr;   // written by prep_stages for the stages of this call
    Ffn2Args a2;
    a2.h2 = bb.h2; a2.x = bb.xmid; a2.a3s = ((flags & LG_FLAG_SAVE) && !pl->ffn_h3_recompute(bb.e)) ? bb.a3 : nullptr; a2.g3s = ((flags & LG_FLAG_SAVE) && !pre && !pl->ffn_dw_x32(bb.e, bb.h, bb.w)) ? bb.g3 : nullptr; a2.y = bb.xout;   // g3s null with a3s set: a3 receives the PRE-activation h3
    a2.g = g_next;
    a2.dww = P + pl->blk(stage, j, B_DWW); a2.dwb = P + pl->blk(stage, j, B_DWB);
    a2.w3 = P + pl->blk(stage, j, B_W3); a2.b3 = P + pl->blk(stage, j, B_B3);
    a2.n1g = g_next ? P + pl->blk(stage, next_blk, B_LN1G) : nullptr;
    a2.n1b = g_next ? P + pl->blk(stage, next_blk, B_LN1B) : nullptr;
    a2.B = B; a2.h = bb.h; a2.w = bb.w; a2.hbf = a1.hbf;
    // fused path (e <= 32): h2 only leaves the chip when the backward needs it; e = 64 passes it through HBM between its two kernels
    a1.h2 = ((flags & LG_FLAG_SAVE) || bb.e == 64) ? bb.h2 : nullptr;
    rc = launch_ffn_fused(bb.e, a1, a2, s);
    if (rc != LG_FFN_NOT_FUSED) return rc;
    a1.h2 = bb.h2;
    if ((rc = launch_ffn1(bb.e, a1, s))) return rc;
    return launch_ffn2(bb.e, a2, s);
}

// pos_emb^T of stages [st0, st1) into nb.posT, all tables in ONE launch (a launch per stage was 5 us + a launch gap each)
static int pos_transpose_stages(const lg_plan* pl, const float* P, int st0, int st1, float* posT_all, hipStream_t s) {
    if (st1 <= st0) return 0;
    const float* src[5 * LG_MAX_K];
    float* dst[5 * LG_MAX_K];
    int n = 0;
    for (int st = st0; st < st1; ++st)
        for (int j = 0; j < 5; ++j, ++n) { src[n] = P + pl->blk(st, j, B_POS); dst[n] = posT_all + ((size_t)st * 5 + j) * 2 * 64 * 64; }
    return launch_pos_transpose_n(n, src, dst, s);
}
// what the LGTs of stages [st0, st1) need in front of their first kernel: the transposed pos_emb tables (round 2's vector-pipe mixer alone reads
// them) and the operand scales of the f16-pair FFN arithmetic -- one launch each for all stages of the call
static int prep_stages(const lg_plan* pl, const float* P, int st0, int st1, NetBufs& nb, hipStream_t s) {
    int rc = pl->attn_fwd_valu ? pos_transpose_stages(pl, P, st0, st1, nb.posT, s) : 0;   // only round 2's vector-pipe forward (LG_ATTN_FWD=valu) reads the transposed tables
    if (rc || st1 <= st0) return rc;
    const int E = 4 * pl->cfg.C;
    if (pl->ffn_f16x2(E) || pl->ffn_f16x2(2 * E)) {
        FfnPrepJob jobs[5 * LG_MAX_K];
        int n = 0;
        for (int st = st0; st < st1; ++st)
            for (int j = 0; j < 5; ++j, ++n) {
                FfnPrepJob& q = jobs[n];
                q.ln2g = P + pl->blk(st, j, B_LN2G); q.ln2b = P + pl->blk(st, j, B_LN2B);
                q.w1 = P + pl->blk(st, j, B_W1); q.b1 = P + pl->blk(st, j, B_B1); q.w2 = P + pl->blk(st, j, B_W2); q.b2 = P + pl->blk(st, j, B_B2);
                q.dww = P + pl->blk(st, j, B_DWW); q.dwb = P + pl->blk(st, j, B_DWB); q.w3 = P + pl->blk(st, j, B_W3);
                q.e = j == 2 ? 2 * E : E;
                q.ln1g = P + pl->blk(st, j, B_LN1G); q.ln1b = P + pl->blk(st, j, B_LN1B); q.qkvw = P + pl->blk(st, j, B_QKVW); q.qkvb = P + pl->blk(st, j, B_QKVB);
            }
        if ((rc = launch_ffn_scales(n, jobs, nb.ffn_scales + (size_t)st0 * 5 * 8, s, pl->attn_f16x2() ? nb.attn_scales + (size_t)st0 * 5 * 4 : nullptr))) return rc;
    }
    // the pre-split weight fragments of every e >= 32 block of these stages, behind the scales they are multiplied by (round 5: one launch per
    // forward call instead of one in front of every FFN launch)
    SplitWJob sj[5 * LG_MAX_K];
    int ns = 0;
    const size_t slot = ffn_wsplit_bytes(2 * E) / sizeof(float);
    for (int st = st0; st < st1; ++st)
        for (int j = 0; j < 5; ++j) {
            const int e = j == 2 ? 2 * E : E;
            if (e < 32 || e % 32) continue;
            SplitWJob& q = sj[ns++];
            q.w1 = P + pl->blk(st, j, B_W1); q.w2 = P + pl->blk(st, j, B_W2); q.w3 = P + pl->blk(st, j, B_W3);
            q.e = e; q.np = pl->hidden_bf16(e) ? 1 : (pl->ffn_f16x2(e) ? 2 : 3);
            q.scales = q.np == 2 ? nb.ffn_scales + ((size_t)st * 5 + j) * 8 : nullptr;
            q.out = nb.wsplit + ((size_t)st * 5 + j) * slot;
        }
    return ns ? launch_split_w_jobs(ns, sj, s) : 0;
}

// LGT.forward (LGT.py:314-344) on z -> out with the buffers of `nb`
// pos_ready: nb.posT already holds this stage's transposed pos_emb tables (the net-level entries transpose all stages in one launch)
static int lgt_fwd(const lg_plan* pl, const float* P, int stage, const float* z, float* out, NetBufs& nb, int B, int flags,
                   uint64_t seed, hipStream_t s, bool pos_ready = false) {
    const lg_config& c = pl->cfg;
    const int E = 4 * c.C;
    int rc;
    float* posT = nb.posT + (size_t)stage * 5 * 2 * 64 * 64;
    if (!pos_ready && (rc = prep_stages(pl, P, stage, stage + 1, nb, s))) return rc;
    EmbedArgs ea;
    ea.z = z; ea.x = nb.x0; ea.g = nb.blk[0].g;
    ea.dww = P + pl->lgt(stage, L_PE_DWW); ea.dwb = P + pl->lgt(stage, L_PE_DWB);
    ea.w = P + pl->lgt(stage, L_PE_W); ea.b = P + pl->lgt(stage, L_PE_B);
    ea.lng = P + pl->lgt(stage, L_PE_LNG); ea.lnb = P + pl->lgt(stage, L_PE_LNB);
    ea.n1g = P + pl->blk(stage, 0, B_LN1G); ea.n1b = P + pl->blk(stage, 0, B_LN1B);
    ea.HW = c.H * c.W; ea.total = (long)B * c.H * c.W;
    if ((rc = launch_embed(c.C, ea, s))) return rc;
    // encoder LGB (2 blocks)
    if ((rc = block_mixer_fwd(pl, P, stage, 0, nb.blk[0], posT + 0 * 8192, B, flags, seed, s, nb.fft_scratch, nb.attn_scales))) return rc;
    if ((rc = block_ffn_fwd(pl, P, stage, 0, nb.blk[0], nb.blk[1].g, 1, B, flags, s, nb.wsplit, nb.ffn_scales))) return rc;
    if ((rc = block_mixer_fwd(pl, P, stage, 1, nb.blk[1], posT + 1 * 8192, B, flags, seed, s, nb.fft_scratch, nb.attn_scales))) return rc;
    if ((rc = block_ffn_fwd(pl, P, stage, 1, nb.blk[1], nullptr, 0, B, flags, s, nb.wsplit, nb.ffn_scales))) return rc;
    // down
    DownArgs da;
    da.x = nb.blk[1].xout; da.y = nb.blk[2].xin; da.g = nb.blk[2].g;
    da.u_save = (flags & LG_FLAG_SAVE) ? nb.u_down : nullptr;
    da.w = P + pl->lgt(stage, L_DOWNW); da.b = P + pl->lgt(stage, L_DOWNB);
    da.n1g = P + pl->blk(stage, 2, B_LN1G); da.n1b = P + pl->blk(stage, 2, B_LN1B);
    da.B = B; da.H = c.H; da.W = c.W;
    if ((rc = launch_down(E, da, s))) return rc;
    // bottleneck
    if ((rc = block_mixer_fwd(pl, P, stage, 2, nb.blk[2], posT + 2 * 8192, B, flags, seed, s, nb.fft_scratch, nb.attn_scales))) return rc;
    if ((rc = block_ffn_fwd(pl, P, stage, 2, nb.blk[2], nullptr, 0, B, flags, s, nb.wsplit, nb.ffn_scales))) return rc;
    // up + fusion
    UpFuseArgs ua;
    ua.xb = nb.blk[2].xout; ua.skip = nb.blk[1].xout; ua.y = nb.blk[3].xin; ua.g = nb.blk[3].g;
    ua.t_save = (flags & LG_FLAG_SAVE) ? nb.t_up : nullptr;
    ua.upw = P + pl->lgt(stage, L_UPW); ua.upb = P + pl->lgt(stage, L_UPB);
    ua.fw = P + pl->lgt(stage, L_FUSEW); ua.fb = P + pl->lgt(stage, L_FUSEB);
    ua.n1g = P + pl->blk(stage, 3, B_LN1G); ua.n1b = P + pl->blk(stage, 3, B_LN1B);
    ua.B = B; ua.H = c.H; ua.W = c.W;
    if ((rc = launch_upfuse(E, ua, s))) return rc;
    // decoder LGB (2 blocks)
    if ((rc = block_mixer_fwd(pl, P, stage, 3, nb.blk[3], posT + 3 * 8192, B, flags, seed, s, nb.fft_scratch, nb.attn_scales))) return rc;
    if ((rc = block_ffn_fwd(pl, P, stage, 3, nb.blk[3], nb.blk[4].g, 4, B, flags, s, nb.wsplit, nb.ffn_scales))) return rc;
    if ((rc = block_mixer_fwd(pl, P, stage, 4, nb.blk[4], posT + 4 * 8192, B, flags, seed, s, nb.fft_scratch, nb.attn_scales))) return rc;
    if ((rc = block_ffn_fwd(pl, P, stage, 4, nb.blk[4], nullptr, 0, B, flags, s, nb.wsplit, nb.ffn_scales))) return rc;
    // tail
    TailArgs ta;
    ta.x = nb.blk[4].xout; ta.z = z; ta.out = out;
    ta.w = P + pl->lgt(stage, L_TAILW); ta.b = P + pl->lgt(stage, L_TAILB);
    ta.HW = c.H * c.W; ta.total = (long)B * c.H * c.W;
    return launch_tail(c.C, ta, s);
}

extern "C" int lgteun_forward(const lg_plan* plan, const float* params, const float* ms, const float* pan, float* out,
                              void* workspace, size_t workspace_bytes, int32_t B, int32_t flags, uint64_t seed, void* stream) {
    if (!plan || !params || !ms || !pan || !out || !workspace || B <= 0) { lg_set_error("forward: null/invalid argument"); return -1; }
    const bool chained = (flags & LG_FLAG_CHAINED) != 0;
    const int train = (flags & LG_FLAG_SAVE) ? (chained ? 2 : 1) : 0;
    if (workspace_bytes < lg_workspace_bytes(plan, B, train)) {
        lg_set_error("forward: workspace too small (%zu < %zu)", workspace_bytes, lg_workspace_bytes(plan, B, train));
        return -3;
    }
    hipStream_t s = (hipStream_t)stream;
    const lg_config& c = plan->cfg;
    NetBufs nb;
    carve(plan, B, train, workspace, nb);
    int rc;
    // pos_emb^T of every stage whose LGT runs in this call
    const bool all_stages = chained || ((flags & LG_FLAG_FAITHFUL) && !(flags & LG_FLAG_DEFER_DEAD));
    if ((rc = prep_stages(plan, params, all_stages ? 0 : c.K - 1, c.K, nb, s))) return rc;
    // Z0 = bicubic x4 (unlg_former.py:53)
    if ((rc = launch_resample(2, ms, nb.Z[0], B * c.C, c.H / 4, c.W / 4, s))) return rc;
    if (chained) {
        // intended unfolding: X_{i+1} = LGT_i(data_step_i(X_i)); every stage saves into its own activation set when training
        for (int i = 0; i < c.K; ++i) {
            if ((rc = data_step_fwd(plan, params, i, nb.X[i], ms, pan, nb.Z[i + 1], nb.t1[i], nb.r[i], nb.s1[i], nb.pr, B, s))) return rc;
            NetBufs sv = (train == 2) ? stage_view(nb, i) : nb;
            if ((rc = lgt_fwd(plan, params, i, nb.Z[i + 1], i == c.K - 1 ? out : nb.X[i + 1], sv, B, flags, seed, s, true))) return rc;
        }
        return 0;
    }
    for (int i = 0; i < c.K; ++i) {
        if ((rc = data_step_fwd(plan, params, i, nb.Z[i], ms, pan, nb.Z[i + 1], nb.t1[i], nb.r[i], nb.s1[i], nb.pr, B, s))) return rc;
        const bool last = (i == c.K - 1);
        if (last) {
            if ((rc = lgt_fwd(plan, params, i, nb.Z[i + 1], out, nb, B, flags, seed, s, true))) return rc;
        } else if ((flags & LG_FLAG_FAITHFUL) && !(flags & LG_FLAG_DEFER_DEAD)) {
            // the reference executes these LGTs and discards their result (unlg_former.py:63-67)
            if ((rc = lgt_fwd(plan, params, i, nb.Z[i + 1], nb.deadout, nb, B, flags & ~LG_FLAG_SAVE, seed, s, true))) return rc;
        }
    }
    return 0;
}

extern "C" int lgteun_dead_forward(const lg_plan* plan, const float* params, void* workspace, size_t workspace_bytes, int32_t B,
                                   int32_t flags, uint64_t seed, void* stream) {
    if (!plan || !params || !workspace || B <= 0) { lg_set_error("dead_forward: null/invalid argument"); return -1; }
    if ((flags & LG_FLAG_CHAINED) || !(flags & LG_FLAG_FAITHFUL)) { lg_set_error("dead_forward: only the faithful unfolding has dead stages"); return -2; }
    const int train = (flags & LG_FLAG_SAVE) ? 1 : 0;
    if (workspace_bytes < lg_workspace_bytes(plan, B, train)) { lg_set_error("dead_forward: workspace too small"); return -3; }
    NetBufs nb;
    carve(plan, B, train, workspace, nb);
    {
        const int rc = prep_stages(plan, params, 0, plan->cfg.K - 1, nb, (hipStream_t)stream);
        if (rc) return rc;
    }
    for (int i = 0; i + 1 < plan->cfg.K; ++i) {
        const int rc = lgt_fwd(plan, params, i, nb.Z[i + 1], nb.deadout, nb, B, flags & ~LG_FLAG_SAVE, seed, (hipStream_t)stream, true);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int lgteun_backward(const lg_plan* plan, const float* params, float* grads, const float* ms, const float* pan,
                               const float* dout, void* workspace, size_t workspace_bytes, int32_t B, int32_t flags, uint64_t seed,
                               void* stream) {
    if (!plan || !params || !grads || !ms || !pan || !dout || !workspace || B <= 0) { lg_set_error("backward: null/invalid argument"); return -1; }
    const int train = (flags & LG_FLAG_CHAINED) ? 2 : 1;
    if ((flags & LG_FLAG_CHAINED) && (flags & (LG_FLAG_BWD_LGT | LG_FLAG_BWD_DATA))) {
        lg_set_error("backward: LG_FLAG_BWD_LGT / LG_FLAG_BWD_DATA do not apply to LG_FLAG_CHAINED");
        return -2;
    }
    if (workspace_bytes < lg_workspace_bytes(plan, B, train)) { lg_set_error("backward: workspace too small"); return -3; }
    NetBufs nb;
    carve(plan, B, train, workspace, nb);
    return net_backward(plan, params, grads, ms, pan, dout, nb, (char*)workspace + nb.bytes, B, flags, seed, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// per-op entry points
// ------------------------------------------------------------------------------------------------
extern "C" int lg_op_resample(const float* x, float* y, int32_t planes, int32_t hi, int32_t wi, int32_t mode, void* stream) {
    if (!x || !y || planes <= 0 || hi <= 0 || wi <= 0 || mode < 0 || mode > 2) { lg_set_error("op_resample: invalid argument"); return -1; }
    if (mode == 0 && ((hi & 1) || (wi & 1))) { lg_set_error("op_resample: x0.5 needs even sizes"); return -2; }
    return launch_resample(mode, x, y, planes, hi, wi, (hipStream_t)stream);
}

extern "C" int lg_op_data_step(const lg_plan* plan, const float* params, int32_t stage, const float* z_in, const float* ms,
                               const float* pan, float* z_out, float* tmp, int32_t B, void* stream) {
    if (!plan || !params || !z_in || !ms || !pan || !z_out || !tmp || stage < 0 || stage >= plan->cfg.K) { lg_set_error("op_data_step: invalid argument"); return -1; }
    const lg_config& c = plan->cfg;
    size_t q = (size_t)B * c.C * c.H * c.W / 4;
    return data_step_fwd(plan, params, stage, z_in, ms, pan, z_out, tmp, tmp + q, tmp + 2 * q, tmp + 3 * q, B, (hipStream_t)stream);
}

extern "C" int lg_op_lgt(const lg_plan* plan, const float* params, int32_t stage, const float* z, float* out, void* workspace,
                         size_t workspace_bytes, int32_t B, int32_t flags, uint64_t seed, void* stream) {
    if (!plan || !params || !z || !out || !workspace || stage < 0 || stage >= plan->cfg.K) { lg_set_error("op_lgt: invalid argument"); return -1; }
    const int train = (flags & LG_FLAG_SAVE) ? 1 : 0;
    if (workspace_bytes < lg_workspace_bytes(plan, B, train)) { lg_set_error("op_lgt: workspace too small"); return -3; }
    NetBufs nb;
    carve(plan, B, train, workspace, nb);
    return lgt_fwd(plan, params, stage, z, out, nb, B, flags, seed, (hipStream_t)stream);
}

extern "C" int lg_op_block(const lg_plan* plan, const float* params, int32_t stage, int32_t blk, int32_t which, const float* x,
                           float* y, void* workspace, size_t workspace_bytes, int32_t B, void* stream) {
    if (!plan || !params || !x || !y || !workspace || stage < 0 || stage >= plan->cfg.K || blk < 0 || blk > 4 || which < 0 || which > 2) {
        lg_set_error("op_block: invalid argument");
        return -1;
    }
    if (workspace_bytes < lg_workspace_bytes(plan, B, 0)) { lg_set_error("op_block: workspace too small"); return -3; }
    hipStream_t s = (hipStream_t)stream;
    NetBufs nb;
    carve(plan, B, 0, workspace, nb);
    BlockBufs bb = nb.blk[blk];
    int rc;
    const size_t npix = (size_t)B * bb.h * bb.w;
    if (which == 0 || which == 1) {
        // LN1 + planar split of the global half (normally emitted by the producing kernel's epilogue)
        if ((rc = launch_ln_split(bb.e, x, params + plan->blk(stage, blk, B_LN1G), params + plan->blk(stage, blk, B_LN1B), bb.g, B,
                                  bb.h * bb.w, s)))
            return rc;
    }
    if (which == 0) {
        FftArgs f;
        f.g = bb.g; f.o = y; f.amp = nullptr; f.pha = nullptr; f.sgn = nullptr; f.scratch = nb.fft_scratch;
        f.ampw = params + plan->blk(stage, blk, B_AMPW); f.ampb = params + plan->blk(stage, blk, B_AMPB);
        f.phaw = params + plan->blk(stage, blk, B_PHAW); f.phab = params + plan->blk(stage, blk, B_PHAB);
        f.ch = bb.e / 2; f.planes = B * f.ch; f.n = bb.h; f.h = bb.h; f.w = bb.w; f.full = plan->fft_full;
        return launch_fftmix(f, s);
    }
    if (which == 1) {
        float* posT = nb.posT;
        if ((rc = prep_stages(plan, params, stage, stage + 1, nb, s))) return rc;   // the mixer's static operand scales (and the stage's tables; this block's table goes to slot 0 below)
        if ((rc = launch_pos_transpose(params + plan->blk(stage, blk, B_POS), posT, s))) return rc;
        bb.xin = const_cast<float*>(x);
        bb.xmid = y;
        return block_mixer_fwd(plan, params, stage, blk, bb, posT, B, 0, 0, s, nb.fft_scratch, nb.attn_scales);
    }
    bb.xmid = const_cast<float*>(x);
    bb.xout = y;
    (void)npix;
    if ((rc = prep_stages(plan, params, stage, stage + 1, nb, s))) return rc;   // the FFN's operand scales
    return block_ffn_fwd(plan, params, stage, blk, bb, nullptr, 0, B, 0, s, nb.wsplit, nb.ffn_scales);
}

extern "C" int lg_op_block_bwd(const lg_plan* plan, const float* params, float* grads, int32_t stage, int32_t blk, int32_t which,
                               const float* x, const float* dy, float* dx, void* workspace, size_t workspace_bytes, int32_t B,
                               void* stream) {
    if (!plan || !params || !grads || !x || !dy || !dx || !workspace || stage < 0 || stage >= plan->cfg.K || blk < 0 || blk > 4 ||
        which < 0 || which > 2) {
        lg_set_error("op_block_bwd: invalid argument");
        return -1;
    }
    if (workspace_bytes < lg_workspace_bytes(plan, B, 1)) { lg_set_error("op_block_bwd: workspace too small"); return -3; }
    hipStream_t s = (hipStream_t)stream;
    NetBufs nb;
    carve(plan, B, 1, workspace, nb);
    BlockBufs& bb = nb.blk[blk];
    int rc;
    // forward of the half-block with everything saved
    if (which == 0 || which == 1) {
        if ((rc = launch_ln_split(bb.e, x, params + plan->blk(stage, blk, B_LN1G), params + plan->blk(stage, blk, B_LN1B), bb.g, B,
                                  bb.h * bb.w, s)))
            return rc;
        if ((rc = prep_stages(plan, params, stage, stage + 1, nb, s))) return rc;   // the mixer's static operand scales
        if ((rc = launch_pos_transpose(params + plan->blk(stage, blk, B_POS), nb.posT, s))) return rc;
        bb.xin = const_cast<float*>(x);
        if ((rc = block_mixer_fwd(plan, params, stage, blk, bb, nb.posT, B, LG_FLAG_SAVE, 0, s, nb.fft_scratch, nb.attn_scales))) return rc;
    } else {
        bb.xmid = const_cast<float*>(x);
        if ((rc = prep_stages(plan, params, stage, stage + 1, nb, s))) return rc;
        if ((rc = block_ffn_fwd(plan, params, stage, blk, bb, nullptr, 0, B, LG_FLAG_SAVE, s, nb.wsplit, nb.ffn_scales))) return rc;
    }
    return op_block_bwd(plan, params, grads, stage, blk, which, dy, dx, nb, (char*)workspace + nb.bytes, B, s);
}

extern "C" int lg_op_data_step_bwd(const lg_plan* plan, const float* params, float* grads, int32_t stage, const float* z_in,
                                   const float* ms, const float* pan, const float* dz_out, float* dz_in, void* workspace,
                                   size_t workspace_bytes, int32_t B, void* stream) {
    if (!plan || !params || !grads || !z_in || !ms || !pan || !dz_out || !dz_in || !workspace || B <= 0 || stage < 0 || stage >= plan->cfg.K) {
        lg_set_error("op_data_step_bwd: invalid argument");
        return -1;
    }
    if (workspace_bytes < lg_workspace_bytes(plan, B, 1)) { lg_set_error("op_data_step_bwd: workspace too small"); return -3; }
    hipStream_t s = (hipStream_t)stream;
    NetBufs nb;
    carve(plan, B, 1, workspace, nb);
    int rc;
    // forward of the step: fills the intermediates its backward reads (t1, r, s1)
    if ((rc = data_step_fwd(plan, params, stage, z_in, ms, pan, nb.Z[stage + 1], nb.t1[stage], nb.r[stage], nb.s1[stage], nb.pr, B, s))) return rc;
    return op_data_step_bwd(plan, params, grads, stage, nb, (char*)workspace + nb.bytes, z_in, pan, dz_out, dz_in, B, s);
}

extern "C" int lg_op_lgt_bwd(const lg_plan* plan, const float* params, float* grads, int32_t stage, const float* z, const float* dout,
                             float* dz, void* workspace, size_t workspace_bytes, int32_t B, int32_t flags, uint64_t seed, void* stream) {
    if (!plan || !params || !grads || !z || !dout || !dz || !workspace || B <= 0 || stage < 0 || stage >= plan->cfg.K) {
        lg_set_error("op_lgt_bwd: invalid argument");
        return -1;
    }
    if (flags & ~LG_FLAG_DROPOUT) { lg_set_error("op_lgt_bwd: only LG_FLAG_DROPOUT applies"); return -2; }
    if (workspace_bytes < lg_workspace_bytes(plan, B, 1)) { lg_set_error("op_lgt_bwd: workspace too small"); return -3; }
    hipStream_t s = (hipStream_t)stream;
    NetBufs nb;
    carve(plan, B, 1, workspace, nb);
    int rc;
    if ((rc = lgt_fwd(plan, params, stage, z, nb.deadout, nb, B, flags | LG_FLAG_SAVE, seed, s))) return rc;
    return op_lgt_bwd(plan, params, grads, stage, nb, (char*)workspace + nb.bytes, z, dout, dz, B, flags, seed, s);
}

// ------------------------------------------------------------------------------------------------
// L1 loss (mean) forward + backward -- models/base/losses.py:19-40, unlg_former.py:99-104
// ------------------------------------------------------------------------------------------------
// 16-byte accesses, four of them in flight per thread, and at most 256 workgroups: every workgroup ends in one float atomic on the
// loss scalar, and those serialise in L2 (the 1024-workgroup scalar-load form spent its 21 us there and in load latency)
__global__ __launch_bounds__(256) void k_dropout_mask(uint64_t seed, long first, long n, float* __restrict__ out) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) out[i] = dropout_scale(seed, (uint64_t)(first + i));
}
extern "C" int lg_dropout_mask(uint64_t seed, int32_t stage, int32_t blk, int64_t first, int64_t n, float* out, void* stream) {
    if (!out || n < 0 || first < 0 || stage < 0 || stage >= LG_MAX_K || blk < 0 || blk > 4) { lg_set_error("dropout_mask: invalid argument"); return -1; }
    if (n == 0) return 0;
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    k_dropout_mask<<<grid, 256, 0, (hipStream_t)stream>>>(mix_seed(seed, stage, blk), first, n, out);
    LG_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void k_l1(const float* __restrict__ out, const float* __restrict__ gt, float* __restrict__ dout,
                                            float* loss_accum, long n, float inv_n, float gscale) {
    float part = 0.f;
    const long n4 = n >> 2, stride = (long)gridDim.x * 256L;
    const float4* __restrict__ o4 = reinterpret_cast<const float4*>(out);
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(gt);
    float4* __restrict__ d4 = reinterpret_cast<float4*>(dout);
    auto one = [&](float d) { part += fabsf(d); return d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f); };
    long i = blockIdx.x * 256L + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = o4[i + u * stride]; b[u] = g4[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            d4[i + u * stride] = make_float4(one(a[u].x - b[u].x), one(a[u].y - b[u].y), one(a[u].z - b[u].z), one(a[u].w - b[u].w));
    }
    for (; i < n4; i += stride) {
        const float4 a = o4[i], b = g4[i];
        d4[i] = make_float4(one(a.x - b.x), one(a.y - b.y), one(a.z - b.z), one(a.w - b.w));
    }
    for (long j = 4 * n4 + blockIdx.x * 256L + threadIdx.x; j < n; j += stride) dout[j] = one(out[j] - gt[j]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_accum, (sm[0] + sm[1] + sm[2] + sm[3]) * inv_n);
}

extern "C" int lg_l1_loss(const float* out, const float* gt, float* dout, float* loss_accum, int64_t n_local, int64_t n_global,
                          float scale, void* stream) {
    if (!out || !gt || !dout || !loss_accum || n_local <= 0 || n_global <= 0) { lg_set_error("l1_loss: invalid argument"); return -1; }
    if (((uintptr_t)out | (uintptr_t)gt | (uintptr_t)dout) & 15) { lg_set_error("l1_loss: tensors must be 16-byte aligned"); return -1; }
    int grid = (int)((n_local / 4 + 255) / 256);
    if (grid > 256) grid = 256;
    if (grid < 1) grid = 1;
    k_l1<<<grid, 256, 0, (hipStream_t)stream>>>(out, gt, dout, loss_accum, n_local, 1.0f / (float)n_global, scale / (float)n_global);
    LG_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam single-tensor semantics) over ranges of the flat buffers
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, const int64_t* __restrict__ ranges, float step_size, float b1,
                                              float b2, float inv_bc2_sqrt, float eps, float gscale) {
    const int64_t lo = ranges[2 * blockIdx.y], hi = ranges[2 * blockIdx.y + 1];
    for (int64_t i = lo + blockIdx.x * 256L + threadIdx.x; i < hi; i += (int64_t)gridDim.x * 256L) {
        float gi = g[i] * gscale;
        float mi = b1 * m[i] + (1.0f - b1) * gi;
        float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

extern "C" int lg_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* ranges,
                            int32_t n_ranges, int64_t max_range, int32_t step, float lr, float beta1, float beta2, float eps,
                            float grad_scale, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !ranges || n_ranges <= 0 || step < 1) { lg_set_error("adam_step: invalid argument"); return -1; }
    double bc1 = 1.0 - pow((double)beta1, (double)step);
    double bc2 = 1.0 - pow((double)beta2, (double)step);
    int gx = (int)((max_range + 255) / 256);
    if (gx < 1) gx = 1;
    if (gx > 512) gx = 512;
    dim3 grid(gx, n_ranges);
    k_adam<<<grid, 256, 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, ranges, (float)(lr / bc1), beta1, beta2,
                                                   (float)(1.0 / sqrt(bc2)), eps, grad_scale);
    LG_CHECK_LAUNCH();
    return 0;
}
