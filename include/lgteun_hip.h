/*
 * lgteun_hip.h -- C ABI of the MI355X-native (gfx950) LGTEUN unfolding hot path.
 *
 * The reference (lms-07/LGTEUN) has no FFI: its hot path is a Python nn.Module,
 *   Pansharpening.forward            models/unlg_former.py:50-67
 *   LGT.forward and its sub-modules  models/common/LGT.py:64-344
 *   sampling_/dep_conv/point_conv    models/common/basic_module_unformer_v2.py:13-53
 *   L1 loss + Adam + StepLR          models/unlg_former.py:87-113, models/base/base_model.py:116-147
 * This library replaces the ATen op sequences behind those lines.  It is bound from Python with
 * ctypes (lgteun_amd/_lib.py); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked host.
 *  - the caller (PyTorch) owns all memory: parameters, gradients, optimizer state, workspace.
 *    The library allocates no device memory.  Its only mutable process-global state is the event table of the opt-in
 *    timing facility (lg_prof_*, mutex-protected; off by default) and per-device "kernel attribute set" bits; plans are
 *    immutable after creation, so forward / backward calls on different streams or threads do not interact.
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*); no implicit sync.
 *  - return value: 0 ok; <0 invalid argument / unsupported shape (see lg_last_error());
 *    >0 a hipError_t.  Nothing throws across the boundary.
 *  - parameters live in ONE flat fp32 buffer; `offsets[i]` (in floats) locates the i-th tensor of
 *    Pansharpening.state_dict() in its canonical order (12 shared, K eta, 119 per stage;
 *    SURVEY.md section 8b).  Gradients use the same offsets in a second flat buffer.
 *  - module I/O is NCHW fp32 like the reference: ms [B,C,h,w], pan [B,1,4h,4w], out [B,C,4h,4w].
 */
#ifndef LGTEUN_HIP_H
#define LGTEUN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LG_TENSORS_SHARED 12
#define LG_TENSORS_PER_STAGE 119

/* flags for lgteun_forward */
#define LG_FLAG_FAITHFUL 1 /* run every stage's LGT like the reference does (results of stages 0..K-2 are dead) */
#define LG_FLAG_SAVE 2     /* keep what lgteun_backward needs in the workspace */
#define LG_FLAG_DROPOUT 4  /* training-mode Dropout(0.1) after LGMixer.proj (LGT.py:198,215), counter-hash RNG */

/* flags for lgteun_backward: run only a part (lets the caller start the gradient all-reduce of the LGT bucket
 * while the data-step backward still runs).  Neither bit set = both parts. */
#define LG_FLAG_BWD_LGT 8
#define LG_FLAG_BWD_DATA 16

/* forward + backward: the INTENDED unfolding, default off: stage i+1's data step consumes LGT_i's output
 * instead of the data step's own output (the reference feeds `Z`, not `Z_`, forward: unlg_former.py:56-67), so every stage's
 * LGT is live, every parameter gets a gradient, and training keeps one saved activation set per stage
 * (lg_workspace_bytes(..., train = 2)).  Overrides LG_FLAG_FAITHFUL; LG_FLAG_BWD_LGT / _DATA do not apply (the K LGT and
 * data-step backwards interleave) and are rejected. */
#define LG_FLAG_CHAINED 32

/* forward, with LG_FLAG_FAITHFUL: leave the K-1 dead-stage LGT forwards (the reference executes them and discards the
 * result) to a later lgteun_dead_forward call.  They depend on nothing but the data-step outputs, so a training step can enqueue them
 * on a second stream BEHIND the LGT backward (they reuse the LGT activation buffers) and beside the K data-step backwards + Adam --
 * a chain of ~40 small latency-bound launches that leaves most of the GPU idle.  Same kernels, same work, same results. */
#define LG_FLAG_DEFER_DEAD 64

/* kernel ids for the live HIP-event timing facility (lg_prof_*) */
enum lg_kernel_id {
    LG_K_NONE = 0, LG_K_FFN1, LG_K_FFN2, LG_K_FFT, LG_K_ATTN, LG_K_UPFUSE, LG_K_DOWN, LG_K_EMBED, LG_K_TAIL, LG_K_DATASTEP,
    LG_K_FFN1_BWD, LG_K_FFN2_BWD, LG_K_FFT_BWD, LG_K_ATTN_BWD, LG_K_WGRAD, LG_K_COUNT
};

typedef struct lg_config {
    int32_t C;       /* MS bands: 4 or 8                      (cfg.ms_chans, unlg_former.py:24) */
    int32_t K;       /* unfolding stages                      (stage kwarg, unlg_former.py:22)  */
    int32_t H, W;    /* PAN size = 4 x MS size: multiples of 16, 16..1024 (LGT.py needs 8-px windows at both levels).  Square powers
                      * of two up to 512 take the radix-2 FFT paths (plane in LDS up to 128, split above); everything else
                      * (400x400 full-resolution scenes, rectangles) the Bluestein path -- same results, slower mixer */
    int32_t precision; /* 0 = fp32 storage/compute (parity mode); 1 = bf16 storage of FFN hidden tensors */
    uint32_t variant;  /* 0 = the product path.  LG_VAR_* bits select A/B kernels that compute the SAME function (tests compare them with
                        * the default; the library itself reads no environment variable).  Bits this build does not carry are rejected. */
} lg_config;

/* lg_config.variant: A/B switches (all default off) */
#define LG_VAR_FFN_IMPL_MASK 3u   /* fused FFN forward: 0 = split-bf16 kernels; 1 = the exact f32-MFMA strip kernel (v_mfma_f32_16x16x4_f32: bit
                                   * for bit an fp32 fma chain -- the yardstick of the arithmetic-criterion test); 2 = round 1's per-tile
                                   * f32-MFMA kernel, 3 = the software-pipelined split kernel (both only in `make AB=1` builds) */
#define LG_VAR_FFN_STRIP 1u
#define LG_VAR_FFN_TILE 2u
#define LG_VAR_FFN_XP 3u
#define LG_VAR_FFN_SAVE_MASK (3u << 2) /* what the live stage's e = 16 FFN keeps for the backward: 0 = h2, h3 (default); 1 = h1, h2, h3; 2 = the
                                        * five-tensor GELU-free form */
#define LG_VAR_FFN_SAVE3 (1u << 2)
#define LG_VAR_FFN_SAVE5 (2u << 2)
#define LG_VAR_FFN_BWD32_PAIR (1u << 4) /* e = 32 FFN backward, pixelwise half: round 2's k_ffn1_bwd_x32 + two weight-gradient launches instead of k_ffn1_bwd_xs<32> */
#define LG_VAR_FFN_DWBWD_TILE (1u << 5) /* e = 16 FFN backward, spatial half: round 2's tile kernel + weight-gradient launch */
#define LG_VAR_ATTN_BWD_R3 (1u << 6)    /* e = 16 local-mixer backward: round 3's three-kernel form instead of k_attn_bwd_f */
#define LG_VAR_DSTEP_TILES (1u << 7)    /* data step: the tile kernels (four launches forward, nine backward) also where the one-launch form exists */
#define LG_VAR_ATTN_FWD_VALU (1u << 8)  /* local-mixer forward: round 2's vector-pipe kernel k_attn (lane = token, fp32 FMAs) instead of the matrix-pipe k_attn_m */
#define LG_VAR_FFN_BF16X3 (1u << 9)     /* fused FFN forward: round 2's three-piece bf16 split (six products) instead of the f16 pairs (three products) */
#define LG_VAR_FFT_FULL (1u << 10)     /* global mixer on planes up to 128 x 128: round 1-4's complex-row in-LDS kernels k_fftmix / k_fftmix_bwd instead of the real-input k_fftmix_r / k_fftmix_bwd_r */
#define LG_VAR_FFN_BWD_BF16X3 (1u << 11) /* FFN backward (k_ffn1_bwd_xs): three bf16 pieces / six products (rounds 3 - 4) instead of f16 pairs / three products with scaled operands */
#define LG_VAR_ATTN_BWD_CORE_M (1u << 12) /* e = 32 local-mixer backward core: the matrix-pipe k_attn_bwd_core_m (round 5; same results, not faster yet: DESIGN.md 3.3) instead of the vector-pipe k_attn_bwd_core */
#define LG_VAR_FFN_XS (1u << 13)        /* fused FFN forward at e = 16: the channel-split k_ffn_xs of rounds 2 - 5 (LN(x) / gelu(h1) pieces through LDS, eleven barriers per step) instead of the register-chain k_ffn_xr (round 6) */
#define LG_VAR_ATTN_BF16X3 (1u << 14)  /* local-mixer forward (k_attn_m): to_qkv and Q K^T on three bf16 pieces / six products (round 5) instead of f16 pairs with static operand scales (round 6) */
#define LG_VAR_FFN_H3_RECOMPUTE (1u << 15)  /* e = 16 FFN of the live stage: save h2 ONLY and re-compute h3 in the backward (k_ffn_dw_bwd_h, round 6: the saving forward launch -17 us, the backward kernel +70 us: opt-in) instead of saving h2 and h3 (k_ffn_dw_bwd_xs) */
#define LG_VAR_ATTN_BWD_RESTATS (1u << 16)  /* e = 16 local mixer of the live stage: k_attn_bwd_f re-derives the softmax row statistics and the attention output with a reduction pass of its own (rounds 4 - 5) instead of reading what k_attn_m's saving launch left (round 6: log-sum-exp + attention output, 40 bytes per pixel) */
#define LG_VAR_ALL 0x1ffffu

typedef struct lg_plan lg_plan; /* host-side, immutable after creation */

const char* lg_version(void);
/* Bumped whenever a struct layout, an argument meaning or a caller-provided buffer size changes (2: lg_config.variant, the data step's
 * tmp of 3*B*C*H*W/4 + B*H*W floats).  A binding checks it at load time -- lgteun_amd/_lib.py does -- instead of passing a stale struct. */
#define LG_ABI_VERSION 2
int32_t lg_abi_version(void);
const char* lg_last_error(void); /* thread-local, host string */

/* offsets: host array of n_offsets = 12 + K + 119*K int64 (float offsets into the flat parameter buffer). */
int lg_plan_create(const lg_config* cfg, const int64_t* offsets, int32_t n_offsets, lg_plan** out);
void lg_plan_destroy(lg_plan* plan);
/* bytes of workspace lgteun_forward/backward need for batch B: train = 0 forward only, 1 = forward with LG_FLAG_SAVE + backward,
 * 2 = the same with LG_FLAG_CHAINED (K saved activation sets). */
size_t lg_workspace_bytes(const lg_plan* plan, int32_t B, int32_t train);

/* Pansharpening.forward (unlg_former.py:50-67).  seed: dropout counter seed (used with LG_FLAG_DROPOUT). */
int lgteun_forward(const lg_plan* plan, const float* params, const float* ms, const float* pan, float* out,
                   void* workspace, size_t workspace_bytes, int32_t B, int32_t flags, uint64_t seed, void* stream);

/* The dead-stage LGT forwards a forward with LG_FLAG_FAITHFUL | LG_FLAG_DEFER_DEAD left out (unlg_former.py:63-67 for stages
 * 0..K-2): same workspace, B, flags and seed as that forward.  Overwrites the LGT activation set, so it must be ordered behind the
 * LGT part of lgteun_backward (LG_FLAG_BWD_LGT) and before the next forward on the workspace; reads only dead-stage parameters
 * (Adam never touches those) and the data-step outputs. */
int lgteun_dead_forward(const lg_plan* plan, const float* params, void* workspace, size_t workspace_bytes, int32_t B, int32_t flags,
                        uint64_t seed, void* stream);

/* Backward of the same graph (autograd of unlg_former.py:50-67): needs the workspace of a forward run
 * with LG_FLAG_SAVE.  dout [B,C,H,W].  Accumulates (+=) into `grads` for the live tensors only
 * (shared D/DT/R/RT, eta, last stage's LGT); dead-stage slots are never written. */
int lgteun_backward(const lg_plan* plan, const float* params, float* grads, const float* ms, const float* pan,
                    const float* dout, void* workspace, size_t workspace_bytes, int32_t B, int32_t flags,
                    uint64_t seed, void* stream);

/* nn.L1Loss(mean) forward+backward (losses.py:19-40; unlg_former.py:99-104): loss_sum[0] += sum|out-gt|/N and
 * dout = sign(out-gt) * scale / N.   N = number of elements of the GLOBAL batch (DDP: pass n_global). */
int lg_l1_loss(const float* out, const float* gt, float* dout, float* loss_accum, int64_t n_local, int64_t n_global,
               float scale, void* stream);

/* torch.optim.Adam step (base_model.py:123-124; no weight decay / amsgrad) over [begin,end) float ranges of the
 * flat buffers.  ranges: DEVICE int64 pairs, n_ranges of them (the live tensors).  step is 1-based. */
int lg_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* ranges,
                 int32_t n_ranges, int64_t max_range, int32_t step, float lr, float beta1, float beta2, float eps,
                 float grad_scale, void* stream);

/* Live per-kernel timing: when enabled for `kernel_id`, every launch of that kernel is bracketed by hipEvents recorded on
 * the stream it is launched on.  lg_prof_read synchronises on the recorded events and returns the summed device time (ms)
 * and the number of launches since lg_prof_enable / lg_prof_reset.  Host-side event objects are the only thing the library
 * ever creates; lg_prof_disable destroys them. */
int lg_prof_enable(int32_t kernel_id, int32_t max_launches);
int lg_prof_reset(void);
void lg_prof_pause(int32_t paused);   /* 1: launches are neither timed nor counted until lg_prof_pause(0) (sampling: an event pair costs ~2 us of stream time) */
int lg_prof_read(double* total_ms, int64_t* launches);
void lg_prof_disable(void);
const char* lg_kernel_name(int32_t kernel_id);

/* The dropout the LGMixers apply in train mode (nn.Dropout(0.1) behind proj, LGT.py:197-198,215): out[i] = 0 or 1/0.9 = the factor of
 * element i (= pixel * e + channel, NHWC) of block `blk` of stage `stage` under `seed` -- a counter hash, so forward and backward draw the
 * same mask without storing it, and an integrator can reproduce it.  n elements from index `first`; out is a device pointer. */
int lg_dropout_mask(uint64_t seed, int32_t stage, int32_t blk, int64_t first, int64_t n, float* out, void* stream);

/* ---- per-op entry points (unit-tested against the oracle; same kernels the orchestrators launch) ---- */
/* bmu.sampling_ bicubic (basic_module_unformer_v2.py:21-23): mode 0: x0.5, 1: x2, 2: x4.  x [planes,hi,wi]. */
int lg_op_resample(const float* x, float* y, int32_t planes, int32_t hi, int32_t wi, int32_t mode, void* stream);
/* one data step (unlg_former.py:58-61) for stage `stage`: z_in -> z_out [B,C,H,W]; tmp: 3*B*C*H*W/4 + B*H*W floats (the chain's three
 * intermediates t1 | r | . | s1 in quarters of B*C*H*W/4, then the per-sample plane R Z - pan of the one-launch form). */
int lg_op_data_step(const lg_plan* plan, const float* params, int32_t stage, const float* z_in, const float* ms,
                    const float* pan, float* z_out, float* tmp, int32_t B, void* stream);
/* one LGT forward (LGT.py:314-344) with stage `stage`'s weights: z [B,C,H,W] -> out [B,C,H,W]. */
int lg_op_lgt(const lg_plan* plan, const float* params, int32_t stage, const float* z, float* out, void* workspace,
              size_t workspace_bytes, int32_t B, int32_t flags, uint64_t seed, void* stream);
/* pieces of one LGB block `blk` (0,1: encoder; 2: bottleneck; 3,4: decoder) of stage `stage`, on NHWC x:
 *  which = 0: global_mixer on LN(x)[..., e/2:]  -> y planar [B,e/2,h,w]        (LGT.py:149-180)
 *          1: x + LGMixer(LN(x))                -> y [B,h,w,e]                 (LGT.py:183-219,231-248)
 *          2: x + feed_forward(LN(x))           -> y [B,h,w,e]                 (LGT.py:91-109)
 *  h,w,e are implied by blk (level 0: H,W,4C; level 1: H/2,W/2,8C). */
int lg_op_block(const lg_plan* plan, const float* params, int32_t stage, int32_t blk, int32_t which, const float* x,
                float* y, void* workspace, size_t workspace_bytes, int32_t B, void* stream);

/* backward of the same pieces (autograd of the lines cited at lg_op_block): runs the half-block forward on x with
 * everything saved, then its backward for upstream gradient dy.  which = 0: dy, dx planar [B,e/2,h,w] (gradient wrt the
 * LayerNorm-ed global half); 1, 2: dy, dx NHWC [B,h,w,e].  Parameter gradients accumulate (+=) into grads.
 * workspace: lg_workspace_bytes(plan, B, 1). */
int lg_op_block_bwd(const lg_plan* plan, const float* params, float* grads, int32_t stage, int32_t blk, int32_t which,
                    const float* x, const float* dy, float* dx, void* workspace, size_t workspace_bytes, int32_t B,
                    void* stream);

/* backward of one data step (autograd of unlg_former.py:58-61 with D :29-30, DT :32-33, R :36, RT :37): runs the step's forward
 * on z_in (to have its intermediates), then maps dz_out (gradient wrt the step's output) to dz_in (gradient wrt z_in, including the
 * identity path) and accumulates (+=) the gradients of D / DT / R / RT and eta[stage] into grads.  workspace:
 * lg_workspace_bytes(plan, B, 1). */
int lg_op_data_step_bwd(const lg_plan* plan, const float* params, float* grads, int32_t stage, const float* z_in, const float* ms,
                        const float* pan, const float* dz_out, float* dz_in, void* workspace, size_t workspace_bytes, int32_t B,
                        void* stream);
/* backward of one LGT (autograd of LGT.py:314-344; covers patch_embedding :64-88, down :280-281, up + fusion :294-295,337-338 and
 * tail :302-303,342 besides the five blocks): runs the LGT forward on z with everything saved, then maps dout to dz (gradient wrt z)
 * and accumulates (+=) the 119 parameter gradients of stage `stage` into grads.  flags: 0 or LG_FLAG_DROPOUT (with seed). */
int lg_op_lgt_bwd(const lg_plan* plan, const float* params, float* grads, int32_t stage, const float* z, const float* dout, float* dz,
                  void* workspace, size_t workspace_bytes, int32_t B, int32_t flags, uint64_t seed, void* stream);

#ifdef __cplusplus
}
#endif
#endif
