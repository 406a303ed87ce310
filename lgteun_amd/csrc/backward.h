// Backward orchestration (autograd of Pansharpening.forward, reference models/unlg_former.py:50-67).
#pragma once
#include "common.h"
#include "workspace.h"

size_t bwd_workspace_bytes(const lg_plan* plan, int B);
int net_backward(const lg_plan* plan, const float* params, float* grads, const float* ms, const float* pan, const float* dout,
                 NetBufs& nb, void* bwd_ws, int B, int flags, uint64_t seed, hipStream_t s);
int op_block_bwd(const lg_plan* plan, const float* params, float* grads, int stage, int blk, int which, const float* dy, float* dx,
                 NetBufs& nb, void* bwd_ws, int B, hipStream_t s);
int op_data_step_bwd(const lg_plan* plan, const float* params, float* grads, int stage, NetBufs& nb, void* bwd_ws, const float* z_in,
                     const float* pan, const float* g, float* dz, int B, hipStream_t s);
int op_lgt_bwd(const lg_plan* plan, const float* params, float* grads, int stage, NetBufs& nb, void* bwd_ws, const float* z, const float* dout,
               float* dz, int B, int flags, uint64_t seed, hipStream_t s);
