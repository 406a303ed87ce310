"""Helpers for the -m gpu parity tests: build the product module with deterministic weights and call the
C ABI per-op entry points."""
import ctypes

import numpy as np
import torch

import lgteun_amd
from lgteun_amd import _lib
from lgteun_amd.compat import Config
from lgteun_amd.engine import _ptr, _stream_ptr
from oracle import detweights as dw

from helpers import state_shapes


def make_module(C, K, salt=0, device='cuda'):
    net = lgteun_amd.Pansharpening(Config(ms_chans=C), None, stage=K)
    sd = dw.fill_state_dict(state_shapes(C, K), salt=salt, dtype=np.float32)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to(device)
    net.eval()
    return net


class Ops:
    def __init__(self, net, H, W):
        self.net = net
        self.eng = net.engine()
        self.lib = self.eng.lib
        self.plan = self.eng.plan(H, W)   # precision follows net.precision
        self.H, self.W = H, W

    def ws(self, B, train=False):
        return self.eng.workspace(self.plan, B, train)

    def resample(self, x, mode):
        planes = x.shape[0] * x.shape[1]
        hi, wi = x.shape[2], x.shape[3]
        f = {0: 0.5, 1: 2, 2: 4}[mode]
        y = torch.empty(x.shape[0], x.shape[1], int(hi * f), int(wi * f), device=x.device)
        _lib.check(self.lib.lg_op_resample(_ptr(x), _ptr(y), planes, hi, wi, mode, _stream_ptr()), 'lg_op_resample')
        return y

    def data_step(self, stage, z, ms, pan):
        B = z.shape[0]
        out = torch.empty_like(z)
        tmp = torch.empty(3 * z.numel() // 4 + z.numel() // z.shape[1], device=z.device)   # include/lgteun_hip.h: lg_op_data_step
        _lib.check(self.lib.lg_op_data_step(self.plan, _ptr(self.eng.flat), stage, _ptr(z), _ptr(ms), _ptr(pan), _ptr(out),
                                            _ptr(tmp), B, _stream_ptr()), 'lg_op_data_step')
        return out

    def lgt(self, stage, z):
        B = z.shape[0]
        out = torch.empty_like(z)
        ws = self.ws(B)
        _lib.check(self.lib.lg_op_lgt(self.plan, _ptr(self.eng.flat), stage, _ptr(z), _ptr(out), _ptr(ws), ws.numel(), B, 0, 0,
                                      _stream_ptr()), 'lg_op_lgt')
        return out

    def block(self, stage, blk, which, x):
        """x NHWC [B,h,w,e]; which 0: global mixer (planar out), 1: mixer half-block, 2: ffn half-block"""
        B, h, w, e = x.shape
        y = torch.empty(B, e // 2, h, w, device=x.device) if which == 0 else torch.empty_like(x)
        ws = self.ws(B)
        _lib.check(self.lib.lg_op_block(self.plan, _ptr(self.eng.flat), stage, blk, which, _ptr(x), _ptr(y), _ptr(ws),
                                        ws.numel(), B, _stream_ptr()), 'lg_op_block')
        return y

    def block_bwd(self, stage, blk, which, x, dy):
        """returns (dx, flat_param_grads).  which 0: dy/dx planar [B,e/2,h,w]; 1,2: NHWC."""
        B = x.shape[0]
        dx = torch.empty_like(dy)
        grads = torch.zeros_like(self.eng.flat)
        ws = self.ws(B, train=True)
        _lib.check(self.lib.lg_op_block_bwd(self.plan, _ptr(self.eng.flat), _ptr(grads), stage, blk, which, _ptr(x), _ptr(dy),
                                            _ptr(dx), _ptr(ws), ws.numel(), B, _stream_ptr()), 'lg_op_block_bwd')
        return dx, grads

    def data_step_bwd(self, stage, z, ms, pan, dz_out):
        """returns (dz_in, flat_param_grads) of one data step"""
        B = z.shape[0]
        dz = torch.empty_like(z)
        grads = torch.zeros_like(self.eng.flat)
        ws = self.ws(B, train=True)
        _lib.check(self.lib.lg_op_data_step_bwd(self.plan, _ptr(self.eng.flat), _ptr(grads), stage, _ptr(z), _ptr(ms), _ptr(pan),
                                                _ptr(dz_out), _ptr(dz), _ptr(ws), ws.numel(), B, _stream_ptr()), 'lg_op_data_step_bwd')
        return dz, grads

    def lgt_bwd(self, stage, z, dout):
        """returns (dz, flat_param_grads) of one LGT (dropout off)"""
        B = z.shape[0]
        dz = torch.empty_like(z)
        grads = torch.zeros_like(self.eng.flat)
        ws = self.ws(B, train=True)
        _lib.check(self.lib.lg_op_lgt_bwd(self.plan, _ptr(self.eng.flat), _ptr(grads), stage, _ptr(z), _ptr(dout), _ptr(dz), _ptr(ws),
                                          ws.numel(), B, 0, 0, _stream_ptr()), 'lg_op_lgt_bwd')
        return dz, grads

    def grad_of(self, flat_grads, name):
        i = self.eng.names.index(name)
        o, p = self.eng.offsets[i], self.eng.params[i]
        return flat_grads[o:o + p.numel()].view(p.shape)
