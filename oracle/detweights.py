"""Deterministic, name-seeded parameter fill (TEST INFRASTRUCTURE, see oracle/__init__.py).

Fixtures never contain weights: every parameter is regenerated from its state_dict key by a
counter hash (splitmix64), so the reference net (in the build container) and the build's net
(anywhere) can be loaded with bit-identical values.  SURVEY.md §8(c) "Deterministic weights".
"""
import hashlib

import numpy as np

_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_M64)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_M64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_M64)
    return z ^ (z >> np.uint64(31))


def uniform01(name, n, salt=0):
    """n doubles in [0,1) determined by (name, salt)."""
    h = int.from_bytes(hashlib.sha256(f'{name}#{salt}'.encode()).digest()[:8], 'little')
    with np.errstate(over='ignore'):
        idx = (np.arange(n, dtype=np.uint64) + np.uint64(h)) & np.uint64(_M64)
        z = _splitmix64(idx)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _scale_for(name, shape):
    """Value ranges chosen to resemble PyTorch default init magnitudes for each kind of key."""
    if name.endswith('pos_emb'):
        return ('sym', 1.0)                       # trunc_normal(0,1,[-2,2]) magnitude
    if name.startswith('eta.'):
        return ('range', 0.05, 0.15)             # init 0.1
    if '.norm.weight' in name:
        return ('range', 0.7, 1.3)               # LayerNorm gamma, perturbed from 1
    if '.norm.bias' in name:
        return ('sym', 0.2)
    if 'conv_amp' in name or 'conv_pha' in name:
        return ('sym', 1.0)                       # dw 1x1: fan_in = 1 -> U(-1,1)
    if name.endswith('.weight'):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        return ('sym', 1.0 / np.sqrt(max(fan_in, 1)))
    if name.endswith('.bias'):
        return ('sym', 0.3)
    return ('sym', 0.5)


def fill_state_dict(shapes, salt=0, dtype=np.float32):
    """shapes: dict name -> shape tuple.  Returns dict name -> np.ndarray."""
    out = {}
    for name, shape in shapes.items():
        n = int(np.prod(shape)) if len(shape) else 1
        u = uniform01(name, n, salt)
        kind = _scale_for(name, shape)
        if kind[0] == 'sym':
            v = (2.0 * u - 1.0) * kind[1]
        else:
            v = kind[1] + (kind[2] - kind[1]) * u
        out[name] = v.reshape(shape).astype(dtype)
    return out


def make_inputs(B, C, h, w, seed=19971118, kind='dn', dtype=np.float32):
    """Synthetic LrMS / PAN / target batch.  kind='dn': 11-bit integer DN / 2047.5
    (reference dataset/utils.py:232-249, bit_depth=11 configs/unlg_former.py:41);
    kind='smooth': low-frequency field + noise so FFT bins are not all white."""
    rng = np.random.default_rng(seed)
    H, W = 4 * h, 4 * w
    if kind == 'dn':
        ms = rng.integers(0, 2048, size=(B, C, h, w)).astype(np.float64) / 2047.5
        pan = rng.integers(0, 2048, size=(B, 1, H, W)).astype(np.float64) / 2047.5
        gt = rng.integers(0, 2048, size=(B, C, H, W)).astype(np.float64) / 2047.5
    else:
        yy, xx = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing='ij')
        base = np.zeros((B, C, H, W))
        for b in range(B):
            for c in range(C):
                f1, f2, p1, p2 = rng.uniform(0.5, 4.0, 4)
                base[b, c] = 0.5 + 0.25 * np.sin(2 * np.pi * f1 * yy + p1) * np.cos(2 * np.pi * f2 * xx + p2)
        gt = np.clip(base + 0.03 * rng.standard_normal(base.shape), 0, 1)
        pan = gt.mean(axis=1, keepdims=True) + 0.01 * rng.standard_normal((B, 1, H, W))
        ms = gt.reshape(B, C, h, 4, w, 4).mean(axis=(3, 5)) + 0.01 * rng.standard_normal((B, C, h, w))
    return ms.astype(dtype), pan.astype(dtype), gt.astype(dtype)
