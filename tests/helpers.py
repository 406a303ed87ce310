"""Shared test helpers: deterministic weights + shapes of the reference state_dict."""
import numpy as np
import torch

from oracle import detweights as dw


def state_shapes(C, K):
    """Shapes of Pansharpening.state_dict() (reference models/unlg_former.py:22-48;
    measured key list in SURVEY.md §8b) -- rebuilt here from the architecture definition."""
    E = 4 * C
    s = {}
    for n in ('D.1', 'D.3', 'DT.1', 'DT.3'):
        s[n + '.weight'] = (C, 1, 3, 3)
        s[n + '.bias'] = (C,)
    s['R.weight'] = (1, C, 1, 1)
    s['R.bias'] = (1,)
    s['RT.weight'] = (C, 1, 1, 1)
    s['RT.bias'] = (C,)
    for i in range(K):
        s[f'eta.{i}'] = ()

    def block(pre, e):
        h = e // 2
        m = pre + '0.fn.'
        s[m + 'fn.local_mixer.pos_emb'] = (1, 2, 64, 64)
        s[m + 'fn.local_mixer.to_qkv.weight'] = (3 * h, h, 1, 1)
        s[m + 'fn.local_mixer.to_qkv.bias'] = (3 * h,)
        for n in ('conv_amp', 'conv_pha'):
            s[m + f'fn.global_mixer.{n}.0.weight'] = (h, 1, 1, 1)
            s[m + f'fn.global_mixer.{n}.0.bias'] = (h,)
        s[m + 'fn.proj.weight'] = (e, e, 1, 1)
        s[m + 'fn.proj.bias'] = (e,)
        s[m + 'norm.weight'] = (e,)
        s[m + 'norm.bias'] = (e,)
        f = pre + '1.fn.'
        s[f + 'fn.net.0.weight'] = (4 * e, e, 1, 1)
        s[f + 'fn.net.0.bias'] = (4 * e,)
        s[f + 'fn.net.2.point_conv.weight'] = (4 * e, 4 * e, 1, 1)
        s[f + 'fn.net.2.point_conv.bias'] = (4 * e,)
        s[f + 'fn.net.2.depth_conv.weight'] = (4 * e, 1, 3, 3)
        s[f + 'fn.net.2.depth_conv.bias'] = (4 * e,)
        s[f + 'fn.net.4.weight'] = (e, 4 * e, 1, 1)
        s[f + 'fn.net.4.bias'] = (e,)
        s[f + 'norm.weight'] = (e,)
        s[f + 'norm.bias'] = (e,)

    for i in range(K):
        p = f'prior_module.{i}.'
        s[p + 'patch_embed.proj.0.weight'] = (C, 1, 1, 1)
        s[p + 'patch_embed.proj.0.bias'] = (C,)
        s[p + 'patch_embed.proj.1.weight'] = (E, C, 1, 1)
        s[p + 'patch_embed.proj.1.bias'] = (E,)
        s[p + 'patch_embed.norm.weight'] = (E,)
        s[p + 'patch_embed.norm.bias'] = (E,)
        for j in range(2):
            block(p + f'encoder_layers.0.0.blocks.{j}.', E)
        s[p + 'encoder_layers.0.1.1.weight'] = (2 * E, E, 1, 1)
        s[p + 'encoder_layers.0.1.1.bias'] = (2 * E,)
        block(p + 'bottleneck.blocks.0.', 2 * E)
        s[p + 'decoder_layers.0.0.1.weight'] = (E, 2 * E, 1, 1)
        s[p + 'decoder_layers.0.0.1.bias'] = (E,)
        s[p + 'decoder_layers.0.1.weight'] = (E, 2 * E, 1, 1)
        s[p + 'decoder_layers.0.1.bias'] = (E,)
        for j in range(2):
            block(p + f'decoder_layers.0.2.blocks.{j}.', E)
        s[p + 'tail.1.weight'] = (C, E, 1, 1)
        s[p + 'tail.1.bias'] = (C,)
    return s


def det_params(C, K, salt=0, dtype=torch.float32, requires_grad=False):
    sd = dw.fill_state_dict(state_shapes(C, K), salt=salt, dtype=np.float64)
    P = {k: torch.from_numpy(v).to(dtype) for k, v in sd.items()}
    if requires_grad:
        for v in P.values():
            v.requires_grad_(True)
    return P


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
