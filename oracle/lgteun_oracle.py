"""CPU restatement of the LGTEUN unfolding hot path.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Written from scratch as explicit tap / index arithmetic on torch CPU tensors (so that every
formula here is also the specification of one HIP kernel), functional over a plain dict of
parameters keyed by the reference's state_dict names.  Works in fp32 or fp64 (dtype of inputs).
torch autograd through these explicit ops is the gradient oracle.

Each function cites the reference lines it restates (paths relative to /root/reference).
Third-party arithmetic used as-is: torch.fft.rfft2/irfft2 (pocketfft), torch.erf, torch.atan2.

Parity pin: tests/golden/*.npz were produced by running the reference itself in the build
container (tools/gen_goldens.py); tests/test_oracle_golden.py checks this file against them.
"""
import math

import torch

# --------------------------------------------------------------------------------------
# bicubic resampling: F.interpolate(mode='bicubic', align_corners=False,
# recompute_scale_factor=False)  -- models/common/basic_module_unformer_v2.py:21-34
# --------------------------------------------------------------------------------------
_A = -0.75


def _cubic_w(t):
    """4 cubic-convolution weights (A=-0.75) for fractional offset t, taps at i0-1..i0+2."""
    def c1(x):
        return ((_A + 2.0) * x - (_A + 3.0)) * x * x + 1.0

    def c2(x):
        return ((_A * x - 5.0 * _A) * x + 8.0 * _A) * x - 4.0 * _A
    return [c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)]


def bicubic_plan(n_in, scale):
    """Index/weight plan of the 1-D resampler: for each output o, 4 clamped source indices and
    4 weights.  src = (o + 0.5) / scale - 0.5 ; i0 = floor(src) ; t = src - i0."""
    n_out = int(math.floor(n_in * scale))
    idx = torch.empty(n_out, 4, dtype=torch.long)
    wts = torch.empty(n_out, 4, dtype=torch.float64)
    for o in range(n_out):
        src = (o + 0.5) / scale - 0.5
        i0 = math.floor(src)
        t = src - i0
        w = _cubic_w(t)
        for k in range(4):
            idx[o, k] = min(max(i0 - 1 + k, 0), n_in - 1)
            wts[o, k] = w[k]
    return idx, wts


def resample(x, scale):
    """x: [B,C,H,W] -> [B,C,H*scale,W*scale]; separable clamped 4-tap polyphase filter."""
    if scale == 1:
        return x
    B, C, H, W = x.shape
    iy, wy = bicubic_plan(H, scale)
    ix, wx = bicubic_plan(W, scale)
    wy = wy.to(x.dtype)
    wx = wx.to(x.dtype)
    # rows
    t = x[:, :, iy.reshape(-1), :].reshape(B, C, iy.shape[0], 4, W)
    t = (t * wy.view(1, 1, -1, 4, 1)).sum(dim=3)
    # cols
    u = t[:, :, :, ix.reshape(-1)].reshape(B, C, t.shape[2], ix.shape[0], 4)
    return (u * wx.view(1, 1, 1, -1, 4)).sum(dim=4)


# --------------------------------------------------------------------------------------
# convs  -- basic_module_unformer_v2.py:13-18
# --------------------------------------------------------------------------------------
def point_conv(x, w, b):
    """dense 1x1 conv on NCHW: w [Cout,Cin,1,1]."""
    return torch.einsum('oc,bchw->bohw', w[:, :, 0, 0], x) + b.view(1, -1, 1, 1)


def dep_conv(x, w, b):
    """depthwise k x k conv (k=1 or 3), zero padding k//2, NCHW: w [C,1,k,k]."""
    k = w.shape[-1]
    if k == 1:
        return x * w.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    B, C, H, W = x.shape
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    out = torch.zeros_like(x)
    for dy in range(3):
        for dx in range(3):
            out = out + xp[:, :, dy:dy + H, dx:dx + W] * w[:, 0, dy, dx].view(1, -1, 1, 1)
    return out + b.view(1, -1, 1, 1)


def layer_norm(x, g, b, eps=1e-5):
    """LayerNorm over the last dim (biased variance).  LGT.py:54-61"""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


# --------------------------------------------------------------------------------------
# data module  -- models/unlg_former.py:29-37, 58-61
# --------------------------------------------------------------------------------------
def op_D(P, z):
    t = resample(z, 0.5)
    t = dep_conv(t, P['D.1.weight'], P['D.1.bias'])
    t = resample(t, 0.5)
    return dep_conv(t, P['D.3.weight'], P['D.3.bias'])


def op_DT(P, r):
    t = resample(r, 2)
    t = dep_conv(t, P['DT.1.weight'], P['DT.1.bias'])
    t = resample(t, 2)
    return dep_conv(t, P['DT.3.weight'], P['DT.3.bias'])


def data_step(P, z, ms, pan, eta):
    """Z <- Z - eta * ( DT(D(Z) - ms) + RT(R(Z) - pan) )   unlg_former.py:58-61"""
    ms_term = op_DT(P, op_D(P, z) - ms)
    pan_term = point_conv(point_conv(z, P['R.weight'], P['R.bias']) - pan, P['RT.weight'], P['RT.bias'])
    return z - eta * (ms_term + pan_term)


# --------------------------------------------------------------------------------------
# LGT pieces  -- models/common/LGT.py
# --------------------------------------------------------------------------------------
def patch_embed(P, pre, x):
    """LGT.py:64-88 (patch_size=1): dw1x1 -> 1x1 C->E -> NHWC -> LayerNorm(E)."""
    t = dep_conv(x, P[pre + 'proj.0.weight'], P[pre + 'proj.0.bias'])
    t = point_conv(t, P[pre + 'proj.1.weight'], P[pre + 'proj.1.bias'])
    t = t.permute(0, 2, 3, 1)
    return layer_norm(t, P[pre + 'norm.weight'], P[pre + 'norm.bias'])


def local_mixer(P, pre, x, heads=2, win=8):
    """LGT.py:112-146.  x: [B,H,W,c] (already LayerNorm-ed first half of channels).
    Returns [B,H,W,c] (window merge of LGT.py:207-208 included)."""
    B, H, W, c = x.shape
    d = c // heads
    nh, nw = H // win, W // win
    # windows: token order row-major (i*win + j), window order row-major
    xw = x.reshape(B, nh, win, nw, win, c).permute(0, 1, 3, 2, 4, 5).reshape(B * nh * nw, win * win, c)
    wq = P[pre + 'to_qkv.weight'][:, :, 0, 0]       # [3c, c], chunk order q,k,v
    qkv = xw @ wq.t() + P[pre + 'to_qkv.bias']
    q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]

    def split(t):   # head h owns channels [h*d, (h+1)*d)
        return t.reshape(-1, win * win, heads, d).permute(0, 2, 1, 3)
    q, k, v = split(q) * (d ** -0.5), split(k), split(v)
    sim = q @ k.transpose(-1, -2) + P[pre + 'pos_emb']
    att = torch.softmax(sim, dim=-1)
    out = (att @ v).permute(0, 2, 1, 3).reshape(B, nh, nw, win, win, c)
    return out.permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, c)


# The four bins (ky in {0, H/2}) x (kx in {0, W/2}) of a real plane's spectrum are purely real; where their real part is
# negative, angle() returns +pi or -pi by the SIGN OF THE ZERO the FFT library happened to leave in the imaginary part.  For
# power-of-two sizes pocketfft leaves +0 on every host we ran; for other sizes (radix-3/5 passes) the sign depends on the host
# CPU's code path (measured: +0 in the build container, -0 for the ky = H/2 bins on the GPU box's host, PAN 48x48).  The
# reference's result is therefore machine-dependent there.  With this switch on, the oracle uses the +0 (= +pi) convention
# for those four bins everywhere -- what the reference computes wherever its zero is positive.  Default off = verbatim.
CANONICAL_REAL_BINS = False


def global_mixer(P, pre, x):
    """LGT.py:149-180.  x: [B,H,W,c] -> [B,H,W,c]; FFT amplitude/phase mixer."""
    B, H, W, c = x.shape
    xc = x.permute(0, 3, 1, 2)
    fre = torch.fft.rfft2(xc, norm='backward')
    amp = torch.abs(fre)
    if CANONICAL_REAL_BINS:
        keep = torch.ones(H, W // 2 + 1, dtype=fre.real.dtype)
        for ky in (0, H // 2):
            for kx in (0, W // 2):
                keep[ky, kx] = 0.0
        fre = torch.complex(fre.real, fre.imag * keep + 0.0)     # x * 0 + 0 = +0 for either sign of zero
    pha = torch.angle(fre)
    amp = dep_conv(amp, P[pre + 'conv_amp.0.weight'], P[pre + 'conv_amp.0.bias'])
    pha = dep_conv(pha, P[pre + 'conv_pha.0.weight'], P[pre + 'conv_pha.0.bias'])
    real = amp * torch.cos(pha) + 1e-8
    imag = amp * torch.sin(pha) + 1e-8
    out = torch.complex(real, imag) + 1e-8
    out = torch.abs(torch.fft.irfft2(out, s=(H, W), norm='backward'))
    return out.permute(0, 2, 3, 1)


def lg_mixer(P, pre, x, drop_mask=None):
    """LGT.py:183-219 on LayerNorm-ed x [B,H,W,E].  drop_mask: optional [B,E,H,W] keep-mask
    already divided by keep prob (eval / p=0: None)."""
    half = x.shape[-1] // 2
    x1 = local_mixer(P, pre + 'local_mixer.', x[..., :half])
    x2 = global_mixer(P, pre + 'global_mixer.', x[..., half:])
    out = torch.cat((x1, x2), dim=-1).permute(0, 3, 1, 2)
    out = point_conv(out, P[pre + 'proj.weight'], P[pre + 'proj.bias'])
    if drop_mask is not None:
        out = out * drop_mask
    return out.permute(0, 2, 3, 1)


def feed_forward(P, pre, x):
    """LGT.py:91-109 on LayerNorm-ed x [B,H,W,E]."""
    t = x.permute(0, 3, 1, 2)
    t = gelu(point_conv(t, P[pre + 'net.0.weight'], P[pre + 'net.0.bias']))
    t = point_conv(t, P[pre + 'net.2.point_conv.weight'], P[pre + 'net.2.point_conv.bias'])
    t = gelu(dep_conv(t, P[pre + 'net.2.depth_conv.weight'], P[pre + 'net.2.depth_conv.bias']))
    t = point_conv(t, P[pre + 'net.4.weight'], P[pre + 'net.4.bias'])
    return t.permute(0, 2, 3, 1)


def lgb(P, pre, x, nblocks):
    """LGT.py:222-248.  x [B,H,W,E] -> [B,H,W,E] (the reference's trailing NCHW permute is
    left to the caller)."""
    for j in range(nblocks):
        bp = f'{pre}blocks.{j}.'
        y = layer_norm(x, P[bp + '0.fn.norm.weight'], P[bp + '0.fn.norm.bias'])
        x = x + lg_mixer(P, bp + '0.fn.fn.', y)
        y = layer_norm(x, P[bp + '1.fn.norm.weight'], P[bp + '1.fn.norm.bias'])
        x = x + feed_forward(P, bp + '1.fn.fn.', y)
    return x


def lgt(P, pre, x, num_block=(2, 1)):
    """LGT.forward, LGT.py:314-344, scales=2.  x [B,C,H,W] -> [B,C,H,W]."""
    fea = patch_embed(P, pre + 'patch_embed.', x)
    fea = lgb(P, pre + 'encoder_layers.0.0.', fea, num_block[0])
    skip = fea.permute(0, 3, 1, 2)
    t = resample(skip, 0.5)
    t = point_conv(t, P[pre + 'encoder_layers.0.1.1.weight'], P[pre + 'encoder_layers.0.1.1.bias'])
    fea = lgb(P, pre + 'bottleneck.', t.permute(0, 2, 3, 1), num_block[1])
    t = resample(fea.permute(0, 3, 1, 2), 2)
    t = point_conv(t, P[pre + 'decoder_layers.0.0.1.weight'], P[pre + 'decoder_layers.0.0.1.bias'])
    t = point_conv(torch.cat([t, skip], dim=1), P[pre + 'decoder_layers.0.1.weight'],
                   P[pre + 'decoder_layers.0.1.bias'])
    fea = lgb(P, pre + 'decoder_layers.0.2.', t.permute(0, 2, 3, 1), num_block[0])
    out = point_conv(fea.permute(0, 3, 1, 2), P[pre + 'tail.1.weight'], P[pre + 'tail.1.bias'])
    return out + x


# --------------------------------------------------------------------------------------
# whole net  -- models/unlg_former.py:50-67
# --------------------------------------------------------------------------------------
def forward(P, ms, pan, stage, mode='live'):
    """Pansharpening.forward.  mode='faithful' executes every stage's LGT like the reference
    (results of stages 0..K-2 are discarded: unlg_former.py:63 never feeds Z_ back);
    mode='live' skips them.  Outputs are identical.
    mode='chained' is NOT the reference: the intended unfolding, where the next stage's data step consumes the
    LGT's output -- the same two functions composed the other way, so it is pinned only through them."""
    if mode not in ('faithful', 'live', 'chained'):
        raise ValueError(mode)
    z = resample(ms, 4)
    out = None
    for i in range(stage):
        z = data_step(P, z, ms, pan, P[f'eta.{i}'])
        if mode != 'live' or i == stage - 1:
            out = lgt(P, f'prior_module.{i}.', z)
        if mode == 'chained':
            z = out
    return out


def l1_loss(out, gt):
    """nn.L1Loss() mean  -- models/base/losses.py:19-40"""
    return (out - gt).abs().mean()


def live_param_names(P, stage):
    """Names that receive a gradient: shared data module, all eta, last LGT."""
    last = f'prior_module.{stage - 1}.'
    return [k for k in P if not k.startswith('prior_module.') or k.startswith(last)]


# --------------------------------------------------------------------------------------
# optimiser restatement: torch.optim.Adam (no amsgrad, no weight decay) + StepLR
# models/base/base_model.py:116-147, configs/unlg_former.py:82-86
# --------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def steplr(lr0, it, step_size, gamma):
    """lr used at (0-based) iteration `it` when StepLR.step() runs every iteration."""
    return lr0 * gamma ** (it // step_size)


# --------------------------------------------------------------------------------------
# IQA (float64 numpy in the reference: models/base/metrics.py:22-48,166-182)
# --------------------------------------------------------------------------------------
def psnr(img1, img2, dynamic_range=2047.5):
    import numpy as np
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    if mse <= 1e-10:
        return np.inf
    return 20 * np.log10(dynamic_range / (np.sqrt(mse) + np.finfo(np.float64).eps))


def sam(img1, img2):
    import numpy as np
    a = img1.astype(np.float64)
    b = img2.astype(np.float64)
    inner = (a * b).sum(axis=2)
    n1 = np.sqrt((a * a).sum(axis=2))
    n2 = np.sqrt((b * b).sum(axis=2))
    cos = (inner / (n1 * n2 + np.finfo(np.float64).eps)).clip(min=0, max=1)
    return np.mean(np.arccos(cos))


def ergas(img_fake, img_real, scale=4):
    import numpy as np
    a = img_fake.astype(np.float64)
    b = img_real.astype(np.float64)
    means_real = b.reshape(-1, b.shape[2]).mean(axis=0)
    mses = ((a - b) ** 2).reshape(-1, a.shape[2]).mean(axis=0)
    return 100 / scale * np.sqrt((mses / (means_real ** 2 + np.finfo(np.float64).eps)).mean())
