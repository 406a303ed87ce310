"""Reference-based image-quality indices of the evaluation loop, written from their published definitions.

    PSNR   10 log10(peak^2 / MSE)
    SAM    mean spectral angle between the pixel vectors of the two images            (Yuhas et al., 1992)
    ERGAS  100/ratio * sqrt(mean_band(MSE_band / mean_band^2))                         (Wald, 2000)
    SSIM   mean over windows of  l(x,y) * cs(x,y)  with Gaussian-weighted local moments   (Wang et al., 2004)
    Q      universal image quality index: SSIM with C1 = C2 = 0 on box windows          (Wang & Bovik, 2002)

Conventions the evaluation of reference `models/base/metrics.py` fixes, so that numbers are comparable with its tables
(`ref_evaluate`, metrics.py:409-417): arrays are (H, W) or (H, W, bands) in digital numbers, arithmetic in float64, peak
value 2047.5 (11-bit data); a multi-band SSIM / Q is the plain mean of the per-band values; SSIM uses an 11-tap Gaussian
(sigma 1.5) and Q an 8 x 8 box, and both average only over windows that lie fully inside the image; SAM is in radians with
the cosine clipped to [0, 1]; ERGAS uses ratio 4.

Parity status: PSNR / SAM / ERGAS are pinned by values the reference itself produced (tests/golden/net_*.npz `metrics`).
SSIM / Q call cv2.filter2D in the reference and cv2 is not installed in the build image, so they are PARITY-UNPINNED: they
are checked against brute-force evaluations of the definitions above only (tests/test_metrics_cpu.py).  Restricting the
average to fully covered windows makes the result independent of any border rule.
The no-reference family of the full-resolution pass (`no_ref_evaluate`: D_lambda, D_s, QNR; Alparone et al., 2008) is written from the
published definitions as well: Q on 32 x 32 windows, the PAN image brought to MS resolution by a Gaussian low-pass whose gain at the MS
Nyquist frequency is the sensor's MTF value, then decimation.  The reference builds that filter with a 2-D window method on top of cv2 /
scipy.ndimage and a per-satellite table; this is the separable Gaussian of the same MTF gain, so the family is PARITY-UNPINNED like SSIM / Q
and is checked against brute-force evaluations of its definitions only.
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

PEAK = 2047.5
_TINY = np.finfo(np.float64).eps
SSIM_TAPS, SSIM_SIGMA, Q_BLOCK, ERGAS_RATIO = 11, 1.5, 8, 4
QNR_BLOCK, MTF_TAPS, MTF_GAIN_PAN = 32, 41, 0.15


def _as_pair(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        raise ValueError(f'images differ in shape: {a.shape} vs {b.shape}')
    if a.ndim not in (2, 3):
        raise ValueError(f'expected an (H, W) or (H, W, bands) array, got {a.ndim} dimensions')
    return a.astype(np.float64), b.astype(np.float64)


def _band_mean(index, a, b, *args):
    """a per-band index averaged over the bands of (H, W, bands) inputs"""
    a, b = _as_pair(a, b)
    if a.ndim == 2:
        return float(index(a, b, *args))
    return float(np.mean([index(a[..., k], b[..., k], *args) for k in range(a.shape[2])]))


def gaussian_taps(n=SSIM_TAPS, sigma=SSIM_SIGMA):
    """n samples of a centred Gaussian, normalised to sum 1"""
    t = np.arange(n, dtype=np.float64) - 0.5 * (n - 1)
    g = np.exp(-0.5 * (t / sigma) ** 2)
    return g / g.sum()


def _inside_windows(img, taps):
    """separable weighted sum over every window position that lies fully inside `img`: (H, W) -> (H-n+1, W-n+1)"""
    rows = sliding_window_view(img, taps.size, axis=0) @ taps
    return sliding_window_view(rows, taps.size, axis=1) @ taps


def _local_moments(x, y, taps):
    """local means, variances and covariance under the window `taps` x `taps`"""
    mx, my = _inside_windows(x, taps), _inside_windows(y, taps)
    vx = _inside_windows(x * x, taps) - mx * mx
    vy = _inside_windows(y * y, taps) - my * my
    cxy = _inside_windows(x * y, taps) - mx * my
    return mx, my, vx, vy, cxy


def _ssim_band(x, y, peak):
    c1, c2 = (0.01 * peak) ** 2, (0.03 * peak) ** 2
    mx, my, vx, vy, cxy = _local_moments(x, y, gaussian_taps())
    luminance = (2.0 * mx * my + c1) / (mx * mx + my * my + c1)
    structure = (2.0 * cxy + c2) / (vx + vy + c2)
    return (luminance * structure).mean()


def _q_band(x, y, block):
    if block < 2:
        raise ValueError('the Q index needs windows of at least 2 x 2 pixels')
    mx, my, vx, vy, cxy = _local_moments(x, y, np.full(block, 1.0 / block))
    energy, spread = mx * mx + my * my, vx + vy
    # a factor whose denominator vanishes (flat window / zero-mean window) is taken as 1, the value of identical inputs
    luminance = np.divide(2.0 * mx * my, energy, out=np.ones_like(energy), where=energy > 1e-8)
    structure = np.divide(2.0 * cxy, spread, out=np.ones_like(spread), where=spread > 1e-8)
    return (luminance * structure).mean()


def psnr(img1, img2, dynamic_range=PEAK):
    a, b = _as_pair(img1, img2)
    mse = np.mean(np.square(a - b))
    if mse <= 1e-10:                      # identical up to rounding: reported as infinite, like the reference's table code
        return np.inf
    return float(20.0 * np.log10(dynamic_range / (np.sqrt(mse) + _TINY)))


def sam(img1, img2):
    a, b = _as_pair(img1, img2)
    if a.ndim != 3 or a.shape[2] < 2:
        raise ValueError('the spectral angle needs at least two bands: (H, W, bands)')
    dot = np.einsum('hwc,hwc->hw', a, b)
    norms = np.linalg.norm(a, axis=2) * np.linalg.norm(b, axis=2)
    return float(np.arccos(np.clip(dot / (norms + _TINY), 0.0, 1.0)).mean())


def ergas(img_fake, img_real, scale=ERGAS_RATIO):
    fake, real = _as_pair(img_fake, img_real)
    if fake.ndim == 2:
        fake, real = fake[..., None], real[..., None]
    band_mse = np.square(fake - real).mean(axis=(0, 1))
    band_mean = real.mean(axis=(0, 1))
    return float(100.0 / scale * np.sqrt(np.mean(band_mse / (np.square(band_mean) + _TINY))))


def ssim(img1, img2, dynamic_range=PEAK):
    return _band_mean(_ssim_band, img1, img2, dynamic_range)


def qindex(img1, img2, block_size=Q_BLOCK):
    return _band_mean(_q_band, img1, img2, block_size)


def ref_evaluate(pred, gt):
    """the five reference-based indices in the order of the reference's result tables: PSNR, SSIM, Q, SAM, ERGAS"""
    return [psnr(pred, gt), ssim(pred, gt), qindex(pred, gt), sam(pred, gt), ergas(pred, gt)]


# ---------------------------------------------------------------------------------------------------------------------
# no-reference indices of the full-resolution pass (reference models/base/metrics.py:290-408, `no_ref_evaluate`)
# ---------------------------------------------------------------------------------------------------------------------
def mtf_taps(gain_at_nyquist, ratio=ERGAS_RATIO, n=MTF_TAPS):
    """n samples of the Gaussian low-pass whose FREQUENCY response has the value `gain_at_nyquist` at the Nyquist frequency of the
    coarser grid (1 / (2 ratio) cycles per fine pixel): H(f) = exp(-2 pi^2 s^2 f^2) for a spatial Gaussian of standard deviation s"""
    if not 0.0 < gain_at_nyquist < 1.0:
        raise ValueError('the MTF gain at Nyquist lies in (0, 1)')
    f_nyq = 1.0 / (2.0 * ratio)
    sigma = np.sqrt(-np.log(gain_at_nyquist) / (2.0 * np.pi ** 2 * f_nyq ** 2))
    return gaussian_taps(n, sigma)


def mtf_degrade(img, gain_at_nyquist=MTF_GAIN_PAN, ratio=ERGAS_RATIO):
    """(H, W[, bands]) -> (H / ratio, W / ratio[, bands]): separable MTF-matched low-pass (edges replicated), then every ratio-th sample"""
    x = np.asarray(img, dtype=np.float64)
    taps = mtf_taps(gain_at_nyquist, ratio)
    half = taps.size // 2
    pad = [(half, half), (half, half)] + [(0, 0)] * (x.ndim - 2)
    xp = np.pad(x, pad, mode='edge')
    rows = np.tensordot(sliding_window_view(xp, taps.size, axis=0), taps, axes=([-1], [0]))
    full = np.tensordot(sliding_window_view(rows, taps.size, axis=1), taps, axes=([-1], [0]))
    return full[::ratio, ::ratio]


def d_lambda(fused, ms, block_size=QNR_BLOCK, p=1):
    """spectral distortion: how much the inter-band Q indices of the fused image differ from those of the MS image,
    ( mean over band pairs l < r of |Q(F_l, F_r) - Q(M_l, M_r)|^p )^(1/p)"""
    f, m = np.asarray(fused, np.float64), np.asarray(ms, np.float64)
    if f.ndim != 3 or m.ndim != 3 or f.shape[2] != m.shape[2] or f.shape[2] < 2:
        raise ValueError('fused and MS images are (H, W, bands) with the same number (>= 2) of bands')
    nb = f.shape[2]
    bs_m = min(block_size, min(m.shape[:2]))
    diffs = [abs(_q_band(f[..., l], f[..., r], block_size) - _q_band(m[..., l], m[..., r], bs_m)) ** p
             for l in range(nb) for r in range(l + 1, nb)]
    return float(np.mean(diffs) ** (1.0 / p))


def d_s(fused, ms, pan, pan_lr=None, block_size=QNR_BLOCK, q=1, ratio=ERGAS_RATIO, gain_at_nyquist=MTF_GAIN_PAN):
    """spatial distortion: ( mean over bands of |Q(F_l, P) - Q(M_l, P_lr)|^q )^(1/q); P_lr = the PAN image at MS resolution
    (MTF-matched low-pass + decimation unless given)"""
    f, m, pn = np.asarray(fused, np.float64), np.asarray(ms, np.float64), np.asarray(pan, np.float64).squeeze()
    if pn.ndim != 2 or f.shape[:2] != pn.shape:
        raise ValueError('PAN is (H, W) at the fused image\'s resolution')
    plr = mtf_degrade(pn, gain_at_nyquist, ratio) if pan_lr is None else np.asarray(pan_lr, np.float64).squeeze()
    if plr.shape != m.shape[:2]:
        raise ValueError(f'low-resolution PAN {plr.shape} does not match the MS image {m.shape[:2]}')
    bs_lr = min(block_size, min(plr.shape))        # the reference's tables use the SAME window at both resolutions (metrics.py:325,329)
    diffs = [abs(_q_band(f[..., l], pn, block_size) - _q_band(m[..., l], plr, bs_lr)) ** q for l in range(f.shape[2])]
    return float(np.mean(diffs) ** (1.0 / q))


def qnr(fused, ms, pan, pan_lr=None, alpha=1, beta=1, **kw):
    """quality with no reference: (1 - D_lambda)^alpha (1 - D_s)^beta, 1 for a fusion without spectral or spatial distortion"""
    return float((1.0 - d_lambda(fused, ms, kw.get('block_size', QNR_BLOCK))) ** alpha * (1.0 - d_s(fused, ms, pan, pan_lr, **kw)) ** beta)


def no_ref_evaluate(pred, pan, ms):
    """the three no-reference indices in the order of the reference's result tables: D_lambda, D_s, QNR"""
    dl, ds = d_lambda(pred, ms), d_s(pred, ms, pan)
    return [dl, ds, (1.0 - dl) * (1.0 - ds)]
