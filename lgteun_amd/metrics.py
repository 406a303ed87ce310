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
The no-reference indices (D_lambda, D_s, QNR) are outside the scope of this build.
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

PEAK = 2047.5
_TINY = np.finfo(np.float64).eps
SSIM_TAPS, SSIM_SIGMA, Q_BLOCK, ERGAS_RATIO = 11, 1.5, 8, 4


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
