"""Image-quality metrics in float64 numpy on denormalised HWC arrays -- mirror of reference models/base/metrics.py:
reference-based PSNR / SSIM / Q / SAM / ERGAS / SCC (:22-182, ref_evaluate :409-417) and the no-reference D_lambda / D_s / QNR
used by the full-resolution test (:258-327, 389-406).

The reference computes the windowed statistics with cv2.filter2D and then crops to the valid region; a correlation cropped to
its valid region does not depend on the border rule, so here it is a plain `valid` correlation (scipy), no cv2.  cv2.resize
(..., INTER_NEAREST) by an integer factor is a strided slice.  Pinned by tests/test_metrics_cpu.py against brute-force
definitions (cv2 is absent in the build container, so these four cannot be pinned against the reference itself: PSNR, SAM and
ERGAS are, through the goldens)."""
import numpy as np
from scipy import ndimage, signal

dynamic_r = 2047.5


def sam(img1, img2):
    if not img1.shape == img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    assert img1.ndim == 3 and img1.shape[2] > 1, 'image n_channels should be greater than 1'
    a = img1.astype(np.float64)
    b = img2.astype(np.float64)
    inner = (a * b).sum(axis=2)
    na = np.sqrt((a ** 2).sum(axis=2))
    nb = np.sqrt((b ** 2).sum(axis=2))
    cos_theta = (inner / (na * nb + np.finfo(np.float64).eps)).clip(min=0, max=1)
    return np.mean(np.arccos(cos_theta))


def psnr(img1, img2, dynamic_range=dynamic_r):
    if not img1.shape == img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    if mse <= 1e-10:
        return np.inf
    return 20 * np.log10(dynamic_range / (np.sqrt(mse) + np.finfo(np.float64).eps))


def ergas(img_fake, img_real, scale=4):
    if not img_fake.shape == img_real.shape:
        raise ValueError('Input images must have the same dimensions.')
    a = img_fake.astype(np.float64)
    b = img_real.astype(np.float64)
    if a.ndim == 2:
        return 100 / scale * np.sqrt(np.mean((a - b) ** 2) / (b.mean() ** 2 + np.finfo(np.float64).eps))
    means_real = b.reshape(-1, b.shape[2]).mean(axis=0)
    mses = ((a - b) ** 2).reshape(-1, a.shape[2]).mean(axis=0)
    return 100 / scale * np.sqrt((mses / (means_real ** 2 + np.finfo(np.float64).eps)).mean())


def scc(img1, img2):
    """mean per-band correlation coefficient (metrics.py:58-74)"""
    if not img1.shape == img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    a = img1.astype(np.float64)
    b = img2.astype(np.float64)
    if a.ndim == 2:
        return np.corrcoef(a.reshape(1, -1), b.reshape(1, -1))[0, 1]
    if a.ndim == 3:
        return np.mean([np.corrcoef(a[..., i].reshape(1, -1), b[..., i].reshape(1, -1))[0, 1] for i in range(a.shape[2])])
    raise ValueError('Wrong input image dimensions.')


def _valid_corr(img, window):
    """cv2.filter2D(img, -1, window) cropped to the region where the window lies inside the image"""
    return signal.correlate2d(img, window, mode='valid')


def _qindex(img1, img2, block_size=8):
    """universal image quality index of one band over sliding block_size windows (metrics.py:77-113)"""
    assert block_size > 1, 'block_size shold be greater than 1!'
    a = img1.astype(np.float64)
    b = img2.astype(np.float64)
    window = np.ones((block_size, block_size)) / (block_size ** 2)
    mu1 = _valid_corr(a, window)
    mu2 = _valid_corr(b, window)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    sigma1_sq = _valid_corr(a ** 2, window) - mu1_sq
    sigma2_sq = _valid_corr(b ** 2, window) - mu2_sq
    sigma12 = _valid_corr(a * b, window) - mu1_mu2
    q = np.ones(sigma12.shape)
    idx = ((sigma1_sq + sigma2_sq) < 1e-8) * ((mu1_sq + mu2_sq) > 1e-8)
    q[idx] = 2 * mu1_mu2[idx] / (mu1_sq + mu2_sq)[idx]
    idx = ((sigma1_sq + sigma2_sq) > 1e-8) * ((mu1_sq + mu2_sq) < 1e-8)
    q[idx] = 2 * sigma12[idx] / (sigma1_sq + sigma2_sq)[idx]
    idx = ((sigma1_sq + sigma2_sq) > 1e-8) * ((mu1_sq + mu2_sq) > 1e-8)
    q[idx] = ((2 * mu1_mu2[idx]) * (2 * sigma12[idx])) / ((mu1_sq + mu2_sq)[idx] * (sigma1_sq + sigma2_sq)[idx])
    return np.mean(q)


def qindex(img1, img2, block_size=8):
    if not img1.shape == img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    if img1.ndim == 2:
        return _qindex(img1, img2, block_size)
    if img1.ndim == 3:
        return np.array([_qindex(img1[..., i], img2[..., i], block_size) for i in range(img1.shape[2])]).mean()
    raise ValueError('Wrong input image dimensions.')


def gaussian_kernel1d(ksize=11, sigma=1.5):
    """cv2.getGaussianKernel(ksize, sigma) for sigma > 0: exp(-(i - (ksize-1)/2)^2 / (2 sigma^2)), normalised to sum 1"""
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return k / k.sum()


def _ssim(img1, img2, dynamic_range=dynamic_r):
    """SSIM of one band, 11x11 Gaussian window sigma 1.5, valid region (metrics.py:129-150)"""
    C1 = (0.01 * dynamic_range) ** 2
    C2 = (0.03 * dynamic_range) ** 2
    a = img1.astype(np.float64)
    b = img2.astype(np.float64)
    k = gaussian_kernel1d(11, 1.5)
    window = np.outer(k, k)
    mu1 = _valid_corr(a, window)
    mu2 = _valid_corr(b, window)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    sigma1_sq = _valid_corr(a ** 2, window) - mu1_sq
    sigma2_sq = _valid_corr(b ** 2, window) - mu2_sq
    sigma12 = _valid_corr(a * b, window) - mu1_mu2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean()


def ssim(img1, img2, dynamic_range=dynamic_r):
    if not img1.shape == img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    if img1.ndim == 2:
        return _ssim(img1, img2, dynamic_range)
    if img1.ndim == 3:
        return np.array([_ssim(img1[..., i], img2[..., i], dynamic_range) for i in range(img1.shape[2])]).mean()
    raise ValueError('Wrong input image dimensions.')


# ---- observation model of the no-reference indices (metrics.py:190-258) ----
def gaussian2d(N, std):
    t = np.arange(-(N - 1) // 2, (N + 2) // 2)
    t1, t2 = np.meshgrid(t, t)
    std = np.double(std)
    return np.exp(-0.5 * (t1 / std) ** 2) * np.exp(-0.5 * (t2 / std) ** 2)


def kaiser2d(N, beta):
    t = np.arange(-(N - 1) // 2, (N + 2) // 2) / np.double(N - 1)
    t1, t2 = np.meshgrid(t, t)
    t12 = np.sqrt(t1 * t1 + t2 * t2)
    w = np.interp(t12, t, np.kaiser(N, beta))
    w[t12 > t[-1]] = 0
    w[t12 < t[0]] = 0
    return w


def fir_filter_wind(Hd, w):
    hd = np.rot90(np.fft.fftshift(np.rot90(Hd, 2)), 2)
    h = np.fft.fftshift(np.fft.ifft2(hd))
    h = np.rot90(h, 2) * w
    return h / np.sum(h)


def GNyq2win(GNyq, scale=4, N=41):
    """2-D low-pass window whose gain at the Nyquist frequency of the MS grid is GNyq"""
    fcut = 1 / scale
    alpha = np.sqrt(((N - 1) * (fcut / 2)) ** 2 / (-2 * np.log(GNyq)))
    H = gaussian2d(N, alpha)
    return np.real(fir_filter_wind(H / np.max(H), kaiser2d(N, 0.5)))


def mtf_resize(img, satellite='QuickBird', scale=4):
    """MTF-matched low-pass + nearest decimation by `scale` (metrics.py:229-258)"""
    scale = int(scale)
    if satellite == 'QuickBird':
        GNyq, GNyqPan = [0.34, 0.32, 0.30, 0.22], 0.15
    elif satellite == 'IKONOS':
        GNyq, GNyqPan = [0.26, 0.28, 0.29, 0.28], 0.17
    else:
        raise NotImplementedError('satellite: QuickBird or IKONOS')
    x = img.squeeze().astype(np.float64)
    if x.ndim == 2:
        lowpass = GNyq2win(GNyqPan, scale, N=41)
    else:
        lowpass = np.stack([GNyq2win(g, scale, N=41) for g in GNyq], axis=-1)
    x = ndimage.correlate(x, lowpass, mode='nearest')
    H, W = x.shape[:2]
    # cv2.resize(..., INTER_NEAREST) to (H//scale, W//scale): source index = floor(dst * scale)
    return x[:(H // scale) * scale:scale, :(W // scale) * scale:scale]


def D_lambda(img_fake, img_lm, block_size=32, p=1):
    """spectral distortion between the fused image and the LR MS (metrics.py:265-290)"""
    assert img_fake.ndim == img_lm.ndim == 3, 'Images must be 3D!'
    C_f, C_r = img_fake.shape[2], img_lm.shape[2]
    assert C_f == C_r, 'Fake and lm should have the same number of bands!'
    q_fake, q_lm = [], []
    for i in range(C_f):
        for j in range(i + 1, C_f):
            q_fake.append(_qindex(img_fake[..., i], img_fake[..., j], block_size=block_size))
            q_lm.append(_qindex(img_lm[..., i], img_lm[..., j], block_size=block_size))
    d = (np.abs(np.array(q_fake) - np.array(q_lm)) ** p).mean()
    return d ** (1 / p)


def D_s(img_fake, img_lm, pan, satellite='QuickBird', scale=4, block_size=32, q=1):
    """spatial distortion against the PAN and its MTF-degraded copy (metrics.py:293-327)"""
    assert img_fake.ndim == img_lm.ndim == 3, 'MS images must be 3D!'
    H_f, W_f, C_f = img_fake.shape
    H_r, W_r, C_r = img_lm.shape
    assert H_f // H_r == W_f // W_r == scale, 'Spatial resolution should be compatible with scale'
    assert C_f == C_r, 'Fake and lm should have the same number of bands!'
    assert pan.ndim == 3 and pan.shape[2] == 1, 'Panchromatic image must be 3D with one band'
    assert H_f == pan.shape[0] and W_f == pan.shape[1], "Pan's and fake's spatial resolution should be the same"
    pan_lr = mtf_resize(pan, satellite=satellite, scale=scale)
    q_hr = [_qindex(img_fake[..., i], pan[..., 0], block_size=block_size) for i in range(C_f)]
    q_lr = [_qindex(img_lm[..., i], pan_lr, block_size=block_size) for i in range(C_f)]
    d = (np.abs(np.array(q_hr) - np.array(q_lr)) ** q).mean()
    return d ** (1 / q)


def qnr(img_fake, img_lm, pan, satellite='QuickBird', scale=4, block_size=32, p=1, q=1, alpha=1, beta=1):
    return (1 - D_lambda(img_fake, img_lm, block_size, p)) ** alpha * (1 - D_s(img_fake, img_lm, pan, satellite, scale, block_size, q)) ** beta


def ref_evaluate(pred, gt):
    """reference-based metrics in the reference's order (metrics.py:409-417): PSNR, SSIM, Q, SAM, ERGAS"""
    return [psnr(pred, gt), ssim(pred, gt), qindex(pred, gt), sam(pred, gt), ergas(pred, gt)]


def no_ref_evaluate(pred, pan, hs):
    """no-reference metrics (metrics.py:420-425): D_lambda, D_s, QNR; pan is 2-D (H, W)"""
    pan3 = np.expand_dims(pan, -1)
    return [D_lambda(pred, hs), D_s(pred, hs, pan3), qnr(pred, hs, pan3)]
