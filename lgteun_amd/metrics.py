"""PSNR / SAM / ERGAS in float64 numpy on denormalised HWC arrays -- mirror of reference
models/base/metrics.py:22-48,166-182 (the cv2-based SSIM/Q indices are out of scope)."""
import numpy as np

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


def ref_evaluate(pred, gt):
    """subset of reference ref_evaluate (metrics.py:409-417): PSNR, SAM, ERGAS"""
    return [psnr(pred, gt), sam(pred, gt), ergas(pred, gt)]
