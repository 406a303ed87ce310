"""CPU: the reference-based image-quality indices (lgteun_amd/metrics.py) against brute-force evaluations of their
definitions.  cv2 is absent here, so SSIM / Q cannot be pinned against the reference code itself (parity unpinned);
PSNR / SAM / ERGAS are pinned by the reference's own values (tests/golden, test_oracle_golden.py, test_boundary_cpu.py)."""
import numpy as np
import pytest

from lgteun_amd import metrics as mtc


def _img(h, w, c, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 900 + 500 * np.sin(yy / 7.0)[..., None] * np.cos(xx / 5.0)[..., None]
    return np.clip(base + rng.normal(0, 60, (h, w, c)) + 80 * np.arange(c), 0, 2047).astype(np.float64)


def _window_stats_bruteforce(a, b, window):
    kh, kw = window.shape
    H, W = a.shape
    out = np.zeros((5, H - kh + 1, W - kw + 1))
    for y in range(H - kh + 1):
        for x in range(W - kw + 1):
            pa, pb = a[y:y + kh, x:x + kw], b[y:y + kh, x:x + kw]
            out[:, y, x] = [(window * pa).sum(), (window * pb).sum(), (window * pa * pa).sum(), (window * pb * pb).sum(),
                            (window * pa * pb).sum()]
    return out


def test_ssim_matches_bruteforce_definition():
    a, b = _img(24, 26, 1, 0)[..., 0], _img(24, 26, 1, 1)[..., 0]
    k = mtc.gaussian_taps(11, 1.5)
    assert abs(k.sum() - 1) < 1e-15 and np.allclose(k, k[::-1]) and k.argmax() == 5
    mu1, mu2, e11, e22, e12 = _window_stats_bruteforce(a, b, np.outer(k, k))
    C1, C2 = (0.01 * 2047.5) ** 2, (0.03 * 2047.5) ** 2
    want = (((2 * mu1 * mu2 + C1) * (2 * (e12 - mu1 * mu2) + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (e11 - mu1 ** 2 + e22 - mu2 ** 2 + C2))).mean()
    assert abs(mtc.ssim(a, b) - want) < 1e-12
    assert abs(mtc.ssim(a, a) - 1.0) < 1e-12
    x3, y3 = _img(20, 20, 4, 2), _img(20, 20, 4, 3)
    assert abs(mtc.ssim(x3, y3) - np.mean([mtc.ssim(x3[..., i], y3[..., i]) for i in range(4)])) < 1e-15


@pytest.mark.parametrize('bs', [8, 5])
def test_qindex_matches_bruteforce_definition(bs):
    a, b = _img(20, 22, 1, 4)[..., 0], _img(20, 22, 1, 5)[..., 0]
    mu1, mu2, e11, e22, e12 = _window_stats_bruteforce(a, b, np.ones((bs, bs)) / bs ** 2)
    s1, s2, s12 = e11 - mu1 ** 2, e22 - mu2 ** 2, e12 - mu1 * mu2
    want = (4 * mu1 * mu2 * s12 / ((mu1 ** 2 + mu2 ** 2) * (s1 + s2))).mean()   # generic branch (all windows textured, non-zero mean)
    assert abs(mtc.qindex(a, b, bs) - want) < 1e-10
    assert abs(mtc.qindex(a, a, bs) - 1.0) < 1e-10
    flat = np.full((16, 16), 7.0)
    assert mtc.qindex(flat, flat) == 1.0                     # sigma = 0, mean != 0 branch: 2 mu1 mu2 / (mu1^2 + mu2^2)
    assert mtc.qindex(np.zeros((16, 16)), np.zeros((16, 16))) == 1.0   # untouched default of the map


def test_psnr_sam_ergas_properties():
    a, b = _img(16, 16, 4, 6), _img(16, 16, 4, 7)
    mse = ((a - b) ** 2).mean()
    assert abs(mtc.psnr(a, b) - 10 * np.log10(2047.5 ** 2 / mse)) < 1e-9
    ang = [np.arccos(min(1.0, a[y, x] @ b[y, x] / np.sqrt((a[y, x] @ a[y, x]) * (b[y, x] @ b[y, x])))) for y in range(16) for x in range(16)]
    assert abs(mtc.sam(a, b) - np.mean(ang)) < 1e-9
    want = 25.0 * np.sqrt(np.mean([((a[..., k] - b[..., k]) ** 2).mean() / b[..., k].mean() ** 2 for k in range(4)]))
    assert abs(mtc.ergas(a, b) - want) < 1e-9
    assert abs(mtc.ergas(a[..., 0], b[..., 0]) - 25.0 * np.sqrt(((a[..., 0] - b[..., 0]) ** 2).mean() / b[..., 0].mean() ** 2)) < 1e-9
    assert mtc.psnr(a, a) == np.inf and mtc.sam(a, 3 * a) < 1e-7 and mtc.ergas(a, a) == 0.0
    r = mtc.ref_evaluate(a, b)
    assert len(r) == 5 and r[0] == mtc.psnr(a, b) and r[1] == mtc.ssim(a, b) and r[2] == mtc.qindex(a, b) and r[3] == mtc.sam(a, b) \
        and r[4] == mtc.ergas(a, b)
    with pytest.raises(ValueError):
        mtc.ssim(a, b[:-1])
    with pytest.raises(ValueError):
        mtc.sam(a[..., 0], b[..., 0])          # a spectral angle needs bands
    with pytest.raises(ValueError):
        mtc.qindex(a, b, 1)
    assert not hasattr(mtc, 'qnr')               # the no-reference family is out of scope
