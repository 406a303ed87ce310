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


def test_no_reference_indices_follow_their_definitions():
    """D_lambda / D_s / QNR (Alparone et al. 2008; reference models/base/metrics.py:290-408 `no_ref_evaluate`), written from the
    definitions: checked against brute-force evaluations on small images (parity-unpinned like SSIM / Q: the reference's MTF filter
    needs cv2 / scipy.ndimage tables that are not in the image)"""
    from lgteun_amd import metrics as mtc
    rng = np.random.default_rng(7)
    H = 64
    fused = rng.uniform(100, 1800, (H, H, 4))
    pan = fused.mean(axis=2) + rng.normal(0, 20, (H, H))
    ms = mtc.mtf_degrade(fused, 0.3) + rng.normal(0, 5, (H // 4, H // 4, 4))

    def q_brute(x, y, bs):          # universal image quality index, mean over every fully covered window
        vals = []
        for i in range(x.shape[0] - bs + 1):
            for j in range(x.shape[1] - bs + 1):
                a, b = x[i:i + bs, j:j + bs].ravel(), y[i:i + bs, j:j + bs].ravel()
                ma, mb = a.mean(), b.mean()
                va, vb, cab = ((a - ma) ** 2).mean(), ((b - mb) ** 2).mean(), ((a - ma) * (b - mb)).mean()
                vals.append((2 * ma * mb / (ma * ma + mb * mb)) * (2 * cab / (va + vb)))
        return float(np.mean(vals))
    bs = 32
    pairs = [(l, r) for l in range(4) for r in range(l + 1, 4)]
    dl = np.mean([abs(q_brute(fused[..., l], fused[..., r], bs) - q_brute(ms[..., l], ms[..., r], 16)) for l, r in pairs])
    assert abs(mtc.d_lambda(fused, ms, bs) - dl) < 1e-10
    plr = mtc.mtf_degrade(pan)
    ds = np.mean([abs(q_brute(fused[..., l], pan, bs) - q_brute(ms[..., l], plr, 16)) for l in range(4)])
    assert abs(mtc.d_s(fused, ms, pan, block_size=bs) - ds) < 1e-10
    got = mtc.no_ref_evaluate(fused, pan, ms)
    assert abs(got[0] - dl) < 1e-10 and abs(got[1] - ds) < 1e-10 and abs(got[2] - (1 - dl) * (1 - ds)) < 1e-12
    assert abs(mtc.qnr(fused, ms, pan) - got[2]) < 1e-12
    # a fusion whose inter-band and band-to-PAN similarities are those of the inputs has no distortion: QNR = 1
    same = np.repeat(np.repeat(ms, 4, axis=0), 4, axis=1)
    assert mtc.d_lambda(same, same, 8) == 0.0
    # the MTF low-pass: unit DC gain, the prescribed gain at the coarse grid's Nyquist frequency, decimation by the ratio
    taps = mtc.mtf_taps(0.15, 4)
    f = 1.0 / 8.0
    gain = abs(np.sum(taps * np.exp(-2j * np.pi * f * (np.arange(taps.size) - taps.size // 2))))
    assert abs(taps.sum() - 1) < 1e-12 and abs(gain - 0.15) < 2e-3
    assert mtc.mtf_degrade(np.full((32, 48), 5.0)).shape == (8, 12) and np.allclose(mtc.mtf_degrade(np.full((32, 48), 5.0)), 5.0)
    with pytest.raises(ValueError):
        mtc.d_s(fused, ms, pan[:-4])
