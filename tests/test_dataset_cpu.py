"""CPU: the input pipeline (lgteun_amd/dataset.py, mirror of reference dataset/*.py): TIFF codec round trips, the PSDataset
triplet contract, normalisation, augmentation semantics, per-rank sharding and the prefetching loader's pass-through."""
import numpy as np
import pytest
import torch

from lgteun_amd import dataset as ds


@pytest.mark.parametrize('kw', [dict(), dict(compress=True), dict(big_endian=True), dict(rows_per_strip=5), dict(compress=True, rows_per_strip=7)])
@pytest.mark.parametrize('shape,dtype', [((33, 20, 4), np.uint16), ((16, 16), np.uint16), ((9, 11, 8), np.uint16), ((12, 10, 3), np.uint8),
                                         ((8, 8, 4), np.float32), ((8, 8), np.int16)])
def test_tiff_roundtrip(tmp_path, kw, shape, dtype):
    rng = np.random.default_rng(0)
    a = (rng.random(shape) * 2047).astype(dtype)
    p = str(tmp_path / 'x.tif')
    ds.write_tiff(p, a, **kw)
    b = ds.read_tiff(p)
    assert b.shape == a.shape and b.dtype == a.dtype and np.array_equal(a, b)
    assert ds.load_image(p).dtype == np.float64


def test_tiff_rejects_what_it_cannot_read(tmp_path):
    p = tmp_path / 'bad.tif'
    p.write_bytes(b'not a tiff at all')
    with pytest.raises(ValueError):
        ds.read_tiff(str(p))
    with pytest.raises(ValueError):
        ds.write_tiff(str(tmp_path / 'y.tif'), np.zeros((4, 4), dtype=np.float64))


def _make_set(root, n, C=4, h=8, with_target=True, seed=0):
    rng = np.random.default_rng(seed)
    root.mkdir(parents=True, exist_ok=True)
    truth = {}
    for i in range(n):
        lr = rng.integers(0, 2048, (h, h, C)).astype(np.uint16)
        pan = rng.integers(0, 2048, (4 * h, 4 * h)).astype(np.uint16)
        mul = rng.integers(0, 2048, (4 * h, 4 * h, C)).astype(np.uint16)
        ds.write_tiff(str(root / f'{i}_lr.tif'), lr)
        ds.write_tiff(str(root / f'{i}_pan.tif'), pan)
        if with_target:
            ds.save_image(str(root / f'{i}_mul.tif'), mul.transpose(2, 0, 1))
        truth[str(i)] = (lr, pan, mul)
    return truth


def test_psdataset_contract(tmp_path):
    truth = _make_set(tmp_path / 'train', 5)
    d = ds.build_dataset(dict(type='PSDataset', image_dirs=[str(tmp_path / 'train')], bit_depth=11, norm_input=True))
    assert len(d) == 5 and d.image_ids == sorted(truth)
    it = d[2]
    lr, pan, mul = truth[it['image_id']]
    assert set(it) == {'input_lr', 'input_pan', 'target', 'input_pan_l', 'image_id'}
    assert it['input_lr'].shape == (4, 8, 8) and it['input_pan'].shape == (1, 32, 32) and it['target'].shape == (4, 32, 32)
    assert it['input_pan_l'].shape == (1, 8, 8) and it['input_lr'].dtype == torch.float32
    assert torch.allclose(it['input_lr'], torch.from_numpy(lr.transpose(2, 0, 1).astype(np.float32)) / 2047.5)
    assert torch.allclose(it['input_pan'][0], torch.from_numpy(pan.astype(np.float32)) / 2047.5)
    assert torch.allclose(it['target'], torch.from_numpy(mul.transpose(2, 0, 1).astype(np.float32)) / 2047.5)
    raw = ds.PSDataset([str(tmp_path / 'train')], 11, norm_input=False)[2]
    assert float(raw['input_pan'].max()) > 1.0
    back = ds.data_denormalize(it['input_pan'], 11)
    assert torch.allclose(back, raw['input_pan'], atol=1e-3)
    # two directories -> no target (reference ps_dataset.py:52), and no _mul files -> no target
    _make_set(tmp_path / 'full', 2, with_target=False, seed=1)
    assert 'target' not in ds.PSDataset([str(tmp_path / 'full')], 11)[0]
    assert 'target' not in ds.PSDataset([str(tmp_path / 'train'), str(tmp_path / 'full')], 11)[0]
    with pytest.raises(KeyError):
        ds.build_dataset(dict(type='NoSuchDataset'))


def test_pyr_down_is_binomial_blur_and_decimation():
    x = np.arange(64, dtype=np.float64).reshape(8, 8)
    y = ds.pyr_down(x)
    assert y.shape == (4, 4)
    # a linear ramp is reproduced away from the borders (the kernel is symmetric and sums to 1)
    assert abs(y[1, 1] - x[2, 2]) < 1e-12 and abs(y[2, 1] - x[4, 2]) < 1e-12
    assert np.allclose(ds.pyr_down(np.full((6, 6), 3.0)), 3.0)


def test_augmentation_reference_semantics():
    lr = torch.arange(2 * 4 * 8 * 8, dtype=torch.float32).reshape(2, 4, 8, 8)
    pan = torch.arange(2 * 1 * 32 * 32, dtype=torch.float32).reshape(2, 1, 32, 32)
    batch = dict(input_lr=lr, input_pan=pan, image_id=['a', 'b'])
    assert ds.data_augmentation(batch, None) is batch
    assert ds.data_augmentation(batch, dict(ud_flip=0.0, lr_flip=0.0)) is batch
    out = ds.data_augmentation(batch, dict(ud_flip=1.0), rng=lambda: 0.5)
    assert torch.equal(out['input_lr'], lr.flip(2)) and torch.equal(out['input_pan'], pan.flip(2)) and out['image_id'] == ['a', 'b']
    # both flips drawn: each is applied to the ORIGINAL image and the later one wins (reference dataset/utils.py:216-219)
    out = ds.data_augmentation(batch, dict(ud_flip=1.0, lr_flip=1.0), rng=lambda: 0.5)
    assert torch.equal(out['input_lr'], lr.flip(3))
    aug = dict(r4_crop=1.0)
    out = ds.data_augmentation(batch, aug, rng=lambda: 0.5)
    assert aug['r4_crop'] is True and out['input_lr'].shape == lr.shape and out['input_pan'].shape == pan.shape
    # crop start d = int(8 // 4 * 0.5) = 1 on the LR grid and 4 on the PAN grid; align_corners=True keeps the corner pixels
    assert out['input_lr'][0, 0, 0, 0] == lr[0, 0, 1, 1] and out['input_pan'][0, 0, 0, 0] == pan[0, 0, 4, 4]


@pytest.mark.parametrize('n,world,drop', [(10, 2, False), (10, 4, False), (10, 4, True), (7, 8, False), (64, 8, False)])
def test_sharded_sampler_partitions_every_epoch(n, world, drop):
    shards = []
    for r in range(world):
        s = ds.ShardedSampler(n, r, world, shuffle=True, seed=3, drop_last=drop)
        s.set_epoch(5)
        shards.append(list(s))
        assert len(shards[-1]) == len(s)
    assert len({len(x) for x in shards}) == 1                     # equal work per rank
    flat = [i for x in shards for i in x]
    if drop:
        assert len(set(flat)) == len(flat) == (n // world) * world
    else:
        assert set(flat) == set(range(n))                          # every sample visited (a few twice when n % world != 0)
        assert len(flat) == ((n + world - 1) // world) * world
    s0 = ds.ShardedSampler(n, 0, world, shuffle=True, seed=3, drop_last=drop)
    s0.set_epoch(6)
    if n > world:
        assert list(s0) != shards[0]                               # reshuffled per epoch
    assert list(ds.ShardedSampler(n, 0, 1, shuffle=False)) == list(range(n))
    with pytest.raises(ValueError):
        ds.ShardedSampler(4, 2, 2)


def test_build_loader_and_prefetch_passthrough(tmp_path):
    _make_set(tmp_path / 't', 6)
    cfg = dict(dataset=dict(type='PSDataset', image_dirs=[str(tmp_path / 't')], bit_depth=11, norm_input=True), batch_size=2, num_workers=0,
               shuffle=False)
    plain, smp = ds.build_loader(cfg)
    assert smp is None
    ref = [b for b in plain]
    assert len(ref) == 3 and ref[0]['input_lr'].shape == (2, 4, 8, 8) and len(ref[0]['image_id']) == 2
    pre, _ = ds.build_loader(cfg, device='cpu')
    got = [b for b in pre]
    assert len(pre) == 3 and all(torch.equal(a['input_pan'], b['input_pan']) and a['image_id'] == b['image_id'] for a, b in zip(ref, got))
    # two ranks: disjoint halves, same number of steps
    l0, s0 = ds.build_loader(dict(cfg, shuffle=True), rank=0, world=2, seed=1)
    l1, s1 = ds.build_loader(dict(cfg, shuffle=True), rank=1, world=2, seed=1)
    ids0 = [i for b in l0 for i in b['image_id']]
    ids1 = [i for b in l1 for i in b['image_id']]
    assert len(ids0) == len(ids1) == 3 and set(ids0) | set(ids1) == {str(i) for i in range(6)} and not set(ids0) & set(ids1)


# ---- the reference's own dataset/utils.py functions (:155-263), captured by tools/gen_goldens.py --only-r5 -------------------------
# (the augmentation cases and their random() draws live in ONE place: the generator)
def _dataset_cases():
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))          # gen_goldens imports its sibling _ref_import at module level (no reference import happens)
    try:
        spec = importlib.util.spec_from_file_location('lg_gen_goldens', os.path.join(root, 'tools', 'gen_goldens.py'))
        gg = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gg)
    finally:
        sys.path.pop(0)
    return gg.DATASET_CASES, gg.dataset_batch()


def test_normalise_denormalise_and_augmentation_equal_the_reference():
    """data_normalize / data_denormalize (bit depths 11 and 10) and data_augmentation under eight fixed draw sequences -- no
    augmentation dict, nothing selected, ud flip, lr flip, BOTH flips (the later one wins, dataset/utils.py:216-219), the r4 and r2
    crop + bicubic align_corners=True resize with drawn offsets, and all four at once -- against the arrays the reference's own
    functions produced (tests/golden/dataset_utils.npz).  This pins the pure-torch part of SURVEY 8(f)-3; TIFF I/O stays unpinned
    (tifffile / gdal are absent here and the reference ships no image)."""
    from conftest import load_gold
    from lgteun_amd.base_model import data_denormalize, data_normalize
    g = load_gold('dataset_utils')
    cases, base = _dataset_cases()
    for name, probs, draws in cases:
        seq = iter(draws)
        batch = dict({k: torch.from_numpy(v.copy()) for k, v in base.items()}, image_id=['a', 'b'])
        res = ds.data_augmentation(batch, None if probs is None else dict(probs), rng=lambda: next(seq))
        assert next(seq, None) is None, name                   # the same number of draws as the reference
        assert res['image_id'] == ['a', 'b']
        for k in base:
            want = g[f'aug_{name}_{k}']
            assert tuple(res[k].shape) == want.shape, (name, k)
            assert np.allclose(res[k].numpy(), want, rtol=0, atol=1e-3), (name, k, float(np.abs(res[k].numpy() - want).max()))   # digital numbers up to 2047
    for bits in (11, 10):
        batch = dict({k: torch.from_numpy(v.copy()) for k, v in base.items()}, image_id=['a', 'b'])
        nrm = data_normalize(batch, bits)
        assert nrm['image_id'] == ['a', 'b']
        for k in base:
            assert np.array_equal(nrm[k].numpy(), g[f'norm{bits}_{k}']), (bits, k)
        assert np.array_equal(data_denormalize(nrm['target'], bits).numpy(), g[f'denorm{bits}_target'])
    # the cases differ from one another (a generator that ignored its draws would make this fixture vacuous)
    assert not np.array_equal(g['aug_ud_target'], g['aug_lr_target']) and np.array_equal(g['aug_ud_and_lr_target'], g['aug_lr_target'])
    assert not np.array_equal(g['aug_r4_target'], g['aug_r2_target']) and not np.array_equal(g['aug_all_target'], g['aug_ud_and_lr_target'])
