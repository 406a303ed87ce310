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
