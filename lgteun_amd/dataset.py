"""Input pipeline of the hot path's caller -- mirror of reference dataset/{builder,ps_dataset,utils}.py restricted to what
feeds `UnlgFormer.train_iter` / `test`:

  DATASETS / build_dataset      dataset/builder.py:12-26
  PSDataset                     dataset/ps_dataset.py:20-69   (<id>_lr.tif, <id>_pan.tif, optional <id>_mul.tif triplets)
  load_image / save_image       dataset/utils.py:29-91        (tifffile / gdal there; a dependency-free baseline-TIFF codec here)
  data_normalize / denormalize  dataset/utils.py:232-263
  data_augmentation             dataset/utils.py:155-229      (flips and crop-resize, including the reference's "last selected
                                                               transform wins" behaviour)

and what the reference does not have but one process per GPU at thousands of pairs per second needs:

  ShardedSampler                equal, disjoint per-rank shards of every epoch's permutation (DistributedSampler semantics)
  PrefetchLoader                pinned-memory batches staged to the device on a side stream `depth` batches ahead, instead of the
                                synchronous `set_batch_cuda` (dataset/utils.py:98-110); a pair is only 336 KB fp32
  build_loader                  main.py:71-86 (DataLoader from a *_set_cfg dict) + the two above
"""
import os
import struct
import zlib

import numpy as np
import torch
import torch.utils.data as data
from scipy import ndimage

from .base_model import data_denormalize, data_normalize  # noqa: F401  (re-exported: same functions the runner uses)
from .compat import Registry

DATASETS = Registry('dataset')


def build_dataset(cfg, *args, **kwargs):
    cfg_ = dict(cfg)
    dataset_type = cfg_.pop('type')
    if dataset_type not in DATASETS:
        raise KeyError(f'Unrecognized task type {dataset_type}')
    return DATASETS.get(dataset_type)(*args, **kwargs, **cfg_)


# ------------------------------------------------------------------------------------------------
# baseline TIFF (strips or tiles, chunky or planar, uncompressed or Deflate, 8/16/32-bit integer and 32/64-bit float samples)
# ------------------------------------------------------------------------------------------------
_TIFF_TYPES = {1: ('B', 1), 2: ('c', 1), 3: ('H', 2), 4: ('I', 4), 5: ('II', 8), 6: ('b', 1), 8: ('h', 2), 9: ('i', 4), 11: ('f', 4),
               12: ('d', 8), 16: ('Q', 8)}


def _read_ifd(buf, off, bo):
    n, = struct.unpack_from(bo + 'H', buf, off)
    tags = {}
    for i in range(n):
        tag, typ, cnt = struct.unpack_from(bo + 'HHI', buf, off + 2 + 12 * i)
        if typ not in _TIFF_TYPES:
            continue
        fmt, size = _TIFF_TYPES[typ]
        voff = off + 2 + 12 * i + 8
        if size * cnt > 4:
            voff, = struct.unpack_from(bo + 'I', buf, voff)
        if typ == 5:
            vals = struct.unpack_from(bo + 'I' * (2 * cnt), buf, voff)
            vals = tuple(vals[2 * k] / max(vals[2 * k + 1], 1) for k in range(cnt))
        elif typ == 2:
            vals = (bytes(buf[voff:voff + cnt]),)
        else:
            vals = struct.unpack_from(bo + fmt * cnt, buf, voff)
        tags[tag] = vals
    return tags


def read_tiff(path):
    """-> np.ndarray [H, W] (one sample per pixel) or [H, W, C], in the file's sample type"""
    with open(path, 'rb') as fh:
        buf = fh.read()
    if buf[:2] == b'II':
        bo = '<'
    elif buf[:2] == b'MM':
        bo = '>'
    else:
        raise ValueError(f'{path}: not a TIFF file')
    magic, ifd = struct.unpack_from(bo + 'HI', buf, 2)
    if magic != 42:
        raise ValueError(f'{path}: BigTIFF / unknown TIFF magic {magic} is not supported')
    t = _read_ifd(buf, ifd, bo)
    W, H = t[256][0], t[257][0]
    spp = t.get(277, (1,))[0]
    bits = t.get(258, (1,))
    fmt = t.get(339, (1,))[0]
    comp = t.get(259, (1,))[0]
    planar = t.get(284, (1,))[0]
    pred = t.get(317, (1,))[0]
    if len(set(bits)) != 1:
        raise ValueError(f'{path}: mixed BitsPerSample {bits}')
    kinds = {(1, 8): 'u1', (1, 16): 'u2', (1, 32): 'u4', (2, 8): 'i1', (2, 16): 'i2', (2, 32): 'i4', (3, 32): 'f4', (3, 64): 'f8'}
    if (fmt, bits[0]) not in kinds:
        raise ValueError(f'{path}: SampleFormat {fmt} with {bits[0]} bits is not supported')
    dt = np.dtype(bo + kinds[(fmt, bits[0])])
    if comp not in (1, 8, 32946):
        raise ValueError(f'{path}: TIFF compression {comp} is not supported (uncompressed and Deflate are)')

    def chunk(o, n):
        raw = buf[o:o + n]
        return zlib.decompress(raw) if comp != 1 else raw
    planes = spp if planar == 2 else 1
    cpp = 1 if planar == 2 else spp          # samples per pixel inside one chunk
    out = np.zeros((planes, H, W, cpp), dtype=dt.newbyteorder('='))
    if 322 in t:                             # tiles
        tw, th = t[322][0], t[323][0]
        offs, cnts = t[324], t[325]
        tx, ty = (W + tw - 1) // tw, (H + th - 1) // th
        for p in range(planes):
            for j in range(ty):
                for i in range(tx):
                    k = (p * ty + j) * tx + i
                    a = np.frombuffer(chunk(offs[k], cnts[k]), dtype=dt, count=th * tw * cpp).reshape(th, tw, cpp)
                    if pred == 2:
                        a = np.cumsum(a, axis=1, dtype=dt)
                    hh, ww = min(th, H - j * th), min(tw, W - i * tw)
                    out[p, j * th:j * th + hh, i * tw:i * tw + ww] = a[:hh, :ww]
    else:                                    # strips
        rps = min(t.get(278, (H,))[0], H)
        offs, cnts = t[273], t[279]
        spi = (H + rps - 1) // rps
        for p in range(planes):
            for s in range(spi):
                rows = min(rps, H - s * rps)
                a = np.frombuffer(chunk(offs[p * spi + s], cnts[p * spi + s]), dtype=dt, count=rows * W * cpp).reshape(rows, W, cpp)
                if pred == 2:
                    a = np.cumsum(a, axis=1, dtype=dt)
                out[p, s * rps:s * rps + rows] = a
    img = out[0] if planar != 2 else np.moveaxis(out[..., 0], 0, -1)
    return img[..., 0] if img.shape[-1] == 1 else img


def write_tiff(path, array, compress=False, big_endian=False, rows_per_strip=None):
    """array [H, W] or [H, W, C] (chunky); uint8 / uint16 / int16 / float32 samples"""
    a = np.ascontiguousarray(array)
    if a.ndim == 2:
        a = a[..., None]
    H, W, C = a.shape
    kinds = {'uint8': (1, 8), 'uint16': (1, 16), 'int16': (2, 16), 'float32': (3, 32)}
    if a.dtype.name not in kinds:
        raise ValueError(f'write_tiff: dtype {a.dtype} is not supported')
    fmt, bits = kinds[a.dtype.name]
    bo = '>' if big_endian else '<'
    rps = rows_per_strip or H
    strips = []
    for r0 in range(0, H, rps):
        raw = a[r0:r0 + rps].astype(a.dtype.newbyteorder(bo)).tobytes()
        strips.append(zlib.compress(raw) if compress else raw)
    n = len(strips)
    entries = []   # (tag, type, values)
    entries.append((256, 4, [W]))
    entries.append((257, 4, [H]))
    entries.append((258, 3, [bits] * C))
    entries.append((259, 3, [8 if compress else 1]))
    entries.append((262, 3, [1]))
    entries.append((273, 4, None))          # strip offsets (patched below)
    entries.append((277, 3, [C]))
    entries.append((278, 4, [rps]))
    entries.append((279, 4, [len(s) for s in strips]))
    entries.append((284, 3, [1]))
    if C > 3:
        entries.append((338, 3, [0] * (C - 1)))   # ExtraSamples: unspecified
    entries.append((339, 3, [fmt] * C))
    entries.sort(key=lambda e: e[0])
    head = 8
    ifd_size = 2 + 12 * len(entries) + 4
    extra_off = head + ifd_size
    extra = b''
    fmt_of = {3: 'H', 4: 'I'}
    size_of = {3: 2, 4: 4}
    # first pass: size of out-of-line values to know where the pixel data start
    ool = sum(size_of[typ] * (n if vals is None else len(vals)) for _, typ, vals in entries
              if size_of[typ] * (n if vals is None else len(vals)) > 4)
    data_off = extra_off + ool + (ool & 1)
    strip_offs = []
    o = data_off
    for s in strips:
        strip_offs.append(o)
        o += len(s) + (len(s) & 1)
    ifd = struct.pack(bo + 'H', len(entries))
    for tag, typ, vals in entries:
        vals = strip_offs if vals is None else vals
        cnt = len(vals)
        payload = struct.pack(bo + fmt_of[typ] * cnt, *vals)
        if len(payload) <= 4:
            ifd += struct.pack(bo + 'HHI', tag, typ, cnt) + payload.ljust(4, b'\0')
        else:
            ifd += struct.pack(bo + 'HHII', tag, typ, cnt, extra_off + len(extra))
            extra += payload
    ifd += struct.pack(bo + 'I', 0)
    if len(extra) & 1:
        extra += b'\0'
    with open(path, 'wb') as fh:
        fh.write((b'MM' if big_endian else b'II') + struct.pack(bo + 'HI', 42, head))
        fh.write(ifd)
        fh.write(extra)
        for s in strips:
            fh.write(s + (b'\0' if len(s) & 1 else b''))


def load_image(path):
    """TIFF -> float64 array [H, W, C] or [H, W] (reference dataset/utils.py:29-39; tifffile when it is installed)"""
    try:
        import tifffile
        return np.array(tifffile.imread(path), dtype=np.double)
    except ImportError:
        return np.array(read_tiff(path), dtype=np.double)


def save_image(path, array):
    """[C, H, W] or [H, W] -> uint16 TIFF (reference dataset/utils.py:42-91 writes GeoTIFF through gdal with placeholder
    geo-referencing; the pixel payload is the same)"""
    a = np.asarray(array)
    a = np.clip(np.rint(a), 0, 65535).astype(np.uint16)
    write_tiff(path, np.moveaxis(a, 0, -1) if a.ndim == 3 else a)


def _is_pan_image(filename):
    return filename.endswith('pan.tif')


def get_image_id(filename):
    return filename.split('_')[0]


def pyr_down(img):
    """cv2.pyrDown of a 2-D array: 5x5 binomial blur ([1 4 6 4 1] / 16 per axis, BORDER_REFLECT_101) and even rows / columns"""
    k = np.array([1.0, 4.0, 6.0, 4.0, 1.0]) / 16.0
    x = ndimage.correlate1d(np.asarray(img, dtype=np.float64), k, axis=0, mode='mirror')
    x = ndimage.correlate1d(x, k, axis=1, mode='mirror')
    return x[::2, ::2]


@DATASETS.register_module()
class PSDataset(data.Dataset):
    def __init__(self, image_dirs, bit_depth, norm_input=False):
        super().__init__()
        self.image_dirs = image_dirs
        self.bit_depth = bit_depth
        self.norm_input = norm_input
        self.image_ids = []
        self.image_prefix_names = []
        for d in image_dirs:
            for x in sorted(os.listdir(d)):    # sorted: every rank must see the same order (os.listdir order is arbitrary)
                if _is_pan_image(x):
                    self.image_ids.append(get_image_id(x))
                    self.image_prefix_names.append(os.path.join(d, get_image_id(x)))

    def __getitem__(self, index):
        prefix = self.image_prefix_names[index]
        item = dict(input_lr=load_image(f'{prefix}_lr.tif').transpose(2, 0, 1),      # [C, h, w] LR MS
                    input_pan=load_image(f'{prefix}_pan.tif')[np.newaxis, :])         # [1, H, W] PAN
        if os.path.exists(f'{prefix}_mul.tif') and len(self.image_dirs) == 1:
            item['target'] = load_image(f'{prefix}_mul.tif').transpose(2, 0, 1)      # [C, H, W] ground truth
        item['input_pan_l'] = pyr_down(pyr_down(item['input_pan'][0]))[np.newaxis, :]
        item = {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in item.items()}
        if self.norm_input:
            item = data_normalize(item, self.bit_depth)
        item['image_id'] = self.image_ids[index]
        return item

    def __len__(self):
        return len(self.image_ids)


def data_augmentation(img_dict, aug_dict=None, rng=None):
    """reference dataset/utils.py:155-229 on [N, C, H, W] tensors.  `aug_dict` maps 'ud_flip' / 'lr_flip' / 'r4_crop' /
    'r2_crop' to a probability and is overwritten with the drawn booleans, as there.  Every selected transform is applied to the
    ORIGINAL image and the last one wins (the reference assigns `ret[name] = op(imgs)` four times) -- kept, so that a given
    draw yields the reference's batch.  rng: callable returning U[0,1) (numpy.random.random by default)."""
    rnd = rng or np.random.random

    def flip(x, dim):
        return torch.flip(x, dims=[dim])

    def crop_resize(imgs, st, n):
        h, w = imgs.shape[2], imgs.shape[3]
        imgs = imgs[:, :, st[0]:h // n * (n - 1) + st[0], st[1]:w // n * (n - 1) + st[1]]
        return torch.nn.functional.interpolate(imgs, size=[h, w], mode='bicubic', align_corners=True)
    if aug_dict is None:
        return img_dict
    need = False
    for a in aug_dict:
        aug_dict[a] = bool(rnd() < aug_dict[a])
        need = need or aug_dict[a]
    if not need:
        return img_dict
    if aug_dict.get('r4_crop'):
        d1 = int(img_dict['input_lr'].size(2) // 4 * rnd())
        d2 = int(img_dict['input_lr'].size(3) // 4 * rnd())
    if aug_dict.get('r2_crop'):
        d3 = int(img_dict['input_lr'].size(2) // 2 * rnd())
        d4 = int(img_dict['input_lr'].size(3) // 2 * rnd())
    ret = dict(image_id=img_dict['image_id'])
    for name, imgs in img_dict.items():
        if name == 'image_id':
            continue
        low = name in ('input_lr', 'input_pan_l')
        if aug_dict.get('ud_flip'):
            ret[name] = flip(imgs, 2)
        if aug_dict.get('lr_flip'):
            ret[name] = flip(imgs, 3)
        if aug_dict.get('r4_crop'):
            ret[name] = crop_resize(imgs, (d1, d2) if low else (d1 * 4, d2 * 4), 4)
        if aug_dict.get('r2_crop'):
            ret[name] = crop_resize(imgs, (d3, d4) if low else (d3 * 4, d4 * 4), 2)
    return ret


class ShardedSampler(data.Sampler):
    """Rank `rank` of `world` gets an equal, disjoint slice of every epoch's order (torch DistributedSampler semantics: the order
    is padded by wrapping around so that all ranks run the same number of steps -- or truncated with drop_last).  The permutation
    depends on (seed, epoch) only, so all ranks agree without communicating.  set_epoch(e) before each epoch.
    pad=False is the EVALUATION form: rank r gets order[r::world] and nothing else -- every element exactly once over the ranks, which
    then hold unequal counts when world does not divide n (a padded evaluation would score the wrapped images twice)."""

    def __init__(self, n, rank=0, world=1, shuffle=True, seed=0, drop_last=False, pad=True):
        if not 0 <= rank < world:
            raise ValueError(f'rank {rank} outside world {world}')
        self.n, self.rank, self.world, self.shuffle, self.seed, self.drop_last = n, rank, world, shuffle, seed, drop_last
        self.pad = bool(pad)
        self.epoch = 0
        self.per_rank = n // world if drop_last else ((n + world - 1) // world if self.pad else len(range(rank, n, world)))

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        if not self.pad and not self.drop_last:
            return iter(order[self.rank::self.world])
        total = self.per_rank * self.world
        while order and len(order) < total:      # pad by wrapping around (every rank runs the same number of steps)
            order += order[:total - len(order)]
        order = order[:total]
        return iter(order[self.rank:total:self.world])

    def __len__(self):
        return self.per_rank


class PrefetchLoader:
    """Iterates `loader` and yields batches whose tensors already live on `device`: batch i + depth is being copied from
    pinned host memory on a side stream while batch i is consumed (the reference copies synchronously, `set_batch_cuda`
    dataset/utils.py:98-110).  The consumer stream waits on the copy's event, and the tensors are marked as used by it
    (record_stream) so the caching allocator does not recycle them early.  On a CPU device it is a plain pass-through."""

    def __init__(self, loader, device, depth=2):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch, stream):
        with torch.cuda.stream(stream):
            moved = {k: (v.pin_memory().to(self.device, non_blocking=True) if torch.is_tensor(v) and not v.is_pinned()
                         else (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v)) for k, v in batch.items()}
            ev = torch.cuda.Event()
            ev.record(stream)
        return moved, ev

    def __iter__(self):
        if self.device.type != 'cuda':
            for batch in self.loader:
                yield {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
            return
        stream = torch.cuda.Stream(device=self.device)
        it = iter(self.loader)
        queue = []
        for _ in range(self.depth):
            b = next(it, None)
            if b is None:
                break
            queue.append(self._stage(b, stream))
        while queue:
            moved, ev = queue.pop(0)
            b = next(it, None)
            if b is not None:
                queue.append(self._stage(b, stream))
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for v in moved.values():
                if torch.is_tensor(v):
                    v.record_stream(cur)
            yield moved


def build_loader(set_cfg, rank=0, world=1, device=None, seed=0, prefetch_depth=2, evaluation=False):
    """main.py:71-86: `set_cfg` = dict(dataset=dict(type='PSDataset', ...), batch_size=, num_workers=, shuffle=) -> loader.
    With world > 1 the shuffle flag moves into a ShardedSampler (batch_size is per rank); with `device` the loader is wrapped
    in a PrefetchLoader.  evaluation=True: the sampler does not pad (each image on exactly one rank).  Returns (loader, sampler-or-None)."""
    cfg = dict(set_cfg)
    cfg['dataset'] = build_dataset(cfg['dataset'])
    sampler = None
    if world > 1:
        sampler = ShardedSampler(len(cfg['dataset']), rank, world, shuffle=bool(cfg.pop('shuffle', False)), seed=seed,
                                 drop_last=bool(cfg.get('drop_last', False)), pad=not evaluation)
        cfg['sampler'] = sampler
        cfg['shuffle'] = False
    if device is not None and torch.device(device).type == 'cuda':
        cfg.setdefault('pin_memory', True)
    loader = data.DataLoader(**cfg)
    if device is not None:
        loader = PrefetchLoader(loader, device, prefetch_depth)
    return loader, sampler
