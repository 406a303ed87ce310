"""Import the reference hot path (/root/reference) on CPU in THIS container only.

Used by tools/gen_goldens.py to (i) validate the oracle restatement and (ii) generate
the golden fixtures committed under tests/golden/.  Nothing of the reference travels:
this module only *imports* it from where it lies.  Recipe follows SURVEY.md §8(c).
"""
import sys
import types

REF = '/root/reference'


def _stub_modules():
    sys.dont_write_bytecode = True

    class Registry:
        def __init__(self, name):
            self.name = name
            self._d = {}

        def register_module(self, name=None, force=False, module=None):
            def deco(cls):
                self._d[name or cls.__name__] = cls
                return cls
            return deco

        def get(self, key):
            return self._d.get(key)

        def __contains__(self, key):
            return key in self._d

    class Config(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def __getitem__(self, k):
            v = dict.__getitem__(self, k)
            return Config(v) if isinstance(v, dict) and not isinstance(v, Config) else v

        def get(self, k, d=None):
            v = dict.get(self, k, d)
            return Config(v) if isinstance(v, dict) and not isinstance(v, Config) else v

        def copy(self):
            return Config(dict.copy(self))

    class Timer:
        def __init__(self):
            import time
            self.t = time.time()

        def since_last_check(self):
            import time
            now = time.time()
            d = now - self.t
            self.t = now
            return d

    def mkdir_or_exist(path, mode=0o777):
        import os
        os.makedirs(path, exist_ok=True)

    mmcv = types.ModuleType('mmcv')
    mmcv_utils = types.ModuleType('mmcv.utils')
    mmcv_utils.Registry = Registry
    mmcv.utils = mmcv_utils
    mmcv.Config = Config
    mmcv.Timer = Timer
    mmcv.mkdir_or_exist = mkdir_or_exist
    sys.modules['mmcv'] = mmcv
    sys.modules['mmcv.utils'] = mmcv_utils
    for name in ('cv2', 'gdal', 'osr', 'tifffile'):
        sys.modules.setdefault(name, types.ModuleType(name))
    numba = types.ModuleType('numba')
    numba.jit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    sys.modules.setdefault('numba', numba)
    models = types.ModuleType('models')
    models.__path__ = [REF + '/models']
    sys.modules['models'] = models
    return Config


def import_reference():
    Config = _stub_modules()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models.unlg_former import Pansharpening, UnlgFormer  # noqa
    import models.common.LGT as LGT  # noqa
    import models.common.basic_module_unformer_v2 as bmu  # noqa
    return types.SimpleNamespace(Pansharpening=Pansharpening, UnlgFormer=UnlgFormer, LGT=LGT, bmu=bmu,
                                 Config=Config)


def import_reference_dataset_utils():
    """dataset/utils.py of the reference (data_normalize / data_denormalize / data_augmentation, :155-263): pure torch / numpy
    functions behind imports of gdal / osr / tifffile / numba (stubbed as above; none of the three functions touches them).  The
    package's __init__ (which pulls in the TIFF dataset class and its registry) is bypassed the same way `models` is."""
    _stub_modules()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    pkg = types.ModuleType('dataset')
    pkg.__path__ = [REF + '/dataset']
    sys.modules['dataset'] = pkg
    import dataset.utils as du  # noqa
    return du
