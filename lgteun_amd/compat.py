"""Minimal stand-ins for the mmcv 1.x utilities the reference imports around the hot path
(Registry, Config.fromfile, get_logger, Timer, mkdir_or_exist).  Real mmcv is used when importable
(it is absent in this image; its API moved to mmengine in 2.x).  Reference call sites:
models/base/builder.py:8-24, main.py:146-150, models/base/base_model.py:42,173."""
import importlib.util
import logging
import os
import time

try:  # pragma: no cover - mmcv is not installed in the build image
    from mmcv import Config, Timer, mkdir_or_exist  # type: ignore
    from mmcv.utils import Registry, get_logger  # type: ignore
    HAVE_MMCV = True
except Exception:  # noqa: BLE001
    HAVE_MMCV = False

    class Registry:
        def __init__(self, name):
            self._name = name
            self._module_dict = {}

        @property
        def name(self):
            return self._name

        def __len__(self):
            return len(self._module_dict)

        def __contains__(self, key):
            return key in self._module_dict

        def get(self, key):
            return self._module_dict.get(key)

        def register_module(self, name=None, force=False, module=None):
            def _register(cls):
                key = name or cls.__name__
                if not force and key in self._module_dict:
                    raise KeyError(f'{key} is already registered in {self._name}')
                self._module_dict[key] = cls
                return cls
            if module is not None:
                return _register(module)
            return _register

    class ConfigDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def __getitem__(self, k):
            v = dict.__getitem__(self, k)
            if isinstance(v, dict) and not isinstance(v, ConfigDict):
                v = ConfigDict(v)
                dict.__setitem__(self, k, v)
            return v

        def get(self, k, default=None):
            return self[k] if k in self else default

        def copy(self):
            return ConfigDict(dict.copy(self))

    class Config(ConfigDict):
        """Config.fromfile: evaluate a plain-Python config module (configs/unlg_former.py style)."""

        @staticmethod
        def fromfile(path):
            spec = importlib.util.spec_from_file_location('_lgteun_cfg', path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            d = {k: v for k, v in vars(mod).items() if not k.startswith('__') and not callable(v)
                 and not isinstance(v, type(os))}
            return Config(d)

    class Timer:
        def __init__(self):
            self._t = time.time()

        def since_last_check(self):
            now = time.time()
            d = now - self._t
            self._t = now
            return d

    def mkdir_or_exist(path, mode=0o777):
        if path:
            os.makedirs(os.path.expanduser(path), mode=mode, exist_ok=True)

    def get_logger(name, log_file=None, log_level=logging.INFO):
        logger = logging.getLogger(name)
        if not logger.handlers:
            h = logging.StreamHandler()
            h.setFormatter(logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s'))
            logger.addHandler(h)
            if log_file:
                mkdir_or_exist(os.path.dirname(log_file))
                fh = logging.FileHandler(log_file)
                fh.setFormatter(h.formatter)
                logger.addHandler(fh)
        logger.setLevel(log_level if not isinstance(log_level, str) else getattr(logging, log_level))
        return logger
