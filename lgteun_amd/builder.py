"""MODELS registry + build_model -- mirror of reference models/base/builder.py:8-24."""
from .compat import Registry

MODELS = Registry('models')


def build_model(model_type, *args, **kwargs):
    if model_type not in MODELS:
        raise KeyError(f'Unrecognized task type {model_type}')
    model_cls = MODELS.get(model_type)
    return model_cls(*args, **kwargs)
