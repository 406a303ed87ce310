"""lgteun_amd -- MI355X-native (gfx950) implementation of the LGTEUN unfolding hot path behind the
reference's MODELS-registry / nn.Module surface.  Compute = hand-written HIP kernels behind a C ABI
(include/lgteun_hip.h); this package is the host-side mirror of the reference interface."""
from .builder import MODELS, build_model  # noqa: F401
from .unlg_former import Pansharpening, UnlgFormer  # noqa: F401
from .engine import Engine, FusedAdam, canonical_names  # noqa: F401
from .dataset import DATASETS, PSDataset, PrefetchLoader, ShardedSampler, build_dataset, build_loader  # noqa: F401
