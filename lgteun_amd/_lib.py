"""ctypes binding of the gfx950 HIP library (C ABI: include/lgteun_hip.h).

Loaded lazily so modules stay picklable (the reference pickles whole module objects,
models/base/base_model.py:354-369).  There is NO fallback: if the shared object is missing the
product path raises -- the CPU restatement in oracle/ is test infrastructure only.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('LGTEUN_HIP_LIB') or os.path.join(_HERE, '_lgteun_hip.so')   # override: diagnostic builds (tools/build_stamps.sh)

LG_FLAG_FAITHFUL = 1
LG_FLAG_SAVE = 2
LG_FLAG_DROPOUT = 4
LG_FLAG_BWD_LGT = 8
LG_FLAG_BWD_DATA = 16
LG_FLAG_CHAINED = 32
LG_FLAG_DEFER_DEAD = 64
KERNEL_IDS = {n: i for i, n in enumerate(['none', 'ffn1', 'ffn', 'fft', 'attn', 'upfuse', 'down', 'embed', 'tail', 'datastep', 'ffn1_bwd',
                                           'ffn2_bwd', 'fft_bwd', 'attn_bwd', 'wgrad'])}


class LgConfig(ctypes.Structure):
    _fields_ = [('C', c_int32), ('K', c_int32), ('H', c_int32), ('W', c_int32), ('precision', c_int32), ('variant', ctypes.c_uint32)]


# lg_config.variant bits (include/lgteun_hip.h): A/B kernels that compute the same function as the product path
LG_VAR_FFN_STRIP, LG_VAR_FFN_TILE, LG_VAR_FFN_XP = 1, 2, 3
LG_VAR_FFN_SAVE3, LG_VAR_FFN_SAVE5 = 1 << 2, 2 << 2
LG_VAR_FFN_BWD32_PAIR, LG_VAR_FFN_DWBWD_TILE, LG_VAR_ATTN_BWD_R3 = 1 << 4, 1 << 5, 1 << 6
LG_VAR_DSTEP_TILES = 1 << 7
LG_VAR_ATTN_FWD_VALU = 1 << 8
LG_VAR_FFN_BF16X3 = 1 << 9
LG_VAR_FFT_FULL = 1 << 10
LG_VAR_FFN_BWD_BF16X3 = 1 << 11
LG_VAR_ATTN_BWD_CORE_M = 1 << 12
LG_VAR_FFN_XS = 1 << 13
LG_VAR_ATTN_BF16X3 = 1 << 14
LG_VAR_FFN_H3_RECOMPUTE = 1 << 15
LG_VAR_ATTN_BWD_RESTATS = 1 << 16
LG_ABI_VERSION = 2   # include/lgteun_hip.h: checked against lg_abi_version() when the library is loaded


def variant_from_env(env=None):
    """the variant word of a new plan from the diagnostic LG_* environment variables (read HERE, on the Python side, when an Engine
    builds a plan -- the shared library itself never looks at the environment).  Unset = 0 = the product path."""
    env = os.environ if env is None else env
    v = {'strip': LG_VAR_FFN_STRIP, 'tile': LG_VAR_FFN_TILE, 'xp': LG_VAR_FFN_XP}.get(env.get('LG_FFN_IMPL', ''), 0)
    v |= {'3': LG_VAR_FFN_SAVE3, '5': LG_VAR_FFN_SAVE5}.get(env.get('LG_FFN_SAVE', ''), 0)
    if env.get('LG_FFN_BWD32') == 'pair':
        v |= LG_VAR_FFN_BWD32_PAIR
    if env.get('LG_FFN_DWBWD') == 'tile':
        v |= LG_VAR_FFN_DWBWD_TILE
    if env.get('LG_ATTN_BWD') in ('old', 'r3'):
        v |= LG_VAR_ATTN_BWD_R3
    if env.get('LG_DSTEP') == 'tiles':
        v |= LG_VAR_DSTEP_TILES
    if env.get('LG_ATTN_FWD', '') == 'valu':
        v |= LG_VAR_ATTN_FWD_VALU
    if env.get('LG_FFN_SPLIT', '') == 'bf16x3':
        v |= LG_VAR_FFN_BF16X3
    if env.get('LG_FFT', '') == 'full':
        v |= LG_VAR_FFT_FULL
    if env.get('LG_FFN_BWD_SPLIT', '') == 'bf16x3':
        v |= LG_VAR_FFN_BWD_BF16X3
    if env.get('LG_ATTN_BWD_CORE', '') == 'm':
        v |= LG_VAR_ATTN_BWD_CORE_M
    if env.get('LG_FFN_FWD', '') == 'xs':
        v |= LG_VAR_FFN_XS
    if env.get('LG_ATTN_SPLIT', '') == 'bf16x3':
        v |= LG_VAR_ATTN_BF16X3
    if env.get('LG_FFN_H3', '') == 'recompute':
        v |= LG_VAR_FFN_H3_RECOMPUTE
    if env.get('LG_ATTN_BWD_STATS', '') == 'recompute':
        v |= LG_VAR_ATTN_BWD_RESTATS
    return v


class LgteunHipError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes); every symbol include/lgteun_hip.h declares
SIGNATURES = {
    'lg_version': (c_char_p, []),
    'lg_abi_version': (c_int32, []),
    'lg_last_error': (c_char_p, []),
    'lg_plan_create': (c_int32, [POINTER(LgConfig), POINTER(c_int64), c_int32, POINTER(c_void_p)]),
    'lg_plan_destroy': (None, [c_void_p]),
    'lg_workspace_bytes': (c_size_t, [c_void_p, c_int32, c_int32]),
    'lgteun_forward': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int32, c_int32,
                                 c_uint64, c_void_p]),
    'lgteun_backward': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int32,
                                  c_int32, c_uint64, c_void_p]),
    'lgteun_dead_forward': (c_int32, [c_void_p, c_void_p, c_void_p, c_size_t, c_int32, c_int32, c_uint64, c_void_p]),
    'lg_l1_loss': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p]),
    'lg_adam_step': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_float, c_float,
                               c_float, c_float, c_float, c_void_p]),
    'lg_op_resample': (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'lg_op_data_step': (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                  c_void_p]),
    'lg_op_lgt': (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_int32, c_int32, c_uint64,
                            c_void_p]),
    'lg_op_block': (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_int32,
                              c_void_p]),
    'lg_dropout_mask': (c_int32, [c_uint64, c_int32, c_int32, c_int64, c_int64, c_void_p, c_void_p]),
    'lg_prof_enable': (c_int32, [c_int32, c_int32]),
    'lg_prof_reset': (c_int32, []),
    'lg_prof_pause': (None, [c_int32]),
    'lg_prof_read': (c_int32, [POINTER(ctypes.c_double), POINTER(c_int64)]),
    'lg_prof_disable': (None, []),
    'lg_kernel_name': (c_char_p, [c_int32]),
    'lg_op_data_step_bwd': (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_size_t, c_int32, c_void_p]),
    'lg_op_lgt_bwd': (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int32, c_int32,
                                c_uint64, c_void_p]),
    'lg_op_block_bwd': (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_size_t, c_int32, c_void_p]),
}


def lib():
    """Load (once) and return the ctypes library.  Raises LgteunHipError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LgteunHipError(
                f'HIP extension not built: {LIB_PATH} is missing. Run `make` (or __graft_entry__.build()). '
                'lgteun_amd has no CPU/PyTorch fallback.')
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        got = int(L.lg_abi_version())
        if got != LG_ABI_VERSION:   # a stale .so next to newer Python (or the reverse) would read structs / buffers of the other layout
            raise LgteunHipError(f'{LIB_PATH}: ABI version {got}, this binding was written for {LG_ABI_VERSION} (include/lgteun_hip.h): rebuild with `make`')
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().lg_last_error().decode(errors='replace')
        raise LgteunHipError(f'{what} failed (rc={rc}): {msg}')
