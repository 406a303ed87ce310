"""CPU (no GPU) checks of the drop-in boundary: registry / config / module surface / C-ABI symbol table."""
import ctypes
import os
import pickle
import re

import numpy as np
import pytest
import torch

from helpers import state_shapes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_and_module_surface():
    import lgteun_amd
    from lgteun_amd.compat import Config
    assert 'UnlgFormer' in lgteun_amd.MODELS and lgteun_amd.MODELS.get('UnlgFormer') is lgteun_amd.UnlgFormer
    with pytest.raises(KeyError):
        lgteun_amd.build_model('NoSuchModel')
    for C, K in ((4, 2), (8, 1)):
        net = lgteun_amd.Pansharpening(Config(ms_chans=C), None, stage=K)
        sd = net.state_dict()
        assert list(sd.keys()) == lgteun_amd.canonical_names(C, K)          # same keys, same ORDER as the reference
        assert {k: tuple(v.shape) for k, v in sd.items()} == state_shapes(C, K)
        assert all(float(sd[f'eta.{i}']) == pytest.approx(0.1) for i in range(K))   # unlg_former.py:40
        blob = pickle.dumps(net)                                              # reference pickles whole modules (base_model.py:362)
        net2 = pickle.loads(blob)
        assert list(net2.state_dict().keys()) == list(sd.keys())
        with pytest.raises(RuntimeError):                                     # no CPU compute path
            net(torch.zeros(1, C, 8, 8), torch.zeros(1, 1, 32, 32))


def test_runner_from_config_file(tmp_path):
    """configs/unlg_former.py-style entry -> Config.fromfile -> build_model(cfg.model_type, ...) (main.py:61-122)"""
    import logging
    import lgteun_amd
    from lgteun_amd.compat import Config
    cfg_file = tmp_path / 'unlg_former.py'
    cfg_file.write_text(
        "name = 'LGTEUN'\nms_chans = 4\nmodel_type = 'UnlgFormer'\ndatas = 'GF-2'\n"
        f"work_dir = r'{tmp_path}/out'\ncuda = True\nbit_depth = 11\nmax_iter = 10\nseed = 19971118\n"
        "optim_cfg = {'core_module': dict(type='Adam', betas=(0.9, 0.999), lr=1.5e-3)}\n"
        "sched_cfg = dict(step_size=25900, gamma=0.85)\nloss_cfg = {'rec_loss': dict(type='l1', w=1.)}\n"
        "model_cfg = {'core_module': dict(stage=2)}\n")
    cfg = Config.fromfile(str(cfg_file))
    assert cfg.model_type == 'UnlgFormer' and cfg.loss_cfg['rec_loss'].w == 1.0 and cfg.model_cfg['core_module']['stage'] == 2
    runner = lgteun_amd.build_model(cfg.model_type, cfg, logging.getLogger('t'), None, None, None)
    core = runner.module_dict['core_module']
    assert core.stage == 2 and core.in_channels == 4
    assert 'rec_loss' in runner.loss_module and runner.loss_module['rec_loss'].get_type() == 'l1'
    runner.set_optim()
    runner.set_sched()
    assert runner.optim_dict['core_module'].is_fused_lgteun
    lrs = []
    for _ in range(3):
        lrs.append(runner.optim_dict['core_module'].param_groups[0]['lr'])
        runner.sched_dict['core_module'].step()
    assert lrs == [1.5e-3] * 3
    bad = Config(dict(cfg, loss_cfg={'rec_loss': dict(type='huber', w=1.)}))
    with pytest.raises(SystemExit):                                          # losses.py:33-34 behaviour
        lgteun_amd.build_model('UnlgFormer', bad, None, None, None, None)


def test_c_abi_exports_every_declared_symbol():
    from lgteun_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'lgteun_hip.h')).read()
    declared = set(re.findall(r'\b(lg_[a-z0-9_]+|lgteun_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'lg_config', 'lg_plan'}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('HIP library not built in this environment')
    L = ctypes.CDLL(_lib.LIB_PATH)          # load only: no compute calls without a GPU
    for name in declared:
        assert hasattr(L, name), name
    L.lg_version.restype = ctypes.c_char_p
    assert b'gfx950' in L.lg_version()
    # argument validation works without a device
    L.lg_plan_create.restype = ctypes.c_int32
    cfg = _lib.LgConfig(5, 2, 32, 32, 0)
    offs = (ctypes.c_int64 * 4)(0, 4, 8, 12)
    out = ctypes.c_void_p()
    assert L.lg_plan_create(ctypes.byref(cfg), offs, 4, ctypes.byref(out)) < 0
    L.lg_last_error.restype = ctypes.c_char_p
    assert b'C must be 4 or 8' in L.lg_last_error()
    # A/B switches are lg_config.variant bits: unknown bits are rejected, and the library imports no getenv
    cfg = _lib.LgConfig(4, 2, 32, 32, 0, 1 << 20)
    n = 12 + 2 + 119 * 2
    offs = (ctypes.c_int64 * n)(*[4 * i for i in range(n)])
    assert L.lg_plan_create(ctypes.byref(cfg), offs, n, ctypes.byref(out)) < 0 and b'variant' in L.lg_last_error()
    cfg = _lib.LgConfig(4, 2, 32, 32, 0, _lib.LG_VAR_FFN_STRIP | _lib.LG_VAR_ATTN_BWD_R3)
    assert L.lg_plan_create(ctypes.byref(cfg), offs, n, ctypes.byref(out)) == 0
    L.lg_plan_destroy.argtypes = [ctypes.c_void_p]
    L.lg_plan_destroy(out)
    import shutil
    import subprocess
    if shutil.which('nm'):
        und = subprocess.run(['nm', '-D', '--undefined-only', _lib.LIB_PATH], capture_output=True, text=True).stdout
        assert 'getenv' not in und, 'the product library must not read environment variables'


def test_variant_word_from_the_diagnostic_environment_variables():
    from lgteun_amd import _lib
    assert _lib.variant_from_env({}) == 0
    assert _lib.variant_from_env({'LG_FFN_IMPL': 'strip'}) == _lib.LG_VAR_FFN_STRIP
    assert _lib.variant_from_env({'LG_FFN_SAVE': '5', 'LG_FFN_DWBWD': 'tile'}) == _lib.LG_VAR_FFN_SAVE5 | _lib.LG_VAR_FFN_DWBWD_TILE
    assert _lib.variant_from_env({'LG_FFN_SAVE': '3', 'LG_FFN_BWD32': 'pair', 'LG_ATTN_BWD': 'r3'}) == \
        _lib.LG_VAR_FFN_SAVE3 | _lib.LG_VAR_FFN_BWD32_PAIR | _lib.LG_VAR_ATTN_BWD_R3
    assert _lib.variant_from_env({'LG_FFN_SAVE': '2', 'LG_FFN_IMPL': 'split'}) == 0
    assert _lib.variant_from_env({'LG_DSTEP': 'tiles'}) == _lib.LG_VAR_DSTEP_TILES
    assert _lib.variant_from_env({'LG_ATTN_FWD': 'valu'}) == _lib.LG_VAR_ATTN_FWD_VALU
    assert _lib.variant_from_env({'LG_FFN_SPLIT': 'bf16x3'}) == _lib.LG_VAR_FFN_BF16X3
    assert _lib.variant_from_env({'LG_FFT': 'full'}) == _lib.LG_VAR_FFT_FULL
    assert _lib.variant_from_env({'LG_FFN_BWD_SPLIT': 'bf16x3'}) == _lib.LG_VAR_FFN_BWD_BF16X3
    assert _lib.variant_from_env({'LG_ATTN_BWD_CORE': 'm'}) == _lib.LG_VAR_ATTN_BWD_CORE_M
    assert _lib.variant_from_env({'LG_ATTN_BWD_STATS': 'recompute'}) == _lib.LG_VAR_ATTN_BWD_RESTATS == 1 << 16
    hdr = open(os.path.join(ROOT, 'include', 'lgteun_hip.h')).read()
    for name in ('LG_VAR_FFN_STRIP', 'LG_VAR_FFN_TILE', 'LG_VAR_FFN_XP'):
        assert int(re.search(rf'#define {name} (\d+)u', hdr).group(1)) == getattr(_lib, name)
    for name in ('LG_VAR_FFN_SAVE3', 'LG_VAR_FFN_SAVE5', 'LG_VAR_FFN_BWD32_PAIR', 'LG_VAR_FFN_DWBWD_TILE', 'LG_VAR_ATTN_BWD_R3', 'LG_VAR_DSTEP_TILES', 'LG_VAR_ATTN_FWD_VALU', 'LG_VAR_FFN_BF16X3', 'LG_VAR_FFT_FULL', 'LG_VAR_FFN_BWD_BF16X3', 'LG_VAR_ATTN_BWD_CORE_M'):
        a, b = re.search(rf'#define {name} \((\d+)u << (\d+)\)', hdr).groups()
        assert int(a) << int(b) == getattr(_lib, name), name


def test_metrics_match_oracle():
    from lgteun_amd import metrics as mtc
    from oracle import lgteun_oracle as orc
    rng = np.random.default_rng(0)
    a, b = rng.uniform(0, 2047, (16, 16, 4)), rng.uniform(0, 2047, (16, 16, 4))
    for f in ('psnr', 'sam', 'ergas'):      # the oracle's are pinned by the reference's own values (test_oracle_golden.py)
        assert abs(getattr(mtc, f)(a, b) - getattr(orc, f)(a, b)) < 1e-12 * abs(getattr(orc, f)(a, b)), f


def test_metrics_reproduce_reference_golden_values(manifest):
    """PSNR / SAM / ERGAS of the product's metrics module on the reference's own fp32 output == the numbers the reference's
    metrics.py printed for it (tests/golden/net_*.npz `metrics`, written by tools/gen_goldens.py)"""
    from conftest import load_gold
    from lgteun_amd import metrics as mtc
    from oracle import detweights as dw
    for name in ('net_c4_k2_p32', 'net_c8_k2_p32', 'net_c4_k4_p64'):
        m, g = manifest[name], load_gold(name)
        _, _, gt = dw.make_inputs(m['B'], m['C'], m['h'], m['h'], seed=m['seed'], kind=m['kind'])
        o = np.transpose(g['out_fp32'][0], (1, 2, 0)).astype(np.float64) * 2047.5
        t = np.transpose(gt[0], (1, 2, 0)).astype(np.float64) * 2047.5
        got = np.array([mtc.psnr(o, t), mtc.sam(o, t), mtc.ergas(o, t)])
        assert np.allclose(got, g['metrics'], rtol=1e-12, atol=0), (name, got, g['metrics'])


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='needs the reference tree (build container only)')
def test_reference_checkpoint_converter(tmp_path):
    """a checkpoint written the reference's way (pickled module objects, base_model.py:354-369) converts to a plain
    state_dict that the product module loads bit-exactly"""
    import subprocess
    import sys
    import lgteun_amd
    from lgteun_amd.compat import Config
    src, dst = tmp_path / 'model_iter_7.pth', tmp_path / 'model_iter_7.state.pth'
    gen = (
        "import sys, torch; sys.path.insert(0, r'%s/tools'); from _ref_import import import_reference; R = import_reference();"
        "torch.manual_seed(3); net = R.Pansharpening(R.Config(ms_chans=4), None, stage=2);"
        "torch.save({'core_module': net, 'iter_num': 7}, r'%s')" % (ROOT, src))
    subprocess.run([sys.executable, '-W', 'ignore', '-c', gen], check=True)
    subprocess.run([sys.executable, '-W', 'ignore', os.path.join(ROOT, 'tools', 'convert_checkpoint.py'), str(src), str(dst)], check=True)
    ck = torch.load(dst, weights_only=True)              # reference-free: plain tensors only
    assert ck['iter_num'] == 7
    net = lgteun_amd.Pansharpening(Config(ms_chans=4), None, stage=2)
    net.load_state_dict(ck['core_module'])
    for k, v in net.state_dict().items():
        assert torch.equal(v, ck['core_module'][k]), k


@pytest.mark.skipif(not os.path.isfile('/root/reference/configs/unlg_former.py'), reason='needs the reference tree (build container only)')
def test_reference_config_file_builds_the_model(tmp_path):
    """the reference's own entry config (configs/unlg_former.py:23 model_type, :92-94 model_cfg, :82-86 optimiser / schedule)
    loads through Config.fromfile -> build_model exactly as main.py:90 does, and yields the canonical parameter surface"""
    import logging
    import lgteun_amd
    from lgteun_amd.compat import Config
    cfg = Config.fromfile('/root/reference/configs/unlg_former.py')
    assert cfg.model_type == 'UnlgFormer' and cfg.ms_chans == 8 and cfg.model_cfg['core_module']['stage'] == 2
    cfg.work_dir = str(tmp_path)                       # the file's own work_dir is relative to the author's checkout
    runner = lgteun_amd.build_model(cfg.model_type, cfg, logging.getLogger('cfgtest'), None, None, None)
    assert isinstance(runner, lgteun_amd.UnlgFormer)
    core = runner.module_dict['core_module']
    assert list(core.state_dict().keys()) == lgteun_amd.canonical_names(8, 2)
    assert sum(p.numel() for p in core.parameters()) == 2 * 269848 + 2 + 4 * 80 + 9 + 16      # 2 LGTs + eta + D/DT + R/RT at C=8
    runner.set_optim()
    runner.set_sched()
    opt = runner.optim_dict['core_module']
    assert getattr(opt, 'is_fused_lgteun', False) and opt.param_groups[0]['lr'] == 1.5e-3 and opt.param_groups[0]['betas'] == (0.9, 0.999)
    assert runner.sched_dict['core_module'].step_size == cfg.step and runner.sched_dict['core_module'].gamma == 0.85
    assert runner.train_out.endswith('/WV-3/train_out')


def test_checkpoints_are_plain_tensors_and_pickled_ones_need_consent(tmp_path):
    """Base_model.save writes {module: state_dict, iter_num, optim} that `weights_only=True` loads (nothing in the file is executed);
    the reference's own format -- whole pickled module objects, base_model.py:362-368 -- is written only with cfg.pickle_modules and
    read back only with explicit consent (allow_pickle / cfg.allow_pickled_checkpoint); losses: unknown types end the run like
    the reference (losses.py:34), zero-weight losses are dropped (:229)"""
    import logging
    import lgteun_amd
    from lgteun_amd.compat import Config
    from lgteun_amd.losses import get_loss_module

    def mk(**extra):
        cfg = Config(dict(ms_chans=4, work_dir=str(tmp_path), datas='GF-2', loss_cfg={'rec_loss': dict(type='l1', w=1.)},
                          model_cfg={'core_module': dict(stage=1)}, **extra))
        return lgteun_amd.build_model('UnlgFormer', cfg, logging.getLogger('ck'), None, None, None)
    torch.manual_seed(5)
    a = mk()
    path = a.save(iter_id=9)
    ck = torch.load(path, weights_only=True)
    assert ck['iter_num'] == 9 and set(ck) == {'iter_num', 'core_module', 'optim'}
    b = mk()
    b.load_checkpoint(path)
    assert b.last_iter == 9
    for (k, v), (k2, v2) in zip(a.module_dict['core_module'].state_dict().items(), b.module_dict['core_module'].state_dict().items()):
        assert k == k2 and torch.equal(v, v2)
    # the reference's format
    c = mk(pickle_modules=True)
    p2 = c.save(iter_id=11)
    with pytest.raises(RuntimeError, match='convert_checkpoint'):
        mk().load_checkpoint(p2)
    d = mk()
    d.load_checkpoint(p2, allow_pickle=True)
    assert d.last_iter == 11
    e = mk(allow_pickled_checkpoint=True)
    e.load_pretrained(p2)
    for (k, v), (_, v2) in zip(c.module_dict['core_module'].state_dict().items(), e.module_dict['core_module'].state_dict().items()):
        assert torch.equal(v, v2), k
    # loss factory
    assert set(get_loss_module(Config(dict(loss_cfg={'rec_loss': dict(type='l2', w=0.5)})), None)) == {'rec_loss'}
    assert get_loss_module(Config(dict(loss_cfg={'rec_loss': dict(type='l1', w=0.0)})), None) == {}
    with pytest.raises(SystemExit):
        get_loss_module(Config(dict(loss_cfg={'rec_loss': dict(type='huber', w=1.0)})), None)
    with pytest.raises(SystemExit):
        get_loss_module(Config(dict(loss_cfg={'QNR_loss': dict(w=1.0)})), None)
    m = get_loss_module(Config(dict(loss_cfg={'rec_loss': dict(type='l1', w=1.0)})), None)['rec_loss']
    x, y = torch.tensor([1.0, 3.0]), torch.tensor([0.0, 1.0])
    assert m.get_type() == 'l1' and float(m(out=x, gt=y)) == 1.5
