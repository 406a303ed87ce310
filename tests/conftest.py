import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """-m gpu sessions on a GPU box: start the two ranks of the Engine data-parallel check (tests/ddp_engine_worker.py) and the one-rank
    RCCL check (tests/rccl_one_rank_worker.py) NOW, as fresh child processes, before this process initialises the GPU (a process that has must not spawn programs on this
    pool).  tests/test_gpu_ddp_engine.py joins them.  Nothing here imports torch or touches the device."""
    import shutil
    import socket
    import subprocess
    import tempfile
    config = session.config
    markexpr = config.getoption('-m') or ''
    if 'gpu' not in markexpr or 'not gpu' in markexpr or not os.path.exists('/dev/kfd'):
        return
    if not os.path.exists(os.path.join(ROOT, 'lgteun_amd', '_lgteun_hip.so')):
        return
    if config.getoption('collectonly', False):
        return
    outdir = tempfile.mkdtemp(prefix='lgteun_ddp_')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs, logs = [], []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        log = os.path.join(outdir, f'rank{rank}.log')
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, '-W', 'ignore', os.path.join(ROOT, 'tests', 'ddp_engine_worker.py'), outdir],
                                      stdout=open(log, 'w'), stderr=subprocess.STDOUT, env=env, cwd=ROOT))
    config._lgteun_ddp_job = (outdir, procs, logs)
    # a third child: a process group of ONE rank on backend nccl (RCCL) -- tests/rccl_one_rank_worker.py
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port1 = s.getsockname()[1]
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port1), HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('LG_DDP_OVERLAP', None)
    log1 = os.path.join(outdir, 'rccl1.log')
    p1 = subprocess.Popen([sys.executable, '-W', 'ignore', os.path.join(ROOT, 'tests', 'rccl_one_rank_worker.py'), outdir],
                          stdout=open(log1, 'w'), stderr=subprocess.STDOUT, env=env, cwd=ROOT)
    config._lgteun_rccl_job = (outdir, p1, log1)
    # a fourth child, which never touches the device: it waits for the three above to exit, then runs `bench.py --gpus 2` under
    # torch.distributed.run (tests/bench_rehearsal_launcher.py; joined by tests/test_gpu_zz_bench_rehearsal.py)
    p2 = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'bench_rehearsal_launcher.py'), outdir] + [str(p.pid) for p in procs] + [str(p1.pid)],
                          stdout=open(os.path.join(outdir, 'launcher.log'), 'w'), stderr=subprocess.STDOUT, cwd=ROOT)
    config._lgteun_bench2_job = (outdir, p2)
    config.add_cleanup(lambda: shutil.rmtree(outdir, ignore_errors=True))


def pytest_sessionfinish(session, exitstatus):
    job = getattr(session.config, '_lgteun_ddp_job', None)
    if job:
        for p in job[1]:
            if p.poll() is None:
                p.kill()
    job = getattr(session.config, '_lgteun_rccl_job', None)
    if job and job[1].poll() is None:
        job[1].kill()
    job = getattr(session.config, '_lgteun_bench2_job', None)
    if job and job[1].poll() is None:
        job[1].kill()


@pytest.fixture(scope='session')
def manifest():
    import json
    with open(os.path.join(GOLD, 'manifest.json')) as f:
        return json.load(f)


def load_gold(name):
    import numpy as np
    return np.load(os.path.join(GOLD, name + '.npz'))
