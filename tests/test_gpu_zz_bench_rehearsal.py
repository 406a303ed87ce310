"""-m gpu: the driver's multi-GPU command rehearsed on the box's one MI355X: two ranks of `bench.py --gpus 2` under
torch.distributed.run, transport gloo, started by tests/bench_rehearsal_launcher.py (a child of the session start that never touches the
device) once the session's other child processes have exited.  What an 8-GPU SCALE run depends on and no single-rank run shows: the
rendezvous from the launcher's environment, ONE JSON line on rank 0's stdout and nothing else there, n_gpus / global_batch /
parallelism of the weak-scaling contract, the armed watchdog, a clean exit of both ranks."""
import json
import os
import time

import pytest

pytestmark = pytest.mark.gpu


def test_two_rank_bench_prints_one_json_line(request):
    job = getattr(request.config, '_lgteun_bench2_job', None)
    assert job is not None, 'conftest did not start the bench rehearsal launcher (no /dev/kfd?)'
    outdir, proc = job
    rcf = os.path.join(outdir, 'bench2.rc')
    t0 = time.time()
    while not os.path.exists(rcf) and time.time() - t0 < 900 and proc.poll() is None:
        time.sleep(1.0)
    for _ in range(10):
        if os.path.exists(rcf):
            break
        time.sleep(0.5)
    err = open(os.path.join(outdir, 'bench2.err')).read() if os.path.exists(os.path.join(outdir, 'bench2.err')) else ''
    assert os.path.exists(rcf), 'the rehearsal did not finish:\n' + err[-3000:]
    assert int(open(rcf).read()) == 0, err[-3000:]
    lines = [ln for ln in open(os.path.join(outdir, 'bench2.out')).read().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # the contract: ONE line on stdout (librccl / gloo banners go to stderr)
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 1
    assert d['config']['global_batch'] == 64 and d['config']['parallelism'] == 'dp2' and d['scaling'] == 'weak'
    assert d['config']['workload'].startswith('BASELINE configs[1]') and d['metric'].startswith('train image-pairs/sec, GF-2 4-band')
    assert d['watchdog_s'] == 600                        # armed by default for multi-rank runs (LG_BENCH_WATCHDOG overrides)
    assert d['value'] > 0 and abs(d['value'] - 64 / (d['ms_per_step'] * 1e-3)) < 1e-2 * d['value']
    assert d['ddp']['backend'] == 'gloo' and d['ddp']['bucket_form'] == 'one stream-ordered all-reduce behind the backward'
    assert 'roofline' in d and 'cpu_baseline' not in d   # the CPU leg runs at N = 1 only
