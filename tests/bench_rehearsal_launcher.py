"""Started by tests/conftest.py at the start of a `-m gpu` session, BEFORE pytest initialises the GPU (a process that has must not spawn
programs on this pool).  This launcher never touches the device: it waits until the session's other children (the two data-parallel
ranks and the one-rank RCCL worker) have exited -- the box allows six GPU processes at once -- and then runs the driver's own multi-GPU
command on the box's single MI355X as a rehearsal:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...

with LGTEUN_DDP_BACKEND=gloo (two ranks share the one GPU; the collectives are the ones RCCL serves on a node).
tests/test_gpu_zz_bench_rehearsal.py reads what it leaves in <outdir>: bench2.out / bench2.err / bench2.rc."""
import os
import socket
import subprocess
import sys
import time

outdir, pids = sys.argv[1], [int(p) for p in sys.argv[2:]]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
deadline = time.time() + 900
while time.time() < deadline and any(os.path.exists(f'/proc/{p}') and open(f'/proc/{p}/stat').read().split()[2] != 'Z' for p in pids):
    time.sleep(1.0)
with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
env = dict(os.environ, LGTEUN_DDP_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'LG_DDP_OVERLAP'):
    env.pop(k, None)
cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
       os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-live']
with open(os.path.join(outdir, 'bench2.out'), 'w') as fo, open(os.path.join(outdir, 'bench2.err'), 'w') as fe:
    try:
        rc = subprocess.run(cmd, stdout=fo, stderr=fe, env=env, cwd=ROOT, timeout=600).returncode
    except subprocess.TimeoutExpired:
        rc = -9
with open(os.path.join(outdir, 'bench2.rc.tmp'), 'w') as f:
    f.write(str(rc))
os.replace(os.path.join(outdir, 'bench2.rc.tmp'), os.path.join(outdir, 'bench2.rc'))
