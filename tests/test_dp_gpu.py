"""Data-parallel path with REAL HIP kernels: two ranks share the one GPU of the test box
(PASSL_DEVICE_INDEX=0) and talk through gloo (PASSL_DIST_BACKEND=gloo; RCCL refuses two ranks on one
device).  Everything but the transport is the production multi-GPU path: start-up broadcast, bucketed
overlapped gradient all-reduce fed by the backward kernels' ready marks, 1/world scaling inside the
optimizer kernel, gathered MoCo keys / SimCLR embeddings.  The worker asserts that the replicas see
different data yet hold bit-identical parameters after three steps (tests/dp_worker.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize('workload', ['moco', 'simclr', 'mae', 'clip', 'linprobe'])
def test_two_ranks_one_gpu(workload):
    env = dict(os.environ, PASSL_DIST_BACKEND='gloo', PASSL_DEVICE_INDEX='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'tests', 'dp_worker.py'), workload]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and ('DP-OK %s' % workload) in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
