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


@pytest.mark.parametrize('workload', ['moco', 'simclr', 'mae', 'clip', 'clipx', 'linprobe'])
def test_two_ranks_one_gpu(workload):
    _run_worker(workload, 2, dict(PASSL_DIST_BACKEND='gloo', PASSL_DEVICE_INDEX='0'))


def _run_worker(workload, nproc, env_extra, launcher=True):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **env_extra)
    worker = os.path.join(ROOT, 'tests', 'dp_worker.py')
    if launcher:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), worker, workload]
    else:
        cmd = [sys.executable, worker, workload]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:                    # keep the whole worker output (pytest truncates the assertion message)
        try:
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            with open(os.path.join(ROOT, 'gpurun_out', 'dp_fail_%s.log' % workload), 'w') as f:
                f.write(r.stdout + '\n---- stderr ----\n' + r.stderr)
        except OSError:
            pass
    assert r.returncode == 0 and ('DP-OK %s' % workload) in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith('DP-OK')][-1]
    return line


def test_two_ranks_mocov3_cross_rank_keys():
    """MoCo-v3's gathered keys and rank-offset labels (reference passl/models/mocov3.py:161-183) against the
    reference's own two-rank forward (tests/golden/mocov3_small_2rank.npz) — see dp_worker.mocov3_run."""
    _run_worker('mocov3', 2, dict(PASSL_DIST_BACKEND='gloo', PASSL_DEVICE_INDEX='0'))


def test_two_ranks_linear_probe_engine_equals_one_rank_on_the_joint_batches(tmp_path):
    """The v2 linear-probe recipe under data parallelism through Engine.train() — see dp_worker.lp_run."""
    _run_worker('lp', 2, dict(PASSL_DIST_BACKEND='gloo', PASSL_DEVICE_INDEX='0', PASSL_DP_OUT=str(tmp_path)))


def test_two_ranks_simsiam_sync_batchnorm_equals_one_rank_on_the_joint_batch():
    """SyncBatchNorm (reference passl/models/simsiam.py:160-162): cross-rank BatchNorm statistics, forward and
    backward — see dp_worker.simsiam_run."""
    _run_worker('simsiam', 2, dict(PASSL_DIST_BACKEND='gloo', PASSL_DEVICE_INDEX='0'))


def test_two_ranks_simsiam_engine_overlapped_reducers_equal_blocking_grad_sync(tmp_path):
    """SimSiam's two parameter groups live in two arenas; the Engine gives each an overlapped gradient reducer
    (core/sync_utils.py:ReducerGroup) instead of the blocking per-buffer grad_sync of the reference
    (passl/core/sync_utils.py:18-43).  Two ranks, three steps through Engine.train(): replicas identical, and the
    overlapped run equals the blocking run bit for bit — see dp_worker.simsiam_engine_run."""
    env = dict(PASSL_DIST_BACKEND='gloo', PASSL_DEVICE_INDEX='0', PASSL_DP_OUT=str(tmp_path))
    a = _run_worker('simsiam_engine', 2, env)
    b = _run_worker('simsiam_engine', 2, dict(env, PASSL_DP_BLOCKING_GROUPS='1'))
    assert a.split('digest=')[1] == b.split('digest=')[1], (a, b)


def test_two_ranks_shuffle_bn_is_output_neutral():
    """MoCo's cross-rank batch shuffle (reference passl_v110/modeling/architectures/moco.py:107-152): all-gather
    the key view, a permutation drawn on rank 0 and broadcast, every rank encodes its slice of the permuted batch,
    the keys are gathered back and un-permuted.  The key encoder's BatchNorm runs on running statistics in PASSL
    (SURVEY 3.1 note A), so a key does not depend on its batch mates: the two-rank run with the shuffle on must
    reproduce the two-rank run without it — which it can only do if gather / broadcast / slice / inverse permutation
    are all right (a key attached to the wrong image moves the loss by O(0.1)).  Not bit for bit: a sample's
    position in the batch changes the rounding of its key in the 7th digit (the single-process check,
    test_moco_gpu.py::test_shuffle_bn_is_output_neutral, sees the same), and the tiny random-init net amplifies that
    through every update — the first step is held to 1e-5, the second to 5e-4."""
    env = dict(PASSL_DIST_BACKEND='gloo', PASSL_DEVICE_INDEX='0')
    plain = _run_worker('moco', 2, env)
    shuffled = _run_worker('moco_shuffle', 2, env)
    la, lb = ([float(v) for v in line.split('losses=')[1].split(',')] for line in (plain, shuffled))
    assert abs(la[0] - lb[0]) < 1e-5 and abs(la[1] - lb[1]) < 5e-4, (plain, shuffled)


@pytest.mark.parametrize('workload', ['moco', 'simclr'])
def test_rccl_world1(workload):
    """RCCL itself (backend "nccl") on the hardware: ONE rank on the one GPU, every data-parallel
    collective forced on (PASSL_DP_FORCE=1): communicator creation, flat start-up broadcast, the
    bucketed asynchronous gradient all-reduce launched from backward on RCCL's stream, the key /
    embedding all-gather (+ reduce-scatter for SimCLR).  A 1-rank all-reduce is the identity and the
    1/world scale is 1, so the run must end bit-identical to the same run with gloo as transport."""
    rccl = _run_worker(workload, 1, dict(PASSL_DP_FORCE='1', PASSL_EXPECT_BACKEND='nccl'))
    gloo = _run_worker(workload, 1, dict(PASSL_DP_FORCE='1', PASSL_DIST_BACKEND='gloo',
                                         PASSL_EXPECT_BACKEND='gloo'))
    plain = _run_worker(workload, 1, {}, launcher=False)        # no process group at all
    assert rccl.split('digest=')[1] == plain.split('digest=')[1], (rccl, plain)
    assert gloo.split('digest=')[1] == plain.split('digest=')[1], (gloo, plain)


def test_rccl_world1_dedicated_collective_stream_gives_the_same_bits():
    """The gradient collectives are issued from the weight-gradient stream by default (hip/streams.py:comm_stream, round 6:
    a fifth stream shares a hardware queue with a product stream); the dedicated stream stays selectable and must end in
    the same state."""
    own = _run_worker('moco', 1, dict(PASSL_DP_FORCE='1', PASSL_EXPECT_BACKEND='nccl', PASSL_DP_COMM_STREAM='own'))
    side = _run_worker('moco', 1, dict(PASSL_DP_FORCE='1', PASSL_EXPECT_BACKEND='nccl'))
    assert own.split('digest=')[1] == side.split('digest=')[1], (own, side)


def test_collectives_are_issued_from_the_side_stream_by_default(monkeypatch):
    import torch
    from passl_amd.hip import streams
    dev = torch.device('cuda', 0)
    monkeypatch.delenv('PASSL_DP_COMM_STREAM', raising=False)
    streams._comm_streams.clear()
    assert streams.comm_stream(dev) is streams.side_stream(dev)
    monkeypatch.setenv('PASSL_DP_COMM_STREAM', 'own')
    streams._comm_streams.clear()
    assert streams.comm_stream(dev) != streams.side_stream(dev)
    streams._comm_streams.clear()


def test_bench_self_launch_gpus1_and_world1_launcher():
    """bench.py under the driver's launcher form with one rank (RCCL world 1 is NOT forced here:
    the production gate is world > 1) and the plain form; both print exactly one JSON line."""
    import json
    for cmd in ([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                 '--batch', '32', '--no-cpu-baseline', '--roofline-steps', '1'],
                [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                 '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
                 os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '32',
                 '--no-cpu-baseline', '--roofline-steps', '1']):
        r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, lines
        out = json.loads(lines[0])
        assert out['n_gpus'] == 1 and out['steps'] == 2 and out['value'] > 0 and 'roofline' in out


def test_bench_two_ranks_one_gpu_reports_its_ranks():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per process), here two ranks on
    the one GPU over gloo: the JSON line proves who took part (`dist.ranks`: device uuid / host / pid of EVERY rank,
    all-gathered), names the backend, counts the gradient buckets (--dp-buckets) and prices the part of the
    gradient all-reduce that backward does not hide (`allreduce_exposed_ms`); `value` is the whole-job rate."""
    import json
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'),
           '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '16', '--no-cpu-baseline',
           '--roofline-steps', '0', '--dp-buckets', '3']
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PASSL_DIST_BACKEND='gloo', PASSL_DEVICE_INDEX='0')
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:                    # keep the whole worker output (pytest truncates the assertion message)
        try:
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            with open(os.path.join(ROOT, 'gpurun_out', 'dp_fail_%s.log' % workload), 'w') as f:
                f.write(r.stdout + '\n---- stderr ----\n' + r.stderr)
        except OSError:
            pass
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 32 and out['value'] > 0
    d = out['dist']
    assert d['backend'] == 'gloo' and d['world'] == 2 and len(d['ranks']) == 2
    assert sorted(e['rank'] for e in d['ranks']) == [0, 1] and len({e['pid'] for e in d['ranks']}) == 2
    assert d['distinct_devices'] == 1                      # both ranks on the single GPU of the test box
    assert d['grad_buckets'] == 3 and d['grad_bytes'] > 100e6
    assert d['allreduce_exposed_ms'] is not None and d['allreduce_exposed_ms'] >= 0


def test_two_ranks_bf16_gradient_wire():
    """GradReducer's optional bf16 wire (PASSL_DP_WIRE=bf16: every bucket cast into a resident bf16 twin by the
    library's cast kernel right before its all-reduce, cast back after the final wait — half the bytes per link) on the
    real MoCo step, replayed from a step plan: the replicas stay bit-identical (the worker asserts it), the first loss
    is the fp32-wire run's (no gradient has been applied yet), later losses agree to the bf16 rounding of the averaged
    gradients, and the parameters are NOT the fp32-wire run's bit for bit (the wire really was bf16)."""
    env = dict(PASSL_DIST_BACKEND='gloo', PASSL_DEVICE_INDEX='0')
    f32 = _run_worker('moco', 2, env)
    b16 = _run_worker('moco', 2, dict(env, PASSL_DP_WIRE='bf16'))
    la, lb = ([float(v) for v in line.split('losses=')[1].split(',')] for line in (f32, b16))
    assert la[0] == lb[0], (f32, b16)
    assert all(abs(x - y) < 2e-2 for x, y in zip(la, lb)), (f32, b16)
    assert f32.split('digest=')[1].split()[0] != b16.split('digest=')[1].split()[0], (f32, b16)
