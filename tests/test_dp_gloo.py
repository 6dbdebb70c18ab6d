"""CPU, world_size 2 over gloo: the data-parallel plumbing (qcnn_amd.dp) -- sharding by batch rows,
flat parameter/gradient buffers, one sum all-reduce -- reproduces the single-process gradient of
the concatenated batch.  Per-shard gradients come from the CPU oracle (the HIP kernels need a GPU;
their N>1 path differs only by the device the same buffers live on and the backend string)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    rng = np.random.RandomState(3)
    x = rng.randn(6, 11, 8).astype(np.float32)
    w = (0.3 * rng.randn(3, 2, 12)).astype(np.float32)
    b = (0.1 * rng.randn(12)).astype(np.float32)
    dy = rng.randn(6, 11, 12).astype(np.float32)
    return x, w, b, dy


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from qcnn_amd import dp
    from oracle import oracle
    r, ws, _ = dp.init_from_env(backend='gloo')
    assert (r, ws) == (rank, world) and dp.world_size() == world
    x, w, b, dy = _problem()
    # replicas: rank 1 starts from garbage and must receive rank 0's weights
    kernel = torch.nn.Parameter(torch.tensor(w if rank == 0 else w * 0 + 7))
    bias = torch.nn.Parameter(torch.tensor(b if rank == 0 else b * 0 - 3))
    flat = dp.FlatParams([kernel, bias])
    assert kernel.data_ptr() == flat.param.data_ptr() and flat.numel % 64 == 0
    dp.broadcast_params(flat)
    assert np.array_equal(kernel.detach().numpy(), w) and np.array_equal(bias.detach().numpy(), b)
    lo, hi = dp.shard_rows(x.shape[0], rank, world)
    kw = dict(padding='same', activation='relu')
    _, dw, db = oracle.backward(x[lo:hi], kernel.detach().numpy(), bias.detach().numpy(), dy[lo:hi], 1, **kw)
    flat.grad_view(0).copy_(torch.tensor(dw, dtype=torch.float32))
    flat.grad_view(1).copy_(torch.tensor(db, dtype=torch.float32))
    assert kernel.grad.data_ptr() == flat.grad.data_ptr()
    work = dp.allreduce_sum_(flat.grad, async_op=True)
    work.wait()
    np.save(os.path.join(out_dir, 'grad_%d.npy' % rank), flat.grad.numpy())
    np.save(os.path.join(out_dir, 'offs_%d.npy' % rank), np.array(flat.offsets))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_full_batch_gradient(tmp_path):
    from oracle import oracle
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    x, w, b, dy = _problem()
    _, dw, db = oracle.backward(x, w, b, dy, 1, padding='same', activation='relu')
    g0 = np.load(tmp_path / 'grad_0.npy')
    g1 = np.load(tmp_path / 'grad_1.npy')
    assert np.array_equal(g0, g1)                      # every replica holds the same reduced gradient
    offs = np.load(tmp_path / 'offs_0.npy')
    got_dw = g0[offs[0]:offs[0] + dw.size].reshape(dw.shape)
    got_db = g0[offs[1]:offs[1] + db.size]
    assert np.abs(got_dw - dw).max() <= 1e-5 * np.abs(dw).max()
    assert np.abs(got_db - db).max() <= 1e-5 * np.abs(db).max()


def test_shard_rows_partitions_the_batch():
    from qcnn_amd import dp
    for n, world in ((64, 8), (10, 4), (3, 8), (2048, 8)):
        spans = [dp.shard_rows(n, r, world) for r in range(world)]
        covered = [i for lo, hi in spans for i in range(lo, hi)]
        assert covered == list(range(n))
    assert dp.world_size() == 1 and dp.allreduce_sum_(torch.zeros(4)) is None


@pytest.mark.gpu
def test_bench_step_through_one_rank_rccl_group():
    """The 1-GPU test box cannot form a 2-rank RCCL group (one device), so the exact calls of the
    multi-GPU bench -- torchrun env, nccl process group with device_id, async all-reduce overlapping
    backward-data, barrier, MAX-reduce of the elapsed time -- run here with a one-rank communicator
    (QK_DP_FORCE_COLLECTIVES, qcnn_amd/dp.py).  The default workload is the full TIMIT QCNN step: its gradients go
    out as bucketed all-reduces launched from autograd hooks during the backward."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, QK_DP_FORCE_COLLECTIVES='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(root, 'bench.py'),
           '--gpus', '1', '--steps', '10', '--warmup', '2', '--no-cpu-baseline', '--no-kernel-timing']
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
    rec = json.loads(line)
    assert rec['n_gpus'] == 1 and rec['value'] > 0 and rec['config']['launch'] == 'eager'
    assert rec['config']['workload'] == 'cfg3_qcnn_timit_b256_bf16' and rec['config']['gemm_view']['allreduce_buckets'] >= 3
    assert rec['pre_warmup_steps'] == 2 and rec['steps'] == 10


def _bucket_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from qcnn_amd import dp
    dp.init_from_env(backend='gloo')
    torch.manual_seed(0)                                   # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(20, 300), torch.nn.ReLU(), torch.nn.Linear(300, 300), torch.nn.ReLU(),
                              torch.nn.Linear(300, 7))
    frozen = net[2].bias
    frozen.requires_grad_(False)                           # a parameter that never gets a gradient
    flat = dp.FlatParams(list(net.parameters()))
    dp.broadcast_params(flat)
    red = dp.BucketedAllReduce(flat, bucket_bytes=16 * 1024)
    assert red.active and len(red.buckets) >= 3 and red.buckets[0][0] == 0 and red.buckets[-1][1] == flat.numel
    g = torch.Generator().manual_seed(5)
    x = torch.randn(12, 20, generator=g)
    y = torch.randn(12, 7, generator=g)
    lo, hi = dp.shard_rows(12, rank, world)
    for step in range(2):                                  # twice: the counters must re-arm
        flat.zero_grad()
        loss = ((net(x[lo:hi]) - y[lo:hi]) ** 2).sum()
        loss.backward()                                    # hooks launch the buckets as their gradients complete
        order = list(red.launch_order)
        red.finish()
        assert sorted(order) == order[::-1] or len(order) <= 1 or order[0] > order[-1], order   # last layers first
    np.save(os.path.join(out_dir, 'bgrad_%d.npy' % rank), flat.grad.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_overlapping_backward_equals_full_batch_gradient(tmp_path):
    """dp.BucketedAllReduce: per-bucket asynchronous all-reduces fired from autograd hooks while the backward is
    still running; the reduced flat gradient must equal the single-process gradient of the whole batch."""
    world, port = 2, _free_port()
    mp.spawn(_bucket_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from qcnn_amd import dp
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(20, 300), torch.nn.ReLU(), torch.nn.Linear(300, 300), torch.nn.ReLU(),
                              torch.nn.Linear(300, 7))
    net[2].bias.requires_grad_(False)
    flat = dp.FlatParams(list(net.parameters()))
    g = torch.Generator().manual_seed(5)
    x = torch.randn(12, 20, generator=g)
    y = torch.randn(12, 7, generator=g)
    ((net(x) - y) ** 2).sum().backward()
    want = flat.grad.numpy()
    g0, g1 = np.load(tmp_path / 'bgrad_0.npy'), np.load(tmp_path / 'bgrad_1.npy')
    assert np.array_equal(g0, g1)
    assert np.abs(g0 - want).max() <= 1e-5 * np.abs(want).max()
    red = dp.BucketedAllReduce(flat)                       # no process group here: inert
    assert not red.active and red.finish() is None
