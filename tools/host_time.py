"""Host-side issue time of one bench step (diagnostic): the CPU runs ahead of the GPU, so timing the
issue loop without a final sync gives the Python + launch cost per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from qcnn_amd import dp
rank, world, local = dp.init_from_env()
dev = torch.device('cuda', local)
torch.cuda.set_device(dev)
job = bench.LayerTrainStep(dict(bench.WORKLOADS['cfg2_qconv1d_timit_b64_fp32'], activation='relu'), dev, 0, 1)
for _ in range(20):
    job.step()
torch.cuda.synchronize()
for n in (10, 20, 40):
    t0 = time.perf_counter()
    for _ in range(n):
        job.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('steps %3d: issue %.1f us/step, total %.1f us/step' % (n, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
# per-call breakdown
import torch.distributed as dist
def timeit(fn, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return 1e6 * (t1 - t0) / n
print('k_fwd issue %.1f us' % timeit(job.k_fwd))
print('k_bwd_weight issue %.1f us' % timeit(job.k_bwd_weight))
print('k_bwd_data issue %.1f us' % timeit(job.k_bwd_data))
print('adam issue %.1f us' % timeit(job._adam))
if dist.is_initialized():
    def ar():
        w = dp.allreduce_sum_(job.flat.grad, async_op=True); w.wait()
    print('allreduce+wait issue %.1f us' % timeit(ar))
