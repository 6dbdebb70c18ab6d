"""Launch one hot kernel of a bench workload a few times (for rocprofv3 runs).
   python tools/run_kernel.py <workload> <fwd|bwd_data|bwd_weight|bwd_data_chain|bwd_weight_chain|step> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
wl, which = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device('cuda:0')
job = bench.LayerTrainStep(dict(bench.WORKLOADS[wl], activation='relu'), dev, 0, 1)
fn = {'fwd': job.k_fwd, 'bwd_data': job.k_bwd_data, 'bwd_weight': job.k_bwd_weight, 'step': job.step,
      'bwd_weight_chain': job.k_bwd_weight_chain, 'bwd_data_chain': job.k_bwd_data_chain}[which]
job.k_fwd()
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print('done', wl, which, reps)
