"""Does running backward-weight and backward-data of one layer on TWO streams pay?  (both are power-bound; what could be won is
the kernel-to-kernel gap and the tail of each grid)  python tools/fork_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
for wl in ('cfg3_body_qconv2d_b256_bf16', 'cfg3_stage1_qconv2d_b256_bf16'):
    job = bench.LayerTrainStep(dict(bench.WORKLOADS[wl], activation='relu'), dev, 0, 1)
    job.x = torch.relu(job.x)
    job.dy = job.dy * (torch.rand_like(job.dy.float()) > 0.6).to(job.dy.dtype)
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    def seq():
        job.k_bwd_weight_chain(); job.k_bwd_data_chain()
    def fork():
        ev = torch.cuda.Event(); ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            job.k_bwd_weight_chain()
            ev2 = torch.cuda.Event(); ev2.record(side)
        job.k_bwd_data_chain()
        main.wait_event(ev2)
    for name, fn in (('sequential', seq), ('forked', fork), ('sequential', seq), ('forked', fork)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): fn()
        e1.record(); e1.synchronize()
        print('%-34s %-11s %.1f us per (bwd-weight + bwd-data) pair' % (wl, name, 1e3 * e0.elapsed_time(e1) / 40))
