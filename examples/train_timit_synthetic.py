#!/usr/bin/env python
"""Training loop for the TIMIT quaternion CNN (models/interspeech_model.py:getTimitModel2D of the reference) on synthetic
features, single GPU or data-parallel over one node.  The reference ships the model builder but no caller; this is the loop its
`Model.compile(loss={'ctc': lambda y_true, y_pred: y_pred}, optimizer=Adam)` / `fit` would run, on the MI355X engine:

    python examples/train_timit_synthetic.py --steps 20                      # one GPU
    python examples/train_timit_synthetic.py --gpus 8 --steps 20             # one process per GPU, RCCL all-reduce over xGMI

What a user of the reference changes: `from models.interspeech_model import getTimitModel2D` becomes
`from qcnn_amd.models import getTimitModel2D` (same attribute bag `d`); the Keras fit loop becomes the few lines of `main`.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
import qcnn_amd  # noqa: E402
from qcnn_amd import dp, functional as F  # noqa: E402
from qcnn_amd.models import getTimitModel2D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32, help='utterances per GPU and step')
    ap.add_argument('--frames', type=int, default=200)
    ap.add_argument('--layers', type=int, default=10)
    ap.add_argument('--filters', type=int, default=32)
    ap.add_argument('--aact', default='none', choices=['none', 'prelu'])
    ap.add_argument('--dropout', type=float, default=0.3)
    ap.add_argument('--l2', type=float, default=1e-5)
    ap.add_argument('--lr', type=float, default=5e-4)
    args = ap.parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:          # started plainly: become the launcher of the ranks
        sys.exit(dp.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    rank, world, local = dp.init_from_env()
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    d = types.SimpleNamespace(num_layers=args.layers, start_filter=args.filters, act='relu', aact=args.aact, dropout=args.dropout,
                              l2=args.l2, model='quaternion', quat_init='quaternion')       # the reference's attribute bag
    np.random.seed(0)
    torch.manual_seed(rank)                                       # dropout masks differ per rank; the weights do not (numpy seed)
    model, _ = getTimitModel2D(d)
    model.train()
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    B, T = args.batch, args.frames
    x = torch.randn(B, 4, 41, T, device=dev, generator=gen).to(torch.bfloat16)          # channels_first quaternion features
    labels = torch.randint(0, 61, (B, 40), device=dev, generator=gen, dtype=torch.int32)
    input_length = torch.full((B, 1), T, dtype=torch.int32, device=dev)
    label_length = torch.randint(10, 41, (B, 1), device=dev, generator=gen, dtype=torch.int32)
    with torch.no_grad():
        model(x[:1])                                              # build-on-first-call, like Keras
    model.to(dev)
    flat = dp.FlatParams([p for p in model.parameters() if p.requires_grad], direct=True)
    dp.broadcast_params(flat)                                     # identical replicas
    reducer = dp.BucketedAllReduce(flat, bucket_bytes=2 << 20)    # gradients leave in 2 MB buckets while the backward runs
    decay = flat.l2_decay()                                       # the l2 kernel regularisers, folded into the Adam kernel
    m, v = torch.zeros_like(flat.param), torch.zeros_like(flat.param)
    for step in range(1, args.steps + 1):
        cost = model.ctc_loss(x, labels, input_length, label_length).mean()           # K.ctc_batch_cost, one HIP kernel
        cost.backward()
        reducer.finish()
        F.adam_step(flat.param, flat.grad, m, v, step, lr=args.lr, grad_scale=1.0 / world, zero_grad=True, decay=decay)
        if rank == 0 and (step == 1 or step % 5 == 0 or step == args.steps):
            print('step %3d  ctc cost %.4f' % (step, float(cost)))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
