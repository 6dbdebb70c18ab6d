#!/usr/bin/env python
"""DECODA theme-identification example on MI355X -- counterpart of the reference's working_example.py.

    python examples/working_example.py --model QCNN --decoda /path/to/decoda

Same recipe as working_example.py:106-136: Adam(lr=0.0005) (the reference parses --lr and then
ignores it, :106-107), categorical cross-entropy, 15 epochs, batch size 3, evaluation on TEST.
The reference checkout ships DEV and TEST only (250_TRAIN_Q.data is missing); without a TRAIN file
the script trains on DEV so that the plumbing is exercised end to end.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
import qcnn_amd  # noqa: E402
from qcnn_amd.data import dataPrepDecodaQuaternion  # noqa: E402
from qcnn_amd.models import CNN, DNN  # noqa: E402


def getArgParser():
    parser = argparse.ArgumentParser(description='Parameters for the Neural Networks')
    parser.add_argument('--lr', default='0.001', type=float)
    parser.add_argument('--model', '--m', default='QCNN', type=str, choices=['QCNN', 'QDNN', 'CNN', 'DNN'])
    parser.add_argument('--decoda', default='decoda', help='directory holding 250_{TRAIN,DEV,TEST}_Q.data')
    parser.add_argument('--epochs', default=15, type=int)
    parser.add_argument('--batch-size', default=3, type=int)
    return parser.parse_args()


def main():
    params = getArgParser()
    dev = torch.device('cuda:0')
    quat = params.model in ('QCNN', 'QDNN')
    path = lambda split: os.path.join(params.decoda, '250_%s_Q.data' % split)
    x_dev, y_dev = dataPrepDecodaQuaternion(path('DEV'), isquat=quat)
    x_test, y_test = dataPrepDecodaQuaternion(path('TEST'), isquat=quat)
    if os.path.exists(path('TRAIN')):
        x_train, y_train = dataPrepDecodaQuaternion(path('TRAIN'), isquat=quat)
    else:
        print('250_TRAIN_Q.data not found: training on DEV (plumbing run)')
        x_train, y_train = x_dev, y_dev
    print('Train size : %d\nDev size   : %d\nTest size  : %d' % (len(x_train), len(x_dev), len(x_test)))
    np.random.seed(0)
    torch.manual_seed(0)
    model = CNN(params) if params.model in ('CNN', 'QCNN') else DNN(params)
    to_t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    xt, yt = to_t(x_train), to_t(y_train)
    model(xt[:2])                                        # build (Keras builds on first call)
    model.to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.0005, eps=1e-7)

    def evaluate(x, y):
        model.eval()
        with torch.no_grad():
            p = torch.cat([model(to_t(x[i:i + 64])) for i in range(0, len(x), 64)])
        yy = to_t(y)
        loss = -(yy * torch.log(p.clamp_min(1e-7))).sum(1).mean()
        acc = (p.argmax(1) == yy.argmax(1)).float().mean()
        return float(loss), float(acc)

    for epoch in range(params.epochs):
        model.train()
        perm = torch.randperm(len(xt), device=dev)
        for i in range(0, len(xt), params.batch_size):
            idx = perm[i:i + params.batch_size]
            p = model(xt[idx])
            loss = -(yt[idx] * torch.log(p.clamp_min(1e-7))).sum(1).mean()     # categorical_crossentropy
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        print('epoch %2d  dev loss %.4f acc %.4f' % ((epoch + 1,) + evaluate(x_dev, y_dev)))
    print('Test Loss = %s | Test accuracy = %s' % evaluate(x_test, y_test))


if __name__ == '__main__':
    main()
