"""DECODA example networks -- counterpart of models/example_model.py (CNN :15-48, DNN :54-81).

`CNN(params)` / `DNN(params)` take the reference's `params` bag (`params.model` in
{'QCNN','CNN'} / {'QDNN','DNN'}) and return a torch module with the same topology, quirks
included: the QCNN/QDNN heads see `Flatten()` of component-interleaved data (example_model.py:
29-32,69-73) and the QDNN's second/third layers consume h0/h1, not the dropout outputs (:74-77).
"""
import torch

from ..complexnn import QuaternionConv1D, QuaternionDense
from ..layers import AveragePooling1D, Dense, Dropout, Flatten


class _Sequential(torch.nn.Module):
    def __init__(self, *layers):
        super(_Sequential, self).__init__()
        self.layers = torch.nn.ModuleList(layers)

    def forward(self, x):
        for l in self.layers:
            x = l(x)
        return x


class _RealConv1D(torch.nn.Module):
    """keras Conv1D(filters, k, padding='same', activation='relu') on (B, T, C)."""

    def __init__(self, filters, kernel_size):
        super(_RealConv1D, self).__init__()
        self.filters, self.k, self.conv = filters, kernel_size, None

    def forward(self, x):
        if self.conv is None:
            self.conv = torch.nn.Conv1d(x.shape[-1], self.filters, self.k, padding=self.k // 2).to(x.device, x.dtype)
        return torch.relu(self.conv(x.transpose(1, 2))).transpose(1, 2)


def CNN(params):
    if params.model == 'QCNN':        # input (B, 250, 4)
        return _Sequential(
            QuaternionConv1D(32, 3, strides=1, activation='relu', padding='same'),
            AveragePooling1D(2, padding='same'),
            QuaternionConv1D(64, 3, strides=1, activation='relu', padding='same'),
            AveragePooling1D(4, padding='same'),
            Flatten(),
            QuaternionDense(256, activation='relu'),
            Dense(8, activation='softmax'))
    return _Sequential(                # input (B, 250, 3)
        _RealConv1D(32, 3), AveragePooling1D(2, padding='same'),
        _RealConv1D(64, 3), AveragePooling1D(4, padding='same'),
        Flatten(), Dense(256, activation='relu'), Dense(8, activation='softmax'))


class _QDNN(torch.nn.Module):
    def __init__(self):
        super(_QDNN, self).__init__()
        self.flat = Flatten()
        self.h0 = QuaternionDense(512, activation='relu')
        self.h1 = QuaternionDense(512, activation='relu')
        self.h2 = QuaternionDense(512, activation='relu')
        self.d0, self.d1 = Dropout(0.3), Dropout(0.3)      # created, outputs unused (example_model.py:74-77)
        self.out = Dense(8, activation='softmax')

    def forward(self, x):
        h0 = self.h0(self.flat(x))
        h1 = self.h1(h0)
        h2 = self.h2(h1)
        return self.out(h2)


def DNN(params):
    if params.model == 'QDNN':         # input (B, 250, 4)
        return _QDNN()
    return _Sequential(Flatten(), Dense(512, activation='relu'), Dropout(0.3),
                       Dense(512, activation='relu'), Dropout(0.3), Dense(512, activation='relu'),
                       Dense(8, activation='softmax'))
