import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import qcnn_amd
from qcnn_amd.models import TimitQCNN
dev = torch.device('cuda:0')
dtype = torch.bfloat16
x = torch.randn(2, 41, 24, 4, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).to(dtype).permute(0, 3, 1, 2)
np.random.seed(3)
m = TimitQCNN(num_layers=2, start_filter=32, act='relu', aact='none', dropout=0.0, fuse_head=False)
# step through the forward by hand
with torch.no_grad():
    o = m.conv(x); torch.cuda.synchronize(); print('conv', tuple(o.shape), o.stride()); sys.stdout.flush()
    o = m.pool(o); torch.cuda.synchronize(); print('pool', tuple(o.shape), o.stride()); sys.stdout.flush()
    for c in m.convs:
        o = c(o); torch.cuda.synchronize(); print('conv', tuple(o.shape), o.stride()); sys.stdout.flush()
    o = o.permute(0, 3, 1, 2)
    o = o.reshape(o.shape[0], o.shape[1], o.shape[2] * o.shape[3]); torch.cuda.synchronize()
    print('reshaped', tuple(o.shape), o.stride(), o.dtype, o.data_ptr() % 16); sys.stdout.flush()
    o2 = o.reshape(-1, o.shape[-1])
    print('td in', tuple(o2.shape), o2.is_contiguous()); sys.stdout.flush()
    dl = m.dense[0].layer
    y = dl(o2); torch.cuda.synchronize(); print('dense0', tuple(y.shape), dl.r.shape, dl.r.dtype, dl.r.device, dl.r.data_ptr() % 16, dl.bias.data_ptr() % 16); sys.stdout.flush()
