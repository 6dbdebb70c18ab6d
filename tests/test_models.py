"""Callers of the hot path: stock-layer semantics on CPU, and the example
networks on the GPU against an oracle-composed forward."""
import types

import numpy as np
import pytest
import torch

import qcnn_amd
from qcnn_amd.layers import AveragePooling1D, Dense, Flatten, MaxPooling2D, PReLU, TimeDistributed, ctc_batch_cost


def test_average_pooling_same_excludes_padding_from_divisor():
    x = torch.arange(2 * 7 * 3.).reshape(2, 7, 3)
    y = AveragePooling1D(2, padding='same')(x)
    assert tuple(y.shape) == (2, 4, 3)
    assert torch.allclose(y[0, :, 0], torch.tensor([1.5, 7.5, 13.5, 18.0]))
    y4 = AveragePooling1D(4, padding='same')(torch.ones(1, 125, 2))       # QCNN: 125 -> 32 steps
    assert tuple(y4.shape) == (1, 32, 2) and torch.allclose(y4, torch.ones_like(y4))


def test_maxpool_default_format_pools_the_frequency_axis():
    x = torch.randn(2, 8, 41, 5)
    y = MaxPooling2D(pool_size=(1, 3), padding='same')(x)
    assert tuple(y.shape) == (2, 8, 14, 5)
    assert torch.equal(y[:, :, 0, :], x[:, :, 0:3, :].amax(2))
    assert torch.equal(y[:, :, 13, :], x[:, :, 39:41, :].amax(2))           # last window: 2 real cells


@pytest.mark.parametrize('shape,pool,strides,axes', [
    ((2, 8, 41, 20), (1, 3), (1, 3), (1, 2)),        # the TIMIT model's frequency pooling
    ((2, 5, 10, 11), (2, 2), (2, 2), (2, 3)),
    ((2, 6, 9, 7), (2, 2), (2, 2), (1, 2)),          # pools the channel position too: generic path
    ((2, 5, 9, 7), (3, 2), (2, 2), (2, 3)),          # low-side padding: generic path
])
@pytest.mark.parametrize('mode', ['max', 'avg'])
def test_in_place_pooling_of_channels_last_buffers_matches_generic_path(shape, pool, strides, axes, mode):
    from qcnn_amd.layers import _pool_nd
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(5))
    fast = _pool_nd(x.contiguous(memory_format=torch.channels_last), pool, strides, 'same', axes, mode)
    generic = _pool_nd(x.unsqueeze(-1), pool, strides, 'same', axes, mode).squeeze(-1)   # 5-D: no fast path
    assert fast.shape == generic.shape
    assert float((fast - generic).abs().max()) <= 1e-6


def test_dense_prelu_timedistributed_ctc_on_cpu():
    np.random.seed(0)
    d = TimeDistributed(Dense(5, activation='softmax'))
    y = d(torch.randn(2, 7, 3))
    assert tuple(y.shape) == (2, 7, 5) and torch.allclose(y.sum(-1), torch.ones(2, 7), atol=1e-6)
    p = PReLU(shared_axes=[1, 0])
    z = p(torch.tensor([[[-2.0, 3.0]]]))
    assert tuple(p.alpha.shape) == (1, 1) and torch.equal(z, torch.tensor([[[0.0, 3.0]]]))   # Keras: axis 0 -> index -1
    cost = ctc_batch_cost(y, torch.tensor([[1, 2], [3, 0]]), torch.tensor([[7], [7]]), torch.tensor([[2], [1]]))
    assert tuple(cost.shape) == (2, 1) and torch.isfinite(cost).all() and (cost > 0).all()
    assert tuple(Flatten()(torch.zeros(3, 4, 5)).shape) == (3, 20)


def test_decoda_reader_matches_survey_facts(tmp_path):
    line = ' '.join('0,%g,%g,%g' % (0.01 * i, 0.02, 0.03) for i in range(250)) + '\t' + \
           ' '.join('%d,%d,%d,%d' % ((1,) * 4 if k == 3 else (0,) * 4) for k in range(8)) + '\n'
    f = tmp_path / 'x.data'
    f.write_text(line * 2)
    x, y = qcnn_amd.data.dataPrepDecodaQuaternion(str(f), isquat=True)
    assert x.shape == (2, 250, 4) and y.shape == (2, 8) and x.dtype == np.float64
    assert x[0, 5, 1] == pytest.approx(0.05) and x[0, :, 0].max() == 0 and y[1].tolist() == [0, 0, 0, 1, 0, 0, 0, 0]
    x3, _ = qcnn_amd.data.dataPrepDecodaQuaternion(str(f), isquat=False)
    assert x3.shape == (2, 250, 3) and x3[0, 5, 0] == pytest.approx(0.05)


def _np_avgpool_same(x, k):
    t = x.shape[1]
    out = -(-t // k)
    total = max((out - 1) * k + k - t, 0)
    lo = total // 2
    y = np.zeros((x.shape[0], out, x.shape[2]))
    for o in range(out):
        a, b = max(o * k - lo, 0), min(o * k - lo + k, t)
        y[:, o] = x[:, a:b].mean(1)
    return y


@pytest.mark.gpu
def test_example_networks_match_oracle_composition():
    from oracle import oracle
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(0)
    x = rng.rand(6, 250, 4).astype(np.float32) * 0.8
    np.random.seed(1)
    qcnn = qcnn_amd.models.CNN(types.SimpleNamespace(model='QCNN'))
    xt = torch.tensor(x, device=dev)
    p = qcnn(xt)
    L = qcnn.layers
    g = lambda t: t.detach().cpu().numpy().astype(np.float64)
    h = oracle.forward(x, g(L[0].kernel), g(L[0].bias), 1, padding='same', activation='relu')
    h = _np_avgpool_same(h, 2)
    h = oracle.forward(h, g(L[2].kernel), g(L[2].bias), 1, padding='same', activation='relu')
    h = _np_avgpool_same(h, 4).reshape(6, -1)
    h = oracle.forward(h, g(L[5].r), g(L[5].bias), 0, activation='relu')
    z = h @ g(L[6].kernel) + g(L[6].bias)
    want = np.exp(z - z.max(1, keepdims=True))
    want /= want.sum(1, keepdims=True)
    assert tuple(p.shape) == (6, 8)
    assert np.abs(g(p) - want).max() <= 1e-4
    np.random.seed(2)
    qdnn = qcnn_amd.models.DNN(types.SimpleNamespace(model='QDNN'))
    q = qdnn(xt)
    h = x.reshape(6, -1)
    for lyr in (qdnn.h0, qdnn.h1, qdnn.h2):
        h = oracle.forward(h, g(lyr.r), g(lyr.bias), 0, activation='relu')
    z = h @ g(qdnn.out.kernel) + g(qdnn.out.bias)
    want = np.exp(z - z.max(1, keepdims=True))
    want /= want.sum(1, keepdims=True)
    assert np.abs(g(q) - want).max() <= 1e-4
    q.sum().backward()                                    # gradients reach every quaternion kernel
    assert all(l.r.grad is not None and torch.isfinite(l.r.grad).all() for l in (qdnn.h0, qdnn.h1, qdnn.h2))


@pytest.mark.gpu
def test_timit_qcnn_shapes_and_backward():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    dev = torch.device('cuda:0')
    np.random.seed(0)
    d = types.SimpleNamespace(num_layers=4, start_filter=8, act='relu', aact='prelu', dropout=0.0, l2=1e-4,
                              model='quaternion', quat_init='quaternion')
    model, val = qcnn_amd.models.getTimitModel2D(d)
    x = torch.randn(2, 4, 41, 20, device=dev)
    pred = model(x)
    assert tuple(pred.shape) == (2, 20, 62) and torch.allclose(pred.sum(-1), torch.ones(2, 20, device=dev), atol=1e-4)
    assert tuple(model.conv.kernel.shape) == (3, 5, 1, 32) and tuple(model.convs[2].kernel.shape) == (3, 5, 8, 64)
    assert tuple(model.dense[0].layer.r.shape) == (14 * 4 * 16 // 4, 256)
    labels = torch.randint(0, 61, (2, 5), device=dev)
    cost = model.ctc_loss(x, labels, torch.full((2, 1), 20, device=dev), torch.full((2, 1), 5, device=dev))
    reg = sum(sum(m.regularization_losses()) for m in model.modules() if hasattr(m, 'regularization_losses'))
    (cost.mean() + reg).backward()
    assert torch.isfinite(model.conv.kernel.grad).all() and torch.isfinite(model.dense[2].layer.r.grad).all()
    assert torch.equal(val(x), model(x))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)], ids=['fp32', 'bf16'])
def test_timit_head_as_convolution_equals_permute_reshape_dense(dtype, tol):
    """TimitQCNN(fuse_head=True) computes Permute + reshape + TimeDistributed(QuaternionDense) as an (F, 1)
    conj-convolution on the conv output; outputs and every gradient must match the literal path."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    dev = torch.device('cuda:0')
    from qcnn_amd.models import TimitQCNN
    x = torch.randn(2, 41, 24, 4, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).to(dtype).permute(0, 3, 1, 2)
    np.random.seed(3)
    ref = TimitQCNN(num_layers=2, start_filter=32, act='relu', aact='none', dropout=0.0, fuse_head=False)
    fused = TimitQCNN(num_layers=2, start_filter=32, act='relu', aact='none', dropout=0.0, fuse_head=True)
    with torch.no_grad():
        ref(x), fused(x)                                   # build both, then give them the same weights
    fused.load_state_dict(ref.state_dict())
    outs = []
    for m in (ref, fused):
        y = m(x)
        (y.float() * torch.linspace(0.5, 1.5, 62, device=dev)).sum().backward()
        outs.append((y.detach().float(), m.dense[0].layer.r.grad.clone(), m.dense[0].layer.bias.grad.clone(),
                     m.convs[1].kernel.grad.clone(), m.conv.kernel.grad.clone()))
    for a, b in zip(*outs):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= tol * max(float(a.abs().max()), 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16], ids=['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('shape,pool,padding,axes', [
    ((3, 128, 41, 20), (1, 3), 'same', (1, 2)),       # the TIMIT model's frequency pooling (partial last window)
    ((2, 16, 41, 20), (1, 3), 'valid', (1, 2)),       # trailing rows outside every window: zero gradient
    ((2, 8, 12, 10), (2, 2), 'same', (2, 3)),
    ((2, 24, 9, 7), (2, 3), 'same', (2, 3)),
])
def test_engine_maxpool_matches_torch_including_ties(shape, pool, padding, axes, dtype):
    """MaxPooling2D on a channels-last device buffer runs on qk_maxpool2d_*; it must agree with torch's
    pooling exactly, ties included (relu outputs are full of equal zeros; the first maximum gets the
    gradient)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    dev = torch.device('cuda:0')
    from qcnn_amd.layers import _pool_nd
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.relu(torch.randn(*shape, device=dev, generator=g)).to(dtype)        # many exact ties at 0
    x = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = _pool_nd(x, pool, pool, padding, axes, 'max')
    dy = torch.randn(y.shape, device=dev, generator=g).to(dtype)
    y.backward(dy)
    xr = x.detach().clone().unsqueeze(-1).requires_grad_(True)                      # 5-D: torch's generic path
    yr = _pool_nd(xr, pool, pool, padding, axes, 'max').squeeze(-1)
    yr.backward(dy)
    assert y.shape == yr.shape and torch.equal(y, yr)
    assert torch.equal(x.grad, xr.grad.squeeze(-1))


@pytest.mark.gpu
def test_example_qcnn_learns_a_toy_problem():
    """End to end through the C-ABI kernels: the example QCNN (models/example_model.py:15-40 counterpart)
    trained with Adam(5e-4-style) on a separable toy problem -- the cross-entropy must drop markedly."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    dev = torch.device('cuda:0')
    np.random.seed(4)
    torch.manual_seed(4)
    net = qcnn_amd.models.CNN(types.SimpleNamespace(model='QCNN'))
    g = torch.Generator(device=dev).manual_seed(4)
    labels = torch.randint(0, 8, (64,), device=dev, generator=g)
    x = torch.rand(64, 250, 4, device=dev, generator=g) * 0.2
    x[torch.arange(64, device=dev), labels * 30, :] += 2.0          # the class moves one "topic" spike
    net(x[:2])                                                         # build
    opt = torch.optim.Adam(net.parameters(), lr=2e-3)
    losses = []
    for _ in range(60):
        opt.zero_grad()
        p = net(x)
        loss = -torch.log(p[torch.arange(64, device=dev), labels].clamp_min(1e-7)).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,tol', [(torch.float32, 1e-5), (torch.bfloat16, 3e-2)], ids=['fp32', 'bf16'])
def test_conv_chain_node_equals_layer_by_layer(dtype, tol):
    """TimitQCNN(chain_convs=True) runs its body convolutions as one autograd node whose backward moves
    each relu derivative into the next layer's backward-data epilogue (qk_conv_bwd_chain flags); values
    and every gradient must match the layer-by-layer model."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    dev = torch.device('cuda:0')
    from qcnn_amd.models import TimitQCNN
    x = torch.randn(2, 41, 40, 4, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).to(dtype).permute(0, 3, 1, 2)
    np.random.seed(5)
    ref = TimitQCNN(num_layers=4, start_filter=32, act='relu', aact='none', dropout=0.0, chain_convs=False)
    new = TimitQCNN(num_layers=4, start_filter=32, act='relu', aact='none', dropout=0.0, chain_convs=True)
    with torch.no_grad():
        ref(x), new(x)
    new.load_state_dict(ref.state_dict())
    outs = []
    for m in (ref, new):
        y = m(x)
        (y.float() * torch.linspace(0.5, 1.5, 62, device=dev)).sum().backward()
        outs.append([y.detach().float()] + [c.kernel.grad.clone() for c in m.convs] + [c.bias.grad.clone() for c in m.convs]
                    + [m.conv.kernel.grad.clone()])
    for a, b in zip(*outs):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= tol * max(float(a.abs().max()), 1e-6)


# ---- G13: pinned to the reference's own example-model builders and DECODA reader -----------------------------------
import json
import os

_G13 = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g13_example_nets.npz')
_DEV8 = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'decoda_dev_head8.data')


def test_decoda_reader_reproduces_what_the_reference_reader_parsed():
    """qcnn_amd.data on the first 8 lines of the bundled DEV file (tests/golden/decoda_dev_head8.data) against the
    arrays working_example.py:dataPrepDecodaQuaternion (:19-66) produced for them (fixture g13)."""
    z = np.load(_G13)
    x, y = qcnn_amd.data.dataPrepDecodaQuaternion(_DEV8, isquat=True)
    assert x.dtype == np.float64 and x.shape == (8, 250, 4) and y.shape == (8, 8)
    assert np.array_equal(x.astype(np.float32), z['x']) and np.array_equal(y.astype(np.float32), z['labels'])
    assert z['dev_shape'].tolist() == [174, 250, 4] and z['dev_label_counts'].tolist() == [4, 43, 9, 45, 33, 24, 7, 9]
    x3, _ = qcnn_amd.data.dataPrepDecodaQuaternion(_DEV8, isquat=False)
    assert np.array_equal(x3, x[:, :, 1:])


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['qcnn', 'qdnn'])
def test_example_networks_match_the_reference_model_fixture(tag):
    """models/example_model.py:CNN('QCNN') / DNN('QDNN') of the reference, executed through the keras stand-in on
    the first 8 DEV documents (fixture g13): class posteriors and the gradient of sum(p * dp) w.r.t. every weight."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    dev = torch.device('cuda:0')
    z = np.load(_G13)
    names = json.loads(str(z['config']))[tag]
    np.random.seed(0)
    net = (qcnn_amd.models.CNN if tag == 'qcnn' else qcnn_amd.models.DNN)(types.SimpleNamespace(model=tag.upper()))
    x = torch.tensor(z['x'], device=dev)
    with torch.no_grad():
        net(x)
    if tag == 'qcnn':
        L = net.layers
        params = [L[0].kernel, L[0].bias, L[2].kernel, L[2].bias, L[5].r, L[5].bias, L[6].kernel, L[6].bias]
    else:
        params = [net.h0.r, net.h0.bias, net.h1.r, net.h1.bias, net.h2.r, net.h2.bias, net.out.kernel, net.out.bias]
    assert len(params) == len(names)
    with torch.no_grad():
        for i, (p, (_, wname, shape)) in enumerate(zip(params, names)):
            assert list(p.shape) == shape, (i, wname, tuple(p.shape), shape)
            p.copy_(torch.tensor(z['%s_w%03d' % (tag, i)], device=dev))
    p = net(x)
    want = z[tag + '_p']
    assert float(np.abs(p.detach().cpu().numpy() - want).max()) <= 1e-4 * float(np.abs(want).max())
    (p.double() * torch.tensor(z[tag + '_dp'], device=dev).double()).sum().backward()
    for i, prm in enumerate(params):
        g, w = prm.grad.cpu().numpy().astype(np.float64), z['%s_g%03d' % (tag, i)].astype(np.float64)
        assert float(np.abs(g - w).max()) <= 1e-4 * float(np.abs(w).max()), (i, names[i])


@pytest.mark.gpu
@pytest.mark.parametrize('aact,dropout,l2,loss_kind', [('none', 0.0, 0.0, 'sum'), ('prelu', 0.2, 0.0, 'sum'), ('none', 0.25, 1e-4, 'sum'),
                                                         ('none', 0.25, 1e-4, 'ctc')],
                         ids=['relu', 'prelu_dropout', 'relu_dropout_l2', 'relu_dropout_l2_ctc'])
def test_bench_model_step_trains_through_the_flat_buffers(aact, dropout, l2, loss_kind):
    """bench.ModelTrainStep (what the driver times): autograd accumulates into the views of dp.FlatParams' gradient
    buffer (never re-homing .grad), the fused Adam consumes and zeroes it, the bucketed reducer is inert without a
    process group -- and the synthetic loss goes down."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import bench
    dev = torch.device('cuda:0')
    cfg = dict(kind='model', batch=4, frames=64 if loss_kind == 'ctc' else 24, sf=32, layers=2, dtype='bf16', aact=aact, dropout=dropout,
               l2=l2, activation='relu')
    job = bench.ModelTrainStep(cfg, dev, 0, 1, loss=loss_kind)
    lo, hi = job.flat.grad.data_ptr(), job.flat.grad.data_ptr() + job.flat.grad.numel() * 4
    assert (job.decay is not None) == (l2 > 0)
    assert all(getattr(p, '_qk_direct_grad', False) for p in job.flat.params)

    def loss():
        with torch.no_grad():
            job.model.eval()
            if loss_kind == 'ctc':
                v = float(job.model.ctc_loss(job.x, job.labels, job.input_length, job.label_length).mean())
            else:
                v = float((job.model(job.x).float() * job.target).sum())
            job.model.train()
            return v
    l0 = loss()
    for _ in range(25):
        job.step()
    torch.cuda.synchronize()
    assert all(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in job.flat.params)
    assert float(job.flat.grad.abs().max()) == 0.0                    # Adam left the buffer zeroed
    assert all(torch.isfinite(p).all() for p in job.flat.params)
    assert loss() < l0 - 1e-3 * abs(l0), (l0, loss())


@pytest.mark.gpu
def test_tall_dense_split_reduction_matches_plain_autograd():
    """layers.Dense on many 16-bit rows (the Dense(62) behind the TIMIT head) takes _TallDenseFn: same output as the
    plain matmul, gradients equal to an fp32 autograd reference (the kernel gradient is a 32-way split reduction with
    fp32 partial sums)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from qcnn_amd.layers import Dense
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    x = torch.randn(64, 200, 256, device=dev).to(torch.bfloat16).requires_grad_(True)
    d = Dense(62)
    y = d(x)
    d.to(dev)
    y = d(x)
    dy = torch.randn_like(y)
    y.backward(dy)
    xr = x.detach().float().requires_grad_(True)
    wr = d.kernel.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = d.bias.detach().clone().requires_grad_(True)
    yr = xr @ wr + br
    yr.backward(dy.float())
    rel = lambda a, b: float((a.float() - b).abs().max() / b.abs().max())
    assert rel(y.detach(), yr.detach()) <= 1e-2 and rel(x.grad, xr.grad) <= 1e-2
    assert d.kernel.grad.dtype == torch.float32 and rel(d.kernel.grad, wr.grad) <= 1e-4
    assert rel(d.bias.grad, br.grad) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
@pytest.mark.parametrize('shape', [(5, 30, 10, 6), (3, 27, 62, 9), (256, 200, 62, 50), (2, 300, 62, 20), (3, 160, 20, 70), (3, 40, 200, 8),
                                   (2, 70, 256, 9)],
                         ids=['small', 'odd', 'timit_b256', 'long_lp_from_global', 'more_than_63_labels', 'more_than_128_classes',
                              'classes_256'])
def test_fused_ctc_batch_cost_matches_the_keras_restatement(shape, dtype):
    """qk_ctc_batch_cost (K.ctc_batch_cost of interspeech_model.py:37-39 as one launch: cost and d cost / d y_pred) against the
    torch restatement of the Keras / TensorFlow op that the golden fixture G17 pins (layers.ctc_batch_cost with the fused path
    off): ragged input and label lengths, repeated labels, an empty label sequence, frames past the input length."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    dev = torch.device('cuda:0')
    b, t, c, lmax = shape
    g = torch.Generator().manual_seed(b + t)
    logits = torch.randn(b, t, c, generator=g) * 2.0
    pred = torch.softmax(logits, -1).to(dev).to(dtype)
    labels = torch.randint(0, c - 1, (b, lmax), generator=g)
    labels[0, 1] = labels[0, 0]                                        # a repeated label: needs a blank between the two
    ll = torch.randint(1, lmax + 1, (b, 1), generator=g)
    ll[1, 0] = 0 if b > 1 else ll[0, 0]                                # an empty label sequence
    il = torch.randint(max(2 * lmax + 2, t // 2), t + 1, (b, 1), generator=g).clamp(max=t)
    w = torch.rand(b, 1, generator=g).to(dev) + 0.5                    # upstream gradient of the cost
    from qcnn_amd import _lib
    outs = {}
    for fused in (True, 'two_sweeps', False):
        with _lib.debug_flags(_lib.QK_DBG_CTC_TWO_SWEEPS if fused == 'two_sweeps' else 0 if fused else _lib.QK_DBG_NO_FUSED_CTC):
            p = pred.clone().requires_grad_(True)
            cost = ctc_batch_cost(p, labels.to(dev), il, ll)
            (cost * w).sum().backward()
            torch.cuda.synchronize()
        outs[fused] = (cost.detach().double().cpu(), p.grad.detach().double().cpu())
    (ct, gt) = outs[False]
    for form in (True, 'two_sweeps'):       # the concurrent-sweep kernel (round 4; long utterances fall back) and the round-3 form
        cf, gf = outs[form]
        assert tuple(cf.shape) == (b, 1) and torch.isfinite(cf).all()
        assert float((cf - ct).abs().max() / ct.abs().max()) <= 1e-5, form              # same 16-bit inputs, fp32 math
        tol = 2e-4 if dtype == torch.float32 else 1e-2                                    # bf16: the gradient is stored in bf16
        assert float((gf - gt).abs().max() / gt.abs().max()) <= tol, form
    gf = outs[True][1]
    for i in range(b):                                                 # frames past the input length get no gradient
        assert float(gf[i, int(il[i]):].abs().sum()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
@pytest.mark.parametrize('rows,kin,units', [(8192, 256, 62), (51200, 256, 62), (1000, 128, 40), (37, 64, 2), (4099, 256, 64)],
                         ids=['8192x256x62', 'timit_b256', 'ragged_1000x128x40', 'tiny_37x64x2', 'ragged_4099x256x64'])
def test_fused_dense_softmax_output_layer_matches_float64(dtype, rows, kin, units):
    """TimeDistributed(Dense(62, activation='softmax')) (interspeech_model.py:171-175) through layers._DenseSoftmaxFn -- library
    GEMMs + qk_softmax_rows_fwd / _bwd -- against a float64 restatement on the same 16-bit operands: posteriors, d input,
    d kernel, d bias; with the bench's weighted-sum loss (qk_weighted_sum) on top.  Also against the unfused torch path."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import qcnn_amd
    from qcnn_amd.layers import Dense
    Fq = qcnn_amd.functional
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(8)
    x = torch.randn(rows, kin, generator=g).to(dtype)
    tgt = torch.randn(rows, units, generator=g)
    np.random.seed(3)
    layer = Dense(units, activation='softmax', kernel_initializer='random_uniform')
    layer._build_device = dev
    layer.build((None, kin))
    with torch.no_grad():
        layer.kernel.mul_(20.0)                   # logits of order 1: a softmax that is not flat
        layer.bias.copy_(torch.randn(units, generator=g).to(dev) * 0.5)
    outs = {}
    from qcnn_amd import _lib
    for fused in (True, False):
        with _lib.debug_flags(0 if fused else _lib.QK_DBG_NO_FUSED_SOFTMAX):
            xd = x.to(dev).requires_grad_(True)
            layer.zero_grad()
            y = layer(xd)
            loss = Fq.weighted_sum(y, tgt.to(dev)) if fused else (y.float() * tgt.to(dev)).sum()
            loss.backward()
            outs[fused] = [t.detach().double().cpu() for t in (y, loss, xd.grad, layer.kernel.grad, layer.bias.grad)]
    w64 = layer.kernel.detach().to(dtype).double().cpu().requires_grad_(True)      # the GEMMs multiply the 16-bit image of the kernel
    b64 = layer.bias.detach().double().cpu().requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    y64 = torch.softmax(x64 @ w64 + b64, -1)
    l64 = (y64 * tgt.double()).sum()
    l64.backward()
    want = [y64.detach(), l64.detach(), x64.grad, w64.grad, b64.grad]
    # (the loss is a sum with cancellation: its error against float64 is the posteriors' rounding walked over rows x units random
    #  signs -- checked against the sum of the kernel's OWN posteriors, which is what qk_weighted_sum computes; y has its own line)
    want[1] = (outs[True][0] * tgt.double()).sum()
    tol = dict(bfloat16=(8e-3, 2e-3, 2e-2, 1e-2, 1e-2), float16=(1e-3, 5e-4, 4e-3, 2e-3, 2e-3))[str(dtype).split('.')[-1]]
    assert Fq.dense_softmax_supported(x.to(dev), units)             # the hand-written kernels take every one of these shapes (round 6)
    for name, got, ref, t in zip(('y', 'loss', 'dx', 'dkernel', 'dbias'), outs[True], want, tol):
        err = float((got - ref).abs().max() / ref.abs().max())
        if rows < 1000 and name in ('dkernel', 'dbias'):
            t *= 3.0            # sums over a few dozen rows: the 16-bit roundings of y and d logits do not average out (same for the composition)
        assert err <= t, '%s: %.3g > %.1g' % (name, err, t)
    # the fused path is at least as close to float64 as the unfused one on the posteriors (fp32 logits instead of 16-bit ones)
    e_f = float((outs[True][0] - want[0]).abs().max())
    e_u = float((outs[False][0] - want[0]).abs().max())
    assert e_f <= e_u * 1.05 + 1e-6, (e_f, e_u)
    assert abs(float(outs[True][0].sum(-1).mean()) - 1.0) < 2e-3


@pytest.mark.gpu
def test_fused_ctc_degenerate_samples():
    """Samples TensorFlow's ctc_loss raises for (round-3 advisor): a label sequence that does not fit its frames (with the
    blank a repeated label needs) gets cost +inf and NO gradient -- not a finite softmax / (p + eps) row that would steer
    the optimiser; zero frames with labels likewise; zero frames and zero labels cost 0.  The feasible samples of the same
    batch are unaffected."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    b, t, c = 5, 12, 9
    pred = torch.softmax(torch.randn(b, t, c, generator=g), -1).to(dev)
    labels = torch.tensor([[1, 2, 3, 4], [2, 2, 2, 0], [1, 2, 3, 4], [5, 6, 0, 0], [1, 1, 0, 0]])
    ll = torch.tensor([[4], [3], [4], [2], [0]])
    il = torch.tensor([[12], [4], [3], [0], [0]])       # 0: fits; 1: 3 repeats need 5 frames, has 4; 2: 4 labels in 3 frames; 3: no frames; 4: empty / empty
    p = pred.clone().requires_grad_(True)
    cost = ctc_batch_cost(p, labels.to(dev), il, ll)
    cost[0].sum().backward()
    cst = cost.detach().cpu().reshape(-1)
    assert torch.isfinite(cst[0]) and cst[0] > 0 and float(cst[4]) == 0.0
    assert torch.isinf(cst[1]) and torch.isinf(cst[2]) and torch.isinf(cst[3]) and (cst[1:4] > 0).all()
    p2 = pred.clone().requires_grad_(True)
    c2 = ctc_batch_cost(p2, labels.to(dev), il, ll)
    torch.where(torch.isfinite(c2), c2, torch.zeros_like(c2)).sum().backward(retain_graph=True)
    assert float(p2.grad[0].abs().sum()) > 0 and float(p2.grad[1:].abs().sum()) == 0.0
    # the stored gradient rows of the infeasible samples are zero themselves (not merely masked by the caller)
    from qcnn_amd import functional as Fq
    p3 = pred.clone().requires_grad_(True)
    c3 = Fq.ctc_batch_cost(p3, labels, il, ll)
    gr, = torch.autograd.grad(c3, p3, torch.ones_like(c3))
    assert torch.isfinite(gr).all() and float(gr[1:].abs().sum()) == 0.0 and float(gr[0].abs().sum()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize('shape_loss', [(4, 24, 'sum'), (44, 192, 'ctc')], ids=['small_sumloss', 'fused_softmax_ctc'])
def test_deterministic_mode_makes_the_training_step_bit_repeatable(shape_loss):
    """QK_DBG_DETERMINISTIC (include/qk.h): every backward-weight kernel runs one split of its reduction per gradient
    tile and one owner per bias column, so no float sum depends on the order in which atomics land (TF's CPU
    Conv2DBackpropFilter, the reference's path, is deterministic too).  Three whole training steps of the headline graph
    (relu + dropout, l2 folded into Adam, bf16) started twice from the same seeds must then agree BIT FOR BIT in every
    parameter and both Adam moments; the default mode is only required to agree to rounding."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import bench
    from qcnn_amd import _lib
    dev = torch.device('cuda:0')
    # second case (round-4 advisor): 44 x 192 = 8448 rows reach the fused Dense(62) + softmax layer (>= 8192 rows: its bias
    # gradient was a grid of float atomics) and the model's own CTC cost (LDS float atomics in the occupancy sums) -- both
    # take their fixed-order forms under the flag
    bsz, frames, loss_kind = shape_loss
    cfg = dict(kind='model', batch=bsz, frames=frames, sf=32, layers=4, dtype='bf16', dropout=0.25, l2=1e-4, activation='relu')

    def run(flags):
        with _lib.debug_flags(flags):
            job = bench.ModelTrainStep(cfg, dev, 0, 1, loss=loss_kind)
            torch.manual_seed(7)                     # the dropout masks of the three steps
            for _ in range(3):
                job.step()
            torch.cuda.synchronize()
            return job.flat.param.clone(), job.m.clone(), job.v.clone()
    a = run(_lib.QK_DBG_DETERMINISTIC)
    b = run(_lib.QK_DBG_DETERMINISTIC)
    for x, y, name in zip(a, b, ('parameters', 'first moment', 'second moment')):
        assert torch.equal(x, y), '%s differ between two deterministic runs: max |diff| %g' % (name, float((x - y).abs().max()))
    c = run(0)
    # same arithmetic, other summation order: Adam moves a parameter by at most ~lr per step whatever the gradient's size, so
    # rounding-level differences of near-zero gradients stay below steps x lr = 1.5e-3 in absolute terms
    assert float((a[0] - c[0]).abs().max()) <= 2e-3


# ---- bench.py: the driver's contract ----------------------------------------------------------------------------
def test_bench_defaults_name_the_headline_workload():
    """`python bench.py` with no flags = the full TIMIT QCNN step, per-GPU batch 256, bf16 (BASELINE configs[2]/[3]) for
    every N; the FLOP count of that step is the sum of its quaternion layers' 2MNK."""
    import bench
    assert bench.DEFAULT_WORKLOAD == 'cfg3_qcnn_relu_dropout_b256_bf16'
    cfg = bench.WORKLOADS[bench.DEFAULT_WORKLOAD]
    assert (cfg['kind'], cfg['batch'], cfg['dtype'], cfg['layers'], cfg['sf'], cfg['frames']) == ('model', 256, 'bf16', 10, 32, 200)
    # the graph the reference builds for aact='none': Dropout behind every body conv, l2 on every kernel
    # (interspeech_model.py:63,68,117-121,131-137); the dropout-free step of rounds 1-2 stays available
    assert cfg['dropout'] == 0.3 and cfg['l2'] > 0 and cfg.get('aact', 'none') == 'none'
    old = bench.WORKLOADS['cfg3_qcnn_timit_b256_bf16']
    assert old.get('dropout', 0.0) == 0.0 and old.get('l2', 0.0) == 0.0
    m0, m1, mt = 256 * 41 * 200, 256 * 14 * 200, 256 * 200
    want = 2.0 * m0 * 128 * 60
    want += 2.0 * m1 * (5 * 128 * 1920 + 256 * 1920 + 4 * 256 * 3840)
    want += 2.0 * mt * 256 * 3584 + 2 * 2.0 * mt * 256 * 256
    assert abs(bench.qcnn_flops(32, 10, 256, 200) - want) <= 1e-6 * want
    assert set(bench.PEAK_TFLOPS) >= {'fp32', 'bf16', 'fp16'} and bench.PEAK_TFLOPS['bf16'] == 2500.0


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    """The default bench command (shortened) on the GPU: exactly one JSON line on stdout with the contract's fields, the
    in-step per-call table and a roofline block taken from it."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
                          '--no-extras', '--no-standalone'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'pre_warmup_steps'):
        assert k in d, k
    assert (d['n_gpus'], d['steps'], d['warmup'], d['unit'], d['dtype'], d['data'], d['scaling']) == (1, 3, 1, 'samples/s', 'bf16', 'synthetic', 'weak')
    assert d['config']['workload'] == 'cfg3_qcnn_relu_dropout_b256_bf16' and d['vs_baseline'] is None and d['higher_is_better'] is True
    assert d['config']['dropout'] == 0.3 and d['config']['l2'] > 0 and d['config']['loss'] == 'ctc' and d['config']['rccl_ranks'] is None
    assert d['config']['input'] == [256, 4, 41, 200] and d['config']['input_layout'].startswith('channels_first')
    assert abs(d['value'] - 256 * 1e3 / d['ms_per_step']) <= 1e-6 * d['value']
    calls = d['in_step_kernels']['calls']
    assert {(c['rows'], c['n'], c['k']) for c in calls} >= {(716800, 256, 3840), (716800, 128, 1920), (716800, 256, 1920)}
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['peak'] == 2500.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    assert abs(r['achieved'] - r['flops_per_launch'] / (r['avg_launch_ms'] * 1e-3) / 1e12) <= 1e-6 * r['achieved']


@pytest.mark.gpu
def test_timit_training_example_runs_and_learns():
    """examples/train_timit_synthetic.py: the reference's getTimitModel2D attribute bag, CTC cost, l2 in Adam, flat buffers -- the
    cost of a memorised synthetic batch must fall."""
    import re
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'examples', 'train_timit_synthetic.py'), '--steps', '30', '--batch', '4', '--frames', '96',
                          '--layers', '2', '--dropout', '0.1'], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    costs = [float(v) for v in re.findall(r'ctc cost ([0-9.]+)', out.stdout)]
    assert len(costs) >= 3 and costs[-1] < 0.5 * costs[0], costs


def test_real_branch_of_getTimitModel2D_builds_and_trains_on_cpu():
    """`d.model == "real"` (interspeech_model.py:92-96,109-113,124-128,159-169): the stock-layer comparison network the reference
    builds beside the quaternion one -- Conv2D stack on (B, 3, 41, T), Dense(1024) x 3, Dense(62, softmax), CTC.  Plain torch
    layers (it is not the Hamilton hot path), so it runs anywhere: shapes, parameter count of the Keras graph, a finite
    training loss with every parameter receiving a gradient."""
    from qcnn_amd.models.interspeech_model import getTimitModel2D, TimitRealCNN
    np.random.seed(0)
    torch.manual_seed(0)
    d = types.SimpleNamespace(model='real', num_layers=4, start_filter=8, act='relu', aact='prelu', dropout=0.2, l2=1e-4)
    m, val = getTimitModel2D(d)
    assert isinstance(m, TimitRealCNN)
    x = torch.randn(2, 3, 41, 30)
    y = val(x)
    assert tuple(y.shape) == (2, 30, 62) and abs(float(y.sum(-1).mean()) - 1.0) < 1e-5
    convs = 15 * 3 * 8 + 8 + 2 * (15 * 8 * 8 + 8) + (15 * 8 * 16 + 16) + (15 * 16 * 16 + 16)
    dense = (14 * 16 * 1024 + 1024) + 2 * (1024 * 1024 + 1024) + 1024 * 62 + 62
    prelu = 41 + 4 * 14 + 3 * 1                        # PReLU(shared_axes=[1, 0]): one slope per frequency row / one per dense layer
    assert sum(p.numel() for p in m.parameters()) == convs + dense + prelu
    labels, il, ll = torch.randint(0, 61, (2, 5)), torch.full((2, 1), 30), torch.full((2, 1), 5)
    loss = m.training_loss(x, labels, il, ll)
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    # loss_scale (float16 training) moves GRADIENTS only, of the CTC term and of the l2 term alike: the value returned -- what a
    # training loop logs -- is the unscaled loss (round-5 advisor: the regulariser used to be scaled in its value)
    ps = list(m.parameters())
    torch.manual_seed(5)
    l1 = m.training_loss(x, labels, il, ll)
    g1 = torch.autograd.grad(l1, ps)
    torch.manual_seed(5)
    l2 = m.training_loss(x, labels, il, ll, loss_scale=4096.0)
    g2 = torch.autograd.grad(l2, ps)
    assert float(l1) == float(l2) and float(m.regularization_loss()) > 0
    assert all(torch.allclose(b, a * 4096.0, rtol=1e-4, atol=0) for a, b in zip(g1, g2))
    with pytest.raises(ValueError):
        getTimitModel2D(types.SimpleNamespace(model='complex', num_layers=2, start_filter=4, act='relu', aact='none', dropout=0.0))


@pytest.mark.gpu
@pytest.mark.parametrize('model', ['QDNN', 'QCNN'])
def test_working_example_script_runs_one_epoch_on_the_decoda_fixture(tmp_path, model):
    """examples/working_example.py -- the counterpart of the reference's only runnable entry point (working_example.py:88-137,
    BASELINE configs[0]) -- end to end on the 8-document DECODA fixture (the first lines of the reference's own DEV file):
    data prep, the example network on the engine's layers, Adam, categorical cross-entropy, evaluation.  The reference ships
    no TRAIN file either, so the script trains on DEV, as it says."""
    import shutil
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, 'tests', 'golden', 'decoda_dev_head8.data')
    for split in ('DEV', 'TEST'):
        shutil.copy(src, tmp_path / ('250_%s_Q.data' % split))
    out = subprocess.run([sys.executable, os.path.join(root, 'examples', 'working_example.py'), '--model', model, '--decoda', str(tmp_path),
                          '--epochs', '1', '--batch-size', '3'], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'Train size : 8' in out.stdout and 'training on DEV' in out.stdout
    ep = [l for l in out.stdout.splitlines() if l.startswith('epoch  1')]
    assert len(ep) == 1 and 'dev loss' in ep[0]
    last = [l for l in out.stdout.splitlines() if l.startswith('Test Loss')]
    assert len(last) == 1
    loss, acc = float(last[0].split('=')[1].split('|')[0]), float(last[0].split('=')[2])
    assert np.isfinite(loss) and 0.0 < loss < 5.0 and 0.0 <= acc <= 1.0


@pytest.mark.gpu
def test_model_step_replayed_as_one_graph_equals_the_eager_step():
    """bench.ModelTrainStep.capture(): forward, CTC cost, backward, fused Adam and the kernel re-layout as ONE hipGraph.  The Adam
    step number and the dropout seeds' per-step part live in a device counter (qk_adam_step_dev, qk_postop_t.drop_seed_dev), so
    replays are new steps, not repetitions: three replays must leave the parameters and both Adam moments BIT-identical to
    three eager steps driven by the same device counter (deterministic mode: no order-dependent float sums), the masks of
    consecutive steps must differ, and the counter must have advanced."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import bench
    from qcnn_amd import _lib
    dev = torch.device('cuda:0')
    cfg = dict(kind='model', batch=4, frames=40, sf=32, layers=4, dtype='bf16', dropout=0.25, l2=1e-4, activation='relu')

    def run(graph):
        with _lib.debug_flags(_lib.QK_DBG_DETERMINISTIC):
            job = bench.ModelTrainStep(cfg, dev, 0, 1, loss='ctc')
            job.model._new_drop_base = lambda: setattr(job.model, '_drop_calls', 0) or setattr(job.model, '_drop_base', 12345)
            if graph:
                job.capture()                        # two eager device-counter steps, then the capture pass (which executes nothing)
            else:
                job.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
                job.model.drop_step_dev = job.step_dev
                for _ in range(2):
                    job.step()
            params = [job.flat.param.clone()]
            for _ in range(3):
                job.step()
                params.append(job.flat.param.clone())
            torch.cuda.synchronize()
            return params, job.m.clone(), job.v.clone(), int(job.step_dev.item())
    pe, me, ve, te = run(False)
    pg, mg, vg, tg = run(True)
    assert te == tg == 5
    for i, (a, b) in enumerate(zip(pe, pg)):
        assert torch.equal(a, b), 'parameters after %d steps differ between eager and graph replay: %g' % (2 + i, float((a - b).abs().max()))
    assert torch.equal(me, mg) and torch.equal(ve, vg)
    assert not torch.equal(pg[1] - pg[0], pg[2] - pg[1])           # replays are steps, not repetitions
