"""f2 parity: the TIMIT quaternion CNN (models/interspeech_model.py:45-185) on the GPU against the
float64 oracle composition of tests/np_model_ref.py -- forward AND every gradient, with the fused paths on
(first layer tap-folded, engine max-pool, body convolutions + head as one chain node, head as an (F, 1)
conj-convolution): conj = 1 convolutions at rank 2, QK_BWD_MASK_DX / QK_BWD_DY_PREMASKED, qk_conv_fold_taps and
qk_maxpool2d meet the oracle here directly, not another HIP kernel.

Tolerances: fp32 <= 1e-4 of max|want| per tensor (BASELINE.json north_star); bf16: the composition emulates
the 16-bit storage of every activation and the 16-bit kernels of the matrix-core path, forward <= 2e-2 (4e-3 fp16);
gradients <= 1.5e-1 (8e-2 fp16) in relative 2-norm (observed 3 - 7 % / 2 - 5 %, varying from run to run with the order of
the atomic accumulation): the backward passes through ten 16-bit tensors the composition
does not round, and relu masks are decided by each side's own outputs (an output that rounds across zero moves
single gradient elements by a full term, so the element-wise maximum is not a meaningful bound there).  The tight
statement about the structure is the fp32 test, the tight statements about the 16-bit kernels are the layer tests.
"""
import numpy as np
import pytest
import torch

from np_model_ref import TimitRef, keras_prelu_alpha_shape, model_grads


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def _rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    return float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-30)


def _build(dev, dtype, sf, n, aact, bsz, t, seed, **kw):
    from qcnn_amd.models import TimitQCNN
    np.random.seed(seed)
    torch.manual_seed(seed)
    model = TimitQCNN(num_layers=n, start_filter=sf, act='relu', aact=aact, dropout=0.0, **kw)
    rng = np.random.RandomState(seed + 1)
    x = rng.randn(bsz, 4, 41, t).astype(np.float32)
    xt = torch.tensor(x, device=dev).to(dtype)
    with torch.no_grad():
        model(xt)                                           # build
        # biases / PReLU slopes away from their zero initial values, so that their paths carry signal
        for name, p in model.named_parameters():
            if name.endswith('bias'):
                p.copy_(torch.tensor(0.1 * rng.randn(*p.shape), dtype=torch.float32))
            if name.endswith('alpha'):
                p.copy_(torch.tensor(0.05 + 0.3 * rng.rand(*p.shape), dtype=torch.float32))
    dpred = rng.randn(bsz, t, 62)
    return model, xt, dpred


@pytest.mark.gpu
@pytest.mark.parametrize('aact', ['none', 'prelu'])
@pytest.mark.parametrize('fused', [True, False], ids=['fused', 'layerwise'])
def test_timit_qcnn_fp32_matches_oracle_composition(aact, fused):
    dev = _dev()
    model, xt, dpred = _build(dev, torch.float32, 8, 4, aact, 2, 20, seed=11, fuse_head=fused, chain_convs=fused)
    xt.requires_grad_(False)
    pred = model(xt)
    (pred.double() * torch.tensor(dpred, device=dev)).sum().backward()
    ref = TimitRef(model, act='relu')
    want = ref.forward(xt.detach().cpu().double().numpy())
    assert _rel(pred.detach().cpu().numpy(), want) <= 1e-4
    wg = ref.backward(dpred)
    got = model_grads(model)
    for k, v in got.items():
        assert v is not None, k
        err = _rel(v, wg[k])
        # the scalar slopes behind the dense layers are one fp32 torch reduction over ~10^4 cancelling terms
        tol = 1e-3 if (k.startswith('alpha') and v.size == 1) else 1e-4
        assert err <= tol, '%s: rel err %.3g' % (k, err)


def _round_fn(dtype):
    return lambda a: torch.tensor(a).to(dtype).double().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
def test_timit_qcnn_16bit_matches_oracle_composition(dtype):
    """sf = 32: every quaternion layer is on the 16-bit MFMA path (band kernels, chain flags, conj head, fold)."""
    dev = _dev()
    model, xt, dpred = _build(dev, dtype, 32, 4, 'none', 2, 24, seed=13)
    pred = model(xt)
    (pred.float() * torch.tensor(dpred, device=dev, dtype=torch.float32)).sum().backward()
    rnd = _round_fn(dtype)
    ref = TimitRef(model, act='relu', rnd=rnd, rnd_w=rnd)
    want = ref.forward(xt.detach().cpu().double().numpy())
    tol_f, tol_g = (2e-2, 1.5e-1) if dtype == torch.bfloat16 else (4e-3, 8e-2)
    assert _rel(pred.detach().float().cpu().numpy(), want) <= tol_f
    wg = ref.backward(dpred)
    got = model_grads(model)
    for k, v in got.items():
        assert v is not None, k
        # 2-norm: an output that rounds across zero on one side only (relu mask) moves a few gradient elements by
        # a full term -- with 48 rows behind each element of the head's gradient that is tens of percent of one
        # element, and noise in the norm; the element-wise bound only guards against gross errors
        l2 = float(np.linalg.norm(v - wg[k])) / max(float(np.linalg.norm(wg[k])), 1e-30)
        assert l2 <= tol_g, '%s: relative 2-norm error %.3g' % (k, l2)
        assert _rel(v, wg[k]) <= 0.35, '%s: max-abs rel err %.3g' % (k, _rel(v, wg[k]))


@pytest.mark.gpu
def test_prelu_alpha_follows_keras_shared_axes_and_accepts_other_lengths():
    """PReLU(shared_axes=[1, 0]) (interspeech_model.py:99-101): Keras shares axis 1 and -- through index 0 - 1 = -1 --
    the LAST axis; a second batch with another number of frames must run on the same parameters."""
    dev = _dev()
    model, xt, _ = _build(dev, torch.float32, 8, 2, 'prelu', 2, 16, seed=3)
    shapes = [tuple(p.alpha.shape) for p in model.prelu]
    assert shapes[0] == keras_prelu_alpha_shape((None, 32, 41, None), [1, 0]) == (1, 41, 1)
    assert shapes[1] == shapes[2] == keras_prelu_alpha_shape((None, 32, 14, None), [1, 0]) == (1, 14, 1)
    assert shapes[3] == shapes[4] == shapes[5] == keras_prelu_alpha_shape((None, None, 256), [1, 0]) == (1, 1)
    x2 = torch.randn(3, 4, 41, 23, device=dev)
    assert tuple(model(x2).shape) == (3, 23, 62)


def test_prelu_shapes_on_cpu():
    from qcnn_amd.layers import PReLU
    p = PReLU(shared_axes=[1, 0])
    p(torch.randn(2, 8, 14, 5))
    assert tuple(p.alpha.shape) == (1, 14, 1)
    q = PReLU(shared_axes=[1, 0])
    y = q(torch.tensor([[[-2.0, 3.0]]]))
    assert tuple(q.alpha.shape) == (1, 1) and torch.equal(y, torch.tensor([[[0.0, 3.0]]]))
    r = PReLU(shared_axes=[2])
    r(torch.randn(2, 3, 4, 5))
    assert tuple(r.alpha.shape) == (3, 1, 5)
    p(torch.randn(1, 8, 14, 9))                              # another time length, same parameters


def test_numpy_model_composition_agrees_with_torch_autograd_on_cpu():
    """The float64 composition the GPU tests compare against, checked here against torch autograd through the
    reference op sequence (oracle/ref_model.py on oracle/ref_port.py): two independent restatements of the model."""
    import types
    from oracle import ref_model
    rng = np.random.RandomState(0)
    n, sf, bsz, t = 2, 2, 2, 7
    x = torch.tensor(rng.randn(bsz, 4, 41, t), dtype=torch.float64, requires_grad=True)
    dpred = rng.randn(bsz, t, 62)
    for use_prelu in (False, True):
        p = ref_model.init_params(n, sf, seed=1, dtype=torch.float64, prelu=use_prelu)
        pred = ref_model.timit_forward(x, p, 'relu')
        lv = [x] + ref_model.leaves(p)
        grads = torch.autograd.grad((pred * torch.tensor(dpred)).sum(), lv)
        fake = types.SimpleNamespace(
            conv=types.SimpleNamespace(kernel=p['conv'][0], bias=p['conv'][1]),
            convs=[types.SimpleNamespace(kernel=w, bias=b) for w, b in p['convs']],
            dense=[types.SimpleNamespace(layer=types.SimpleNamespace(r=w, bias=b)) for w, b in p['dense']],
            pred=types.SimpleNamespace(layer=types.SimpleNamespace(kernel=p['pred'][0], bias=p['pred'][1])),
            prelu=[types.SimpleNamespace(alpha=a) for a in p['alphas']] if use_prelu else None)
        ref = TimitRef(fake, act='relu')
        want = ref.forward(x.detach().numpy())
        assert np.abs(want - pred.detach().numpy()).max() <= 1e-12
        g = ref.backward(dpred)
        names = ['x', 'conv.kernel', 'conv.bias'] + [s_ % i for i in range(n) for s_ in ('conv%d.kernel', 'conv%d.bias')] \
            + [s_ % i for i in range(3) for s_ in ('dense%d.r', 'dense%d.bias')] + ['pred.kernel', 'pred.bias']
        if use_prelu:
            names += ['alpha%d' % i for i in range(len(p['alphas']))]
        assert len(names) == len(grads)
        for name, tg in zip(names, grads):
            assert np.abs(g[name] - tg.numpy()).max() <= 1e-10 * max(1.0, float(tg.abs().max())), name


# ---- pinned to the reference's own model file (fixtures g17_*: getTimitModel2D executed through the stand-in) ----
import os

from np_model_ref import fake_model_from_weights, load_timit_fixture, load_weights_into_model

_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('variant', ['relu', 'prelu'])
def test_numpy_composition_reproduces_the_reference_model_fixture(variant):
    """tests/np_model_ref.TimitRef against what the reference's getTimitModel2D (models/interspeech_model.py:45-185)
    computed -- posteriors and the gradient w.r.t. the input and every weight -- to float64 round-off."""
    fx = load_timit_fixture(os.path.join(_GOLD, 'g17_timit_%s.npz' % variant))
    ref = TimitRef(fake_model_from_weights(fx['weights']), act='relu')
    pred = ref.forward(fx['x'].astype(np.float64))
    assert pred.shape == fx['pred'].shape and np.abs(pred - fx['pred']).max() <= 1e-12
    g = ref.backward(fx['dpred'].astype(np.float64))
    assert set(g) == set(fx['grads'])
    for k, want in fx['grads'].items():
        assert np.abs(g[k] - want).max() <= 1e-10 * max(1.0, float(np.abs(want).max())), k


@pytest.mark.gpu
@pytest.mark.parametrize('fused', [True, False], ids=['fused', 'layerwise'])
@pytest.mark.parametrize('variant', ['relu', 'prelu'])
def test_timit_qcnn_matches_the_reference_model_fixture(variant, fused):
    """qcnn_amd.models.getTimitModel2D on the GPU, loaded with the fixture's weights, against the reference's own
    model code: posteriors, CTC cost (K.ctc_batch_cost, interspeech_model.py:37-39) and every gradient, fp32 <= 1e-4."""
    import types
    import qcnn_amd
    dev = _dev()
    fx = load_timit_fixture(os.path.join(_GOLD, 'g17_timit_%s.npz' % variant))
    d = types.SimpleNamespace(model='quaternion', quat_init='quaternion', **fx['d'])
    np.random.seed(0)
    model, val = qcnn_amd.models.getTimitModel2D(d)
    model.fuse_head, model.chain_convs = fused, fused
    x = torch.tensor(fx['x'], device=dev)
    with torch.no_grad():
        model(x)
    load_weights_into_model(model, fx['weights'])
    pred = model(x)
    assert _rel(pred.detach().cpu().numpy(), fx['pred']) <= 1e-4
    cost = model.ctc_loss(x, torch.tensor(fx['labels'], device=dev), torch.tensor(fx['input_length'], device=dev),
                          torch.tensor(fx['label_length'], device=dev))
    assert _rel(cost.detach().cpu().numpy(), fx['ctc_cost']) <= 1e-4
    (pred.double() * torch.tensor(fx['dpred'], device=dev).double()).sum().backward()
    got = model_grads(model)
    for k, v in got.items():
        err = _rel(v, fx['grads'][k])
        tol = 1e-3 if (k.startswith('alpha') and v.size == 1) else 1e-4
        assert err <= tol, '%s: rel err %.3g' % (k, err)
    # Keras adds the l2 terms of every kernel to the training loss (interspeech_model.py:63,68,173)
    l2 = fx['d']['l2']
    want_reg = l2 * sum(float((fx['weights'][k] ** 2).sum()) for k in fx['weights'] if k.endswith('.kernel') or k.endswith('.r'))
    assert abs(float(model.regularization_loss()) - want_reg) <= 1e-5 * want_reg
