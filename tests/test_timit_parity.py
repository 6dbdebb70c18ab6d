"""f2 parity: the TIMIT quaternion CNN (models/interspeech_model.py:45-185) on the GPU against the
float64 oracle composition of tests/np_model_ref.py -- forward AND every gradient, with the fused paths on
(first layer tap-folded, engine max-pool, body convolutions + head as one chain node, head as an (F, 1)
conj-convolution): conj = 1 convolutions at rank 2, QK_BWD_MASK_DX / QK_BWD_DY_PREMASKED, qk_conv_fold_taps and
qk_maxpool2d meet the oracle here directly, not another HIP kernel.

Tolerances: fp32 <= 1e-4 of max|want| per tensor (BASELINE.json north_star); bf16 / fp16: the composition emulates
the 16-bit storage of every activation and the 16-bit kernels of the matrix-core path, forward <= 2e-2 (4e-3 fp16).
16-bit GRADIENTS are compared ELEMENT-WISE on a common relu mask (round 3): the composition is teacher-forced with the
GPU's own layer outputs (functional.chain_tap + module hooks), so that a pre-activation that rounds across zero on
one side only -- which moves single gradient elements by a full term and forced a 15 % 2-norm tolerance in round 2 --
can no longer differ; what remains is the 16-bit rounding of the gradient tensors between layers.  Every forced
output is itself checked against the composition's own value of that layer (a per-layer parity statement).
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


def _np(t):
    return t.detach().double().cpu().numpy()


def _capture_layer_outputs(model, xt, run):
    """Run `run()` (forward [+ backward]) and return the GPU's own outputs of the pooled first layer, every body
    convolution and the three dense layers, as the composition's `forced` dictionary."""
    import qcnn_amd
    Fq = qcnn_amd.functional
    taps, dense_out = [], {}
    hooks = [model.dense[i].register_forward_hook(lambda m, a, o, i=i: dense_out.__setitem__(i, o)) for i in range(3)]
    Fq.chain_tap = taps.append
    try:
        out = run()
    finally:
        Fq.chain_tap = None
        for h in hooks:
            h.remove()
    assert taps, 'the body convolutions did not run as a chain'
    acts = taps[-1]                                         # [pooled input, y_conv0 .. y_conv{n-1}, (head)] channels-last
    n = len(model.convs)
    forced = {'pool': _np(acts[0].movedim(-1, 1))}
    for i in range(n):
        forced['y_c%d' % i] = _np(acts[1 + i].movedim(-1, 1))
    if len(acts) > n + 1:                                   # the first dense layer ran inside the chain as an (F, 1) convolution
        forced['y_d0'] = _np(acts[n + 1].reshape(-1, acts[n + 1].shape[-1]))
    for i, o in dense_out.items():
        forced['y_d%d' % i] = _np(o.reshape(-1, o.shape[-1]))
    return out, forced


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
def test_timit_qcnn_16bit_matches_oracle_composition(dtype):
    """sf = 32: every quaternion layer is on the 16-bit MFMA path (band kernels, chain flags, conj head, fused first
    layer).  Forward against the free-running composition; every layer output against the composition's value GIVEN the
    GPU's input to that layer; gradients element-wise on the GPU's own relu masks."""
    dev = _dev()
    model, xt, dpred = _build(dev, dtype, 32, 4, 'none', 2, 24, seed=13)

    def run():
        pred = model(xt)
        (pred.float() * torch.tensor(dpred, device=dev, dtype=torch.float32)).sum().backward()
        return pred
    pred, forced = _capture_layer_outputs(model, xt, run)
    assert set(forced) == {'pool', 'y_c0', 'y_c1', 'y_c2', 'y_c3', 'y_d0', 'y_d1', 'y_d2'}
    rnd = _round_fn(dtype)
    tol_f, tol_g = (2e-2, 2e-2) if dtype == torch.bfloat16 else (4e-3, 5e-3)
    x64 = xt.detach().cpu().double().numpy()
    free = TimitRef(model, act='relu', rnd=rnd, rnd_w=rnd)
    assert _rel(pred.detach().float().cpu().numpy(), free.forward(x64)) <= tol_f
    ref = TimitRef(model, act='relu', rnd=rnd, rnd_w=rnd)
    want = ref.forward(x64, forced=forced)
    assert max(ref.forced_err.values()) <= tol_f, ref.forced_err            # layer by layer, same inputs
    assert _rel(pred.detach().float().cpu().numpy(), want) <= tol_f
    wg = ref.backward(dpred)
    got = model_grads(model)
    errs = {k: _rel(v, wg[k]) for k, v in got.items()}
    # first layer: its relu / arg-max decisions sit in FRONT of the first forced tensor (the fused kernel never writes
    # the pre-pool activation), so single elements of ITS kernel gradient may still differ by a full term (observed
    # 6e-2 bf16 / 2e-2 fp16; that kernel meets the oracle directly in test_fused_first_layer_conv_relu_pool_matches_oracle);
    # everything behind the pooling shares the GPU's masks: observed <= 7e-3 bf16 / 1e-3 fp16
    for k, e in errs.items():
        lim = 5 * tol_g if k == 'conv.kernel' else tol_g
        assert e <= lim, ('%s: element-wise rel err %.3g' % (k, e), errs)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_timit_qcnn_relu_dropout_matches_oracle_composition(dtype):
    """The graph the reference builds for aact='none' with d.dropout > 0 (interspeech_model.py:117-121,131-137,150-154):
    relu layers with Dropout behind every body convolution and the first two dense layers, here with relu + dropout
    fused into the producing kernels (one output tensor per layer) and the derivative in the consumer's backward-data
    epilogue.  The composition applies the same masks (the kernels' counter-based hash of (seed, element index),
    restated in numpy) -- posteriors and every gradient."""
    from test_gpu_parity import _np_drop_factor
    dev = _dev()
    from qcnn_amd.models import TimitQCNN
    sf = 8 if dtype == torch.float32 else 32
    np.random.seed(21); torch.manual_seed(21)
    rate = 0.25
    model = TimitQCNN(num_layers=4, start_filter=sf, act='relu', aact='none', dropout=rate)
    rng = np.random.RandomState(22)
    bsz, t = 2, 24
    xt = torch.tensor(rng.randn(bsz, 4, 41, t).astype(np.float32), device=dev).to(dtype)
    model.train()
    with torch.no_grad():
        model(xt)
        for name, p in model.named_parameters():
            if name.endswith('bias'):
                p.copy_(torch.tensor(0.1 * rng.randn(*p.shape), dtype=torch.float32))
    dpred = rng.randn(bsz, t, 62)

    def run():
        pred = model(xt)
        (pred.double() * torch.tensor(dpred, device=dev)).sum().backward()
        return pred
    pred, forced = _capture_layer_outputs(model, xt, run)
    # the masks of this forward pass: seed of the k-th activation slot = base + 7919 k (TimitQCNN._post)
    base, n = model._drop_base, len(model.convs)
    seed = lambda k: (base + 7919 * k) & 0xffffffff
    keeps = {}
    for i in range(n):
        b_, c_, f_, t_ = forced['y_c%d' % i].shape
        keeps['c%d' % i] = np.moveaxis(_np_drop_factor((b_, f_, t_, c_), seed(2 + i), rate), -1, 1)
    keeps['d0'] = _np_drop_factor((bsz * t, 256), seed(2 + n), rate)
    keeps['d1'] = _np_drop_factor((bsz * t, 256), seed(3 + n), rate)
    dropped = np.mean([float((k == 0).mean()) for k in keeps.values()])
    assert abs(dropped - rate) < 0.02, dropped
    x64 = xt.detach().cpu().double().numpy()
    if dtype == torch.float32:
        ref = TimitRef(model, act='relu')
        want = ref.forward(x64, keeps=keeps)
        tol_f = tol_g = 1e-4
    else:
        rnd = _round_fn(dtype)
        ref = TimitRef(model, act='relu', rnd=rnd, rnd_w=rnd)
        want = ref.forward(x64, forced=forced, keeps=keeps)
        tol_f, tol_g = 2e-2, 2e-2
        assert max(ref.forced_err.values()) <= tol_f, ref.forced_err
    assert _rel(pred.detach().float().cpu().numpy(), want) <= tol_f
    wg = ref.backward(dpred)
    got = model_grads(model)
    errs = {k: _rel(v, wg[k]) for k, v in got.items()}
    for k, e in errs.items():
        lim = 5 * tol_g if (k == 'conv.kernel' and dtype != torch.float32) else tol_g
        assert e <= lim, ('%s: rel err %.3g' % (k, e), errs)
    # eval mode: Dropout is the identity (Keras' learning phase 0)
    model.eval()
    with torch.no_grad():
        pe = model(xt)
    ref_e = TimitRef(model, act='relu', **({} if dtype == torch.float32 else dict(rnd=_round_fn(dtype), rnd_w=_round_fn(dtype))))
    assert _rel(pe.float().cpu().numpy(), ref_e.forward(x64)) <= tol_f


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


# ---- round 5: float16 training under the model's own loss needs loss scaling (round-4 verdict, missing 3) -------------------
def test_ctc_loss_scale_multiplies_the_gradient_not_the_cost_on_cpu():
    """The torch path of layers.ctc_batch_cost (no GPU): loss_scale leaves the cost alone and multiplies d cost / d y."""
    from qcnn_amd.layers import ctc_batch_cost
    g = torch.Generator().manual_seed(3)
    y = torch.softmax(torch.randn(3, 12, 7, generator=g), -1).requires_grad_(True)
    labels = torch.randint(0, 6, (3, 4), generator=g)
    il, ll = torch.full((3, 1), 12), torch.tensor([[4], [2], [3]])
    c1 = ctc_batch_cost(y, labels, il, ll)
    g1, = torch.autograd.grad(c1.sum(), y)
    c2 = ctc_batch_cost(y, labels, il, ll, loss_scale=4096.0)
    g2, = torch.autograd.grad(c2.sum(), y)
    assert torch.equal(c1, c2)
    assert torch.allclose(g2, g1 * 4096.0, rtol=1e-6, atol=0)


@pytest.mark.gpu
def test_fp16_timit_step_under_ctc_matches_fp32_with_loss_scaling_and_underflows_without():
    """interspeech_model.py:37-39,178: the cost the model is trained on.  At B = 256 its gradients reach the body layers at
    2^-18.5 (profiles/r04_loss_ab.txt) -- below float16's normal range.  Emulated here on a small model by a 2^-16 factor on
    the loss: without loss scaling the float16 kernel gradients are flushed away; with `loss_scale = 2^16` (undone in fp32,
    as adam_step(grad_scale=) does) they match the float32 path (oracle-pinned elsewhere) to the 16-bit tolerance."""
    dev = _dev()
    model, xt, _ = _build(dev, torch.float32, 32, 4, 'none', 4, 40, seed=23, fuse_head=True, chain_convs=True)
    rng = np.random.RandomState(5)
    labels = torch.tensor(rng.randint(0, 61, (4, 10)), device=dev, dtype=torch.int32)
    il = torch.full((4, 1), 40, dtype=torch.int32, device=dev)
    ll = torch.tensor([[10], [7], [9], [4]], dtype=torch.int32, device=dev)
    tiny = 2.0 ** -16

    def grads(x, scale):
        for p in model.parameters():
            p.grad = None
        cost = model.ctc_loss(x, labels, il, ll, loss_scale=scale)
        (cost.mean() * tiny).backward()
        return {n: (p.grad.double() / (scale * tiny)).cpu().numpy() for n, p in model.named_parameters() if p.grad is not None}, cost.detach()

    want, c32 = grads(xt, 1.0)
    x16 = xt.to(torch.float16)
    got_s, c16 = grads(x16, 2.0 ** 16)
    got_p, c16p = grads(x16, 1.0)
    assert torch.equal(c16, c16p)                               # the scale never touches the cost itself
    assert float((c16.float() - c32).abs().max() / c32.abs().max()) <= 2e-2
    worst_s, worst_p = 0.0, 0.0
    for k, w in want.items():
        nw = np.linalg.norm(w)
        if nw == 0:
            continue
        es, ep = np.linalg.norm(got_s[k] - w) / nw, np.linalg.norm(got_p[k] - w) / nw
        assert np.isfinite(got_s[k]).all(), k
        assert es <= 8e-2, '%s: scaled fp16 gradient off by %.3g (2-norm)' % (k, es)
        worst_s, worst_p = max(worst_s, es), max(worst_p, ep)
    assert worst_p >= 0.5, 'the unscaled fp16 gradients were expected to underflow (worst 2-norm error %.3g)' % worst_p


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
def test_ctc_mean_loss_is_one_node_with_the_values_and_gradients_of_the_composition(dtype):
    """TimitQCNN.ctc_mean_loss (round 6): the output layer (qk_dense_softmax_fwd / _bwd), K.ctc_batch_cost and the mean over the batch
    (interspeech_model.py:37-39,171-178: what training minimises) as ONE autograd node whose backward hands the upstream scalar to the
    output layer's backward as a device pointer.  Against `ctc_loss(...).mean()` -- separate nodes, the same kernels: identical value;
    gradients equal up to one 16-bit rounding of d cost / d y (the fused node scales in fp32 inside the kernel) -- with an upstream
    factor and a loss scale; infeasible samples (cost +inf) send nothing back in either form."""
    dev = _dev()
    model, xt, _ = _build(dev, torch.float32, 32, 4, 'none', 4, 40, seed=29, fuse_head=True, chain_convs=True)
    rng = np.random.RandomState(7)
    labels = torch.tensor(rng.randint(0, 61, (4, 10)), device=dev, dtype=torch.int32)
    il = torch.full((4, 1), 40, dtype=torch.int32, device=dev)
    ll = torch.tensor([[10], [7], [9], [4]], dtype=torch.int32, device=dev)
    x = xt.to(dtype)
    up, scale = 0.37, (256.0 if dtype == torch.float16 else 1.0)

    def run(fused):
        for p in model.parameters():
            p.grad = None
        loss = model.ctc_mean_loss(x, labels, il, ll, loss_scale=scale) if fused else model.ctc_loss(x, labels, il, ll, loss_scale=scale).mean()
        (loss * up).backward()
        torch.cuda.synchronize()
        return float(loss), {n: p.grad.double().cpu().numpy() / (up * scale) for n, p in model.named_parameters() if p.grad is not None}
    import qcnn_amd.layers as lay
    calls = []
    orig = lay._DenseSoftmaxCtcMeanFn.apply
    lay._DenseSoftmaxCtcMeanFn.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        lf, gf = run(True)
    finally:
        lay._DenseSoftmaxCtcMeanFn.apply = orig
    assert calls, 'ctc_mean_loss did not take the fused node on a 16-bit device model'
    lu, gu = run(False)
    assert lf == lu and np.isfinite(lf)
    assert set(gf) == set(gu) and len(gf) >= 10
    for k, w in gu.items():
        nw = np.linalg.norm(w)
        if nw == 0:
            continue
        e = np.linalg.norm(gf[k] - w) / nw
        assert np.isfinite(gf[k]).all() and e <= (2e-2 if dtype == torch.bfloat16 else 4e-3), '%s: %.3g' % (k, e)
    # an infeasible sample: both forms give +inf for the mean and finite (zero-contribution) gradients for the feasible rest
    ll_bad = ll.clone()
    il_bad = il.clone()
    il_bad[1, 0] = 3                                         # 7 labels in 3 frames
    lb = model.ctc_mean_loss(x, labels, il_bad, ll_bad)
    assert torch.isinf(lb) and lb > 0


@pytest.mark.gpu
def test_full_b256_bf16_model_in_the_bench_form_meets_the_oracle_at_sampled_frames():
    """Round-4 verdict: "whole-model bf16 at B = 256 never meets the oracle in one piece".  The bench's model -- n = 10, sf = 32,
    relu + Dropout(0.3) fused into the producing kernels, training mode, (256, 4, 41, 200) bf16 channels_first input -- runs
    forward at FULL size; the float64 composition (np_model_ref.TimitRef around the C oracle, 16-bit storage emulated) then
    recomputes single output frames from a WINDOW of the input: eleven (3, 5) convolutions see 22 frames either side, so the
    posterior of frame t of sample n is a function of x[n, :, :, t - 22 : t + 23] alone (windows that reach the tensor's edge
    share its zero padding).  The dropout masks are the kernels' hash evaluated at the window's GLOBAL element indices.
    Reference: interspeech_model.py:81-175 (getTimitModel2D up to the softmax)."""
    from test_gpu_parity import _np_drop_factor
    from qcnn_amd.models import TimitQCNN
    dev = _dev()
    np.random.seed(31); torch.manual_seed(31)
    B, T, rate, n_layers, R = 256, 200, 0.3, 10, 22
    model = TimitQCNN(num_layers=n_layers, start_filter=32, act='relu', aact='none', dropout=rate)
    g = torch.Generator(device=dev).manual_seed(32)
    xt = torch.randn(B, 4, 41, T, device=dev, generator=g).to(torch.bfloat16)
    model.train()
    rng = np.random.RandomState(33)
    with torch.no_grad():
        model(xt[:2])
        for name, p in model.named_parameters():
            if name.endswith('bias'):
                p.copy_(torch.tensor(0.1 * rng.randn(*p.shape), dtype=torch.float32))
        pred = model(xt)
    torch.cuda.synchronize()
    assert tuple(pred.shape) == (B, T, 62)
    base = model._drop_base
    seed = lambda k: (base + 7919 * k) & 0xffffffff
    rnd = _round_fn(torch.bfloat16)
    ref = TimitRef(model, act='relu', rnd=rnd, rnd_w=rnd)
    widths = [c.kernel.shape[-1] for c in model.convs]
    worst = 0.0
    for n, t in ((0, 0), (255, T - 1), (137, 23), (64, 100)):            # both ends of the batch and of the time axis, two interior frames
        t0, t1 = max(0, t - R), min(T, t + R + 1)
        tw = t1 - t0
        x64 = xt[n:n + 1, :, :, t0:t1].detach().cpu().double().numpy()
        keeps = {}
        for i, c in enumerate(widths):                                  # conv i's output buffer: (B, 14, T, c) channels_last
            f_, tt, cc = np.meshgrid(np.arange(14), np.arange(t0, t1), np.arange(c), indexing='ij')
            idx = ((n * 14 + f_) * T + tt) * c + cc
            keeps['c%d' % i] = np.moveaxis(_np_drop_factor((1, 14, tw, c), seed(2 + i), rate, idx=idx), -1, 1)
        rows, cols = np.meshgrid(n * T + np.arange(t0, t1), np.arange(256), indexing='ij')
        keeps['d0'] = _np_drop_factor((tw, 256), seed(2 + n_layers), rate, idx=rows * 256 + cols)
        keeps['d1'] = _np_drop_factor((tw, 256), seed(3 + n_layers), rate, idx=rows * 256 + cols)
        want = ref.forward(x64, keeps=keeps)[0, t - t0]
        got = pred[n, t].float().cpu().numpy()
        err = float(np.abs(got - want).max()) / float(np.abs(want).max())
        worst = max(worst, err)
        assert err <= 6e-2, 'sample %d frame %d: posterior off by %.3g of its maximum' % (n, t, err)
    print('full-size model vs windowed oracle composition: worst relative error %.3g' % worst)


@pytest.mark.gpu
def test_start_filter_16_model_runs_on_the_matrix_cores_and_matches_its_fp32_run():
    """interspeech_model.py:46-50: start_filter = 16 -- a first layer with 16 filters (fused kernels on the kernel zero-padded to
    32), 16 -> 16 and 16 -> 32 body layers (PAD forms of the band kernels, 16 x 32 backward-weight blocks).  bf16 forward against
    the float64 composition on 16-bit storage; every gradient against the same model run in float32 (whose kernels meet the
    oracle to 1e-4 elsewhere) in the 2-norm -- single elements differ by whole terms where a bf16 pre-activation rounds across 0."""
    from qcnn_amd import _lib
    dev = _dev()
    model, xt, dpred = _build(dev, torch.float32, 16, 4, 'none', 2, 24, seed=41, fuse_head=True, chain_convs=True)
    dp = torch.tensor(dpred, device=dev, dtype=torch.float32)

    def run(x):
        for p in model.parameters():
            p.grad = None
        pred = model(x)
        (pred.float() * dp).sum().backward()
        return pred.detach().float().cpu().numpy(), {n: p.grad.double().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None}
    p32, g32 = run(xt)
    x16 = xt.to(torch.bfloat16)
    p16, g16 = run(x16)
    assert _lib.last_path().startswith('mfma16'), _lib.last_path()
    rnd = _round_fn(torch.bfloat16)
    want = TimitRef(model, act='relu', rnd=rnd, rnd_w=rnd).forward(x16.detach().cpu().double().numpy())
    assert _rel(p16, want) <= 2e-2
    assert _rel(p16, p32) <= 5e-2
    for k, w in g32.items():
        nw = np.linalg.norm(w)
        if nw > 0:
            e = np.linalg.norm(g16[k] - w) / nw
            assert e <= 0.15, '%s: bf16 gradient off by %.3g (2-norm) from the float32 run' % (k, e)
