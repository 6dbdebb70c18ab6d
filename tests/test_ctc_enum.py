"""K.ctc_batch_cost (/root/reference/models/interspeech_model.py:37-39, :178) pinned at DEFINITION level.

oracle/ctc_enum.py sums the probability of every frame-level path that collapses to the labels (float64, T <= 6, C <= 4) with
Keras' own recipe for the frame probabilities (softmax of log(y_pred + 1e-7)); its gradient is a central finite difference of that
sum.  No alpha / beta recursion is involved, so this is independent of torch.nn.functional.ctc_loss AND of the builder's reading of
the recursion:

  CPU:  oracle/ref_model.ctc_cost, the stand-in's K.ctc_batch_cost (which produced golden fixture G17's `ctc_cost`) and the package's
        torch path (layers.ctc_batch_cost on CPU tensors: float32, 2e-6 / 1e-4) equal the enumeration -- cost to 1e-10, autograd gradient to 1e-6;
  GPU:  qk_ctc_batch_cost, both kernel forms (concurrent sweeps / two sweeps), fp32 and 16-bit -- cost and gradient.

Tolerances: float64 restatements 1e-10 (cost), 1e-6 (gradient vs central finite differences, relative step 1e-4); the fp32 kernels 2e-5 relative
on the cost and 5e-4 of the largest gradient entry (fast-math exp / log); bf16 / fp16 operands: the enumeration runs on the ROUNDED
posteriors, the stored gradient carries one 16-bit rounding (1e-2 / 2e-3).
"""
import numpy as np
import pytest
import torch

from oracle import ctc_enum

CASES = ctc_enum.tiny_cases()
IDS = [c[0] for c in CASES]


def test_collapse_rule_is_merge_then_drop_blanks():
    B = ctc_enum.collapse
    assert B((0, 0, 3, 0), 3) == (0, 0)            # a blank separates the two 0s
    assert B((0, 0, 0, 3), 3) == (0,)              # a run merges
    assert B((3, 3, 3), 3) == ()
    assert B((1, 3, 3, 1, 1, 2), 3) == (1, 1, 2)
    # two frames, C = 2, label (0): paths 00, 0b, b0 -> p0 p0' + p0 pb' + pb p0'
    y = np.array([[[0.3, 0.7], [0.6, 0.4]]])
    p = ctc_enum.frame_probs(y[0])
    want = -np.log(p[0, 0] * p[1, 0] + p[0, 0] * p[1, 1] + p[0, 1] * p[1, 0])
    got = ctc_enum.ctc_cost_enum(y, np.array([[0]]), np.array([2]), np.array([1]))[0, 0]
    assert abs(got - want) < 1e-14
    # Keras' double normalisation is NOT the identity: softmax(log(y + eps)) = (y + eps) / (1 + C eps)
    assert np.allclose(p, (y[0] + 1e-7) / (1 + 2e-7), rtol=0, atol=1e-14) and not np.array_equal(p, y[0])


@pytest.mark.parametrize('case', CASES, ids=IDS)
def test_recursion_restatements_equal_the_enumerated_cost_and_gradient(case):
    from oracle import keras_standin, ref_model
    from qcnn_amd.layers import ctc_batch_cost
    name, y, labels, il, ll = case
    want = ctc_enum.ctc_cost_enum(y, labels, il, ll)
    up = 0.5 + np.random.RandomState(1).rand(y.shape[0])
    g_want = ctc_enum.ctc_grad_fd(y, labels, il, ll, upstream=up)
    finite = np.isfinite(want[:, 0])
    assert finite.any()
    K = keras_standin._build_backend()           # the torch-float64 keras.backend the golden fixtures were generated through
    forms = {
        'oracle.ref_model.ctc_cost': lambda p: ref_model.ctc_cost(p, torch.tensor(labels), torch.tensor(il), torch.tensor(ll)).reshape(-1, 1),
        'qcnn_amd.layers.ctc_batch_cost (torch path)': lambda p: ctc_batch_cost(p, torch.tensor(labels), torch.tensor(il).reshape(-1, 1),
                                                                                 torch.tensor(ll).reshape(-1, 1)),
    }
    forms['keras_standin K.ctc_batch_cost'] = lambda p: K.ctc_batch_cost(torch.tensor(labels), p, torch.tensor(il).reshape(-1, 1),
                                                                        torch.tensor(ll).reshape(-1, 1))
    for what, fn in forms.items():
        p = torch.tensor(y, dtype=torch.float64, requires_grad=True)
        cost = fn(p)
        got = cost.detach().numpy().reshape(-1, 1)
        assert np.array_equal(np.isfinite(got[:, 0]), finite), (what, got, want)          # infeasible labellings: +inf in both
        assert np.all(got[~finite, 0] > 0) if (~finite).any() else True
        f32 = what.startswith('qcnn_amd')                  # the package's torch path computes in float32 whatever it is given
        assert np.abs(got[finite] - want[finite]).max() <= (2e-6 * np.abs(want[finite]).max() if f32 else 1e-10), (what, np.abs(got[finite] - want[finite]).max())
        w = torch.tensor(np.where(finite, up, 0.0))
        (torch.where(torch.isfinite(cost.reshape(-1)), cost.reshape(-1), torch.zeros_like(w)) * w).sum().backward()
        g = p.grad.numpy()
        err = np.abs(g[finite] - g_want[finite]).max() / np.abs(g_want[finite]).max()
        assert err <= (1e-4 if f32 else 1e-6), (what, err)
        for b in range(y.shape[0]):                                                        # frames past input_length: no gradient
            assert np.all(g[b, int(il[b]):] == 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16], ids=['fp32', 'bf16', 'fp16'])
@pytest.mark.parametrize('form', ['concurrent_sweeps', 'two_sweeps'])
@pytest.mark.parametrize('case', CASES, ids=IDS)
def test_fused_ctc_kernels_equal_the_enumerated_cost_and_gradient(case, form, dtype):
    """qk_ctc_batch_cost (csrc/qk_ctc.hip: k_ctc_fast and the two-sweep k_ctc) against the enumeration on the operands the kernel
    sees (posteriors rounded to the activation dtype first)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from qcnn_amd import _lib
    from qcnn_amd import functional as Fq
    dev = torch.device('cuda:0')
    name, y, labels, il, ll = case
    yd = torch.tensor(y, dtype=torch.float32).to(dtype)
    y_seen = yd.double().numpy()
    want = ctc_enum.ctc_cost_enum(y_seen, labels, il, ll)
    up = 0.5 + np.random.RandomState(1).rand(y.shape[0])
    g_want = ctc_enum.ctc_grad_fd(y_seen, labels, il, ll, upstream=up)
    finite = np.isfinite(want[:, 0])
    if dtype == torch.float16 and np.nanmax(np.abs(g_want[finite])) > 6e4:
        pytest.skip('d cost / d y_pred = (...) / (y_pred + 1e-7) leaves float16\'s range at near-zero posteriors (no loss scaling can help a STORED gradient)')
    with _lib.debug_flags(_lib.QK_DBG_CTC_TWO_SWEEPS if form == 'two_sweeps' else 0):
        p = yd.to(dev).requires_grad_(True)
        assert Fq.ctc_supported(p, torch.tensor(labels))
        cost = Fq.ctc_batch_cost(p, torch.tensor(labels), torch.tensor(il).reshape(-1, 1), torch.tensor(ll).reshape(-1, 1))
        w = torch.tensor(np.where(finite, up, 0.0), dtype=torch.float32, device=dev).reshape(-1, 1)
        (torch.where(torch.isfinite(cost), cost, torch.zeros_like(cost)) * w).sum().backward()
        torch.cuda.synchronize()
    got = cost.detach().double().cpu().numpy()
    assert got.shape == want.shape and np.array_equal(np.isfinite(got[:, 0]), finite), (got, want)
    err_c = np.abs(got[finite] - want[finite]).max() / max(np.abs(want[finite]).max(), 1e-30)
    assert err_c <= 2e-5, err_c
    g = p.grad.double().cpu().numpy()
    err_g = np.abs(g[finite] - g_want[finite]).max() / np.abs(g_want[finite]).max()
    tol = {torch.float32: 5e-4, torch.bfloat16: 1e-2, torch.float16: 2e-3}[dtype]
    assert err_g <= tol, err_g
    assert np.all(g[~finite] == 0.0)                                                       # infeasible samples send nothing back
    for b in range(y.shape[0]):
        assert np.all(g[b, int(il[b]):] == 0.0)
