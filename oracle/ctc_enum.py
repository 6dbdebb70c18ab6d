"""TEST INFRASTRUCTURE -- K.ctc_batch_cost by brute-force enumeration of alignments (float64, definition level).

The reference's loss (/root/reference/models/interspeech_model.py:37-39: `K.ctc_batch_cost(labels, y_pred, input_length,
label_length)`, wired at :178) is pinned elsewhere to a *recursion* (torch.nn.functional.ctc_loss behind oracle/ref_model.ctc_cost
and oracle/keras_standin.py, the alpha/beta sweeps of csrc/qk_ctc.hip).  This file pins the MATH independently of every recursion:

    cost[b] = -log  sum over all frame-level paths pi in {0..C-1}^Tn that collapse to labels_b   prod_t  p_t(pi_t)

  * collapse B(pi): merge runs of equal symbols, THEN drop blanks (ctc_merge_repeated=True, the TF default Keras uses);
    blank = C - 1 (tf.nn.ctc_loss's convention, which K.ctc_batch_cost keeps);
  * per-frame probabilities, the Keras 2.x / TensorFlow recipe: K.ctc_batch_cost hands log(y_pred + epsilon()) to tf.nn.ctc_loss
    as LOGITS and that op applies softmax to its inputs again:  p_t = softmax(log(y_pred[t] + 1e-7));
  * only the first Tn = input_length[b] frames count, only the first label_length[b] labels.

C^Tn paths: for T <= 6, C <= 4 at most 4096 per sample.  Only tests/ import this (never the product, never the bench).
"""
import itertools
import math

import numpy as np

EPSILON = 1e-7          # keras.backend.epsilon()


def collapse(path, blank):
    """B(pi): merge repeated symbols, then remove blanks."""
    out, prev = [], None
    for c in path:
        if c != prev and c != blank:
            out.append(c)
        prev = c
    return tuple(out)


def frame_probs(y_pred_b, eps=EPSILON):
    """softmax(log(y + eps)) per frame, float64: what tf.nn.ctc_loss sees when Keras feeds it log(y_pred + eps) as logits."""
    u = np.log(np.asarray(y_pred_b, dtype=np.float64) + eps)
    u = u - u.max(axis=-1, keepdims=True)
    e = np.exp(u)
    return e / e.sum(axis=-1, keepdims=True)


def ctc_cost_enum(y_pred, labels, input_length, label_length, eps=EPSILON):
    """Per-sample cost (B, 1), float64.  y_pred (B, T, C) softmax outputs; labels (B, Lmax) ints; lengths (B,) or (B, 1).
    A labelling no path collapses to (too few frames) costs +inf; zero frames and zero labels cost 0 (one empty path)."""
    y = np.asarray(y_pred, dtype=np.float64)
    B, T, C = y.shape
    blank = C - 1
    il = np.asarray(input_length).reshape(-1)
    ll = np.asarray(label_length).reshape(-1)
    lab = np.asarray(labels)
    out = np.zeros((B, 1), dtype=np.float64)
    for b in range(B):
        tn, ln = int(il[b]), int(ll[b])
        assert 0 <= tn <= T and C ** tn <= 1 << 16, 'enumeration is for tiny cases only'
        target = tuple(int(v) for v in lab[b, :ln])
        p = frame_probs(y[b, :tn], eps)
        total = 0.0
        for path in itertools.product(range(C), repeat=tn):
            if collapse(path, blank) == target:
                pr = 1.0
                for t, c in enumerate(path):
                    pr *= p[t, c]
                total += pr
        out[b, 0] = -math.log(total) if total > 0.0 else math.inf
    return out


def ctc_grad_fd(y_pred, labels, input_length, label_length, upstream=None, h=1e-4, eps=EPSILON):
    """d sum_b upstream[b] * cost[b] / d y_pred by central finite differences of the ENUMERATED cost (float64): the gradient
    TensorFlow's autodiff hands back through K.ctc_batch_cost, with no recursion and no hand-derived formula in between.
    The step is RELATIVE, h * (y + eps) per entry (the cost depends on y through log(y + eps): a fixed step would leave the domain at
    near-zero posteriors).  Entries of samples whose cost is infinite are returned as NaN (nothing to differentiate)."""
    y = np.asarray(y_pred, dtype=np.float64)
    B, T, C = y.shape
    up = np.ones(B) if upstream is None else np.asarray(upstream, dtype=np.float64).reshape(-1)
    g = np.zeros_like(y)
    il = np.asarray(input_length).reshape(-1)
    for b in range(B):
        args = (labels[b:b + 1], input_length[b:b + 1], label_length[b:b + 1])
        if not np.isfinite(ctc_cost_enum(y[b:b + 1], *args, eps=eps)[0, 0]):
            g[b] = np.nan
            continue
        for t in range(int(il[b])):                # frames past the input length do not enter the cost: gradient exactly 0
            for c in range(C):
                yp, ym = y[b:b + 1].copy(), y[b:b + 1].copy()
                d = h * (y[b, t, c] + eps)
                yp[0, t, c] += d
                ym[0, t, c] -= d
                g[b, t, c] = up[b] * (ctc_cost_enum(yp, *args, eps=eps)[0, 0] - ctc_cost_enum(ym, *args, eps=eps)[0, 0]) / (2 * d)
    return g


def tiny_cases(seed=0):
    """The cases both tests run: (name, y_pred (B, T, C), labels, input_length, label_length).  Repeated labels (a blank must sit
    between them), an empty label sequence, input_length < T, a labelling that exactly fills its frames, one that cannot fit,
    C = 2 (one real class), near-one-hot and near-uniform posteriors."""
    rng = np.random.RandomState(seed)

    def soft(*shape, scale=1.5):
        z = rng.randn(*shape) * scale
        e = np.exp(z - z.max(-1, keepdims=True))
        return e / e.sum(-1, keepdims=True)
    cases = []
    cases.append(('t6_c4_mixed', soft(6, 6, 4),
                  np.array([[0, 1, 2], [1, 1, 0], [2, 0, 0], [0, 0, 0], [2, 2, 2], [1, 0, 1]]),
                  np.array([6, 6, 4, 5, 5, 3]), np.array([3, 2, 1, 0, 3, 3])))         # [4]: 2 2 2 needs 5 frames, has 5; [5]: 1 0 1 in 3 frames
    cases.append(('t5_c3_repeats', soft(4, 5, 3),
                  np.array([[0, 0], [1, 1], [0, 1], [1, 0]]), np.array([5, 3, 2, 5]), np.array([2, 2, 2, 1])))     # [1]: exactly blank-separated
    cases.append(('t4_c2_one_class', soft(3, 4, 2), np.array([[0, 0], [0, 0], [0, 0]]), np.array([4, 3, 4]), np.array([1, 2, 0])))
    cases.append(('t6_c4_peaky', soft(3, 6, 4, scale=6.0), np.array([[0, 1], [2, 2], [1, 0]]), np.array([6, 6, 5]), np.array([2, 2, 1])))
    cases.append(('t3_c4_infeasible', soft(3, 3, 4), np.array([[0, 1, 2], [1, 1, 0], [0, 1, 2]]), np.array([3, 2, 2]), np.array([3, 2, 3])))   # [1], [2] do not fit
    cases.append(('t6_c3_flat', soft(2, 6, 3, scale=0.01), np.array([[0, 1, 0], [1, 1, 1]]), np.array([6, 6]), np.array([3, 3])))
    return cases
