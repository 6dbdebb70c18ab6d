"""TIMIT quaternion CNN + CTC -- counterpart of models/interspeech_model.py:getTimitModel2D (:45-185).

Input (B, 4, 41, T) channels_first quaternion features; QuaternionConv2D(sf,(3,5),'same') ->
MaxPooling2D((1,3),'same') (frequency 41 -> 14, see layers.py) -> n/2 convs of sf filters ->
n/2 convs of 2*sf filters (PReLU + Dropout after each) -> Permute/reshape to (B, T, C*F) ->
3 x TimeDistributed(QuaternionDense(256)) -> TimeDistributed(Dense(62, softmax)) -> CTC.
`d` is the reference's attribute bag (num_layers, start_filter, act, aact, dropout, l2, model,
quat_init); only the quaternion branch (`d.model == 'quaternion'`) is built.
"""

import torch

from .. import _lib as L
from ..complexnn import QuaternionConv2D, QuaternionDense
from ..keras_like import Layer, regularizers
from ..layers import Dense, Dropout, MaxPooling2D, PReLU, TimeDistributed, ctc_batch_cost


class TimitQCNN(torch.nn.Module):
    def __init__(self, num_layers=10, start_filter=32, act='relu', aact='none', dropout=0.0, l2=0.0,
                 quat_init='quaternion', internal_layout='channels_last', fuse_head=True, chain_convs=True):
        super(TimitQCNN, self).__init__()
        n, sf = num_layers, start_filter
        if aact != 'none':
            act = 'linear'                                   # interspeech_model.py:55-56
        reg = regularizers.l2(l2) if l2 else None
        conv_args = dict(activation=act, data_format='channels_first', padding='same', bias_initializer='zeros',
                         kernel_regularizer=reg, kernel_initializer=quat_init, use_bias=True,
                         internal_layout=internal_layout)
        dense_args = dict(activation=act, kernel_regularizer=reg, kernel_initializer='random_uniform',
                          bias_initializer='zeros', use_bias=True)
        self.aact, self.rate, self.act = aact, dropout, act
        self.fuse_head = fuse_head          # first TimeDistributed dense as an (F, 1) convolution (no transpose copy)
        self.chain_convs = (chain_convs and internal_layout == 'channels_last'      # body convs as one autograd node
                            and not L.dbg(L.QK_DBG_NO_CONV_CHAIN))
        self.conv = QuaternionConv2D(sf, (3, 5), name='conv', **conv_args)
        self.pool = MaxPooling2D(pool_size=(1, 3), padding='same')
        widths = [sf] * (n // 2) + [2 * sf] * (n // 2)
        self.convs = torch.nn.ModuleList([QuaternionConv2D(w, (3, 5), name='conv%d' % i, **conv_args)
                                          for i, w in enumerate(widths)])
        self.dense = torch.nn.ModuleList([TimeDistributed(QuaternionDense(256, **dense_args)) for _ in range(3)])
        n_act = 1 + len(widths) + 3
        self.prelu = torch.nn.ModuleList([PReLU(shared_axes=[1, 0]) for _ in range(n_act)]) if aact == 'prelu' else None
        self.drop = Dropout(dropout)
        self._drop_base, self._drop_calls, self._dev = 0, 0, None
        # optional: a one-element int32 device tensor (the step counter of functional.adam_step(step=<tensor>)) mixed into every
        # dropout seed ON THE DEVICE -- with it the launch arguments of a training step do not change from step to step, which
        # is what a captured graph needs (bench.ModelTrainStep.capture); fused post-op path only
        self.drop_step_dev = None
        self.pred = TimeDistributed(Dense(62, activation='softmax', kernel_regularizer=reg, use_bias=True,
                                          bias_initializer='zeros', kernel_initializer='random_uniform'))

    def _act(self, x, i):
        return self.prelu[i](x) if self.prelu is not None else x

    # ---- PReLU / Dropout fused into the kernels (functional.quaternion_conv_chain post-ops) ----------------------
    def _new_drop_base(self):
        """Base seed of this forward pass's dropout masks: drawn from torch's (CPU) generator, so `torch.manual_seed`
        controls the masks as it controls torch's own dropout, and mixed with the data-parallel rank -- replicas seeded
        alike must not drop the same units.  (The fused kernels apply round(rate * 256) / 256: 8 random bits per
        element, functional.PostOp.applied_rate; 0.3 -> 0.30078.)"""
        base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if (self.training and self.rate > 0) else 0
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        self._drop_base, self._drop_calls = (base ^ (rank * 0x9E3779B1)) & 0xffffffff, 0

    def _post(self, k, out_shape, dropout=True):
        """Post-op spec of activation slot k behind a quaternion layer whose (channels_first / TimeDistributed) output
        shape is `out_shape`: the PReLU slopes of self.prelu[k] (built here: (1, F, 1) behind a convolution, i.e. one
        per position of spatial axis 0 of the channels-last buffer; (1, 1) behind a dense layer) -- or alpha=None, the
        relu form, for the aact='none' model -- and the dropout rate while training, with a fresh mask seed per call."""
        rate = self.rate if (dropout and self.training) else 0.0
        self._drop_calls += 1
        seed = (self._drop_base + 7919 * self._drop_calls) & 0xffffffff
        if self.prelu is None:
            return dict(alpha=None, alpha_axis=-1, rate=rate, seed=seed, seed_dev=self.drop_step_dev if rate > 0 else None)
        pl = self.prelu[k]
        if not pl.built:
            pl._build_device = self._dev
            pl.build(tuple(out_shape))
        axis = 0 if pl.alpha.numel() > 1 else -1
        return dict(alpha=pl.alpha, alpha_axis=axis, rate=rate, seed=seed, seed_dev=self.drop_step_dev if rate > 0 else None)

    def forward(self, x):
        self._dev = x.device
        fusable = self.chain_convs and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16, torch.float32)
        if self.prelu is not None and fusable and not L.dbg(L.QK_DBG_NO_FUSED_PRELU):
            return self._forward_fused_post(x)
        # aact == 'none': relu layers with Dropout(d.dropout) behind every body convolution and the first two dense
        # layers (interspeech_model.py:117-121,131-137,150-154) -- relu + dropout fused into the producing kernels
        if (self.prelu is None and self.act == 'relu' and self.training and self.rate > 0 and fusable
                and not L.dbg(L.QK_DBG_NO_FUSED_DROPOUT)):
            return self._forward_fused_post(x)
        o = self._first_layer_fused(x) if self.prelu is None else None
        if o is None:
            o = self._act(self.conv(x), 0)
            o = self.pool(o)
        k = 1
        plain = self.prelu is None and not (self.training and self.rate > 0)
        first = 0
        if self.chain_convs and o.is_cuda and plain and len(self.convs) > 1:
            o, with_head = self._convs_as_chain(o)
            k += len(self.convs)
            if with_head:                                    # first TimeDistributed dense ran inside the chain
                k, first = k + 1, 1
        else:
            for c in self.convs:
                o = self.drop(self._act(c(o), k))
                k += 1
        if first == 0 and self.fuse_head and o.is_cuda:
            o = self._act(self._head_as_conv(o), k)
            o = self.drop(o)
            k, first = k + 1, 1
        elif first == 0:
            o = o.permute(0, 3, 1, 2)                        # Permute((3,1,2)): (B, T, C, F)
            o = o.reshape(o.shape[0], o.shape[1], o.shape[2] * o.shape[3])
        for i in range(first, len(self.dense)):
            o = self._act(self.dense[i](o), k)
            k += 1
            if i < 2:
                o = self.drop(o)
        return self._pred(o)

    def _pred(self, o):
        """The output layer on the (B, T, 256) features -- or the features themselves while ctc_mean_loss collects them for the fused
        output-layer + cost node."""
        if getattr(self, '_features_only', False):
            return o
        return self.pred(o)

    def _first_layer_fused(self, x):
        """conv (3,5) 'same' relu + MaxPooling2D((1,3), 'same') over the frequency axis (interspeech_model.py:97-103)
        as ONE kernel per direction (functional.conv_relu_pool): the 41-bin activation is never written.  Returns None
        when the configuration is outside that kernel (the layers then run one by one)."""
        from .. import functional as Fq
        from ..keras_like import activations
        c, pl = self.conv, self.pool
        if L.dbg(L.QK_DBG_NO_FUSED_FIRST) or not x.is_cuda or x.dim() != 4:
            return None
        if not c.built:
            c._build_device = x.device
            c.build(tuple(x.shape))
        # a plain channels_first buffer (how the reference's callers hold the features) is read plane by plane by the
        # kernel itself; a channels-last buffer behind a channels_first view (an upstream engine layer) is taken as it is
        xl, lay = (x, 'channels_first') if x.is_contiguous() else (x.movedim(1, -1), 'channels_last')
        kernel, bias = c.kernel, c.bias            # (any filter count that is a multiple of 8: start_filter = 16 runs the 32-filter blocks half used)
        ok = (activations.serialize(c.activation) == 'relu' and c.padding == 'same' and c.strides == (1, 1) and
              c.dilation_rate == (1, 1) and c.internal_layout == 'channels_last' and pl.pool_size == (1, 3) and
              pl.strides == (1, 3) and pl.padding == 'same' and pl.data_format == 'channels_last' and
              Fq.conv_relu_pool_supported(xl, kernel, 3, lay))
        if not ok:
            return None
        y = Fq.conv_relu_pool(xl, kernel, bias, 3, lay)
        return y.movedim(-1, 1)

    def _forward_fused_post(self, x):
        """The activation + Dropout behind every layer fused into the quaternion kernels.
        aact == 'prelu' (interspeech_model.py:55-56,99-101,117-121: linear layers, PReLU(shared_axes=[1,0]) and Dropout
        behind each): the producing kernel writes the pre-activation and the activated / dropped tensor, the next
        layer's backward-data applies the derivative.
        aact == 'none' with relu layers and an active Dropout (:117-121,131-137): the producing kernel writes ONLY
        y = dropout(relu(pre)) and the next layer's backward-data multiplies its output by (y > 0) / (1 - rate) --
        the same tensors and traffic as the dropout-free relu chain."""
        from .. import functional as Fq
        from ..keras_like import activations
        self._new_drop_base()
        c = self.conv
        if not c.built:
            c._build_device = x.device
            c.build(tuple(x.shape))
        shape = c.compute_output_shape(tuple(x.shape))
        relu_form = self.prelu is None
        post0 = self._post(0, shape, dropout=False)
        pl = self.pool
        xl, lay = (x, 'channels_first') if x.is_contiguous() else (x.movedim(1, -1), 'channels_last')   # see _first_layer_fused
        o = self._first_layer_fused(x) if relu_form else None
        fused_first = (not relu_form and
                       not L.dbg(L.QK_DBG_NO_FUSED_FIRST) and x.dim() == 4 and c.padding == 'same' and c.strides == (1, 1) and
                       c.dilation_rate == (1, 1) and c.internal_layout == 'channels_last' and pl.pool_size == (1, 3) and
                       pl.strides == (1, 3) and pl.padding == 'same' and pl.data_format == 'channels_last' and
                       Fq.conv_prelu_pool_supported(xl, c.kernel, post0['alpha'], post0['alpha_axis'], 3, lay))
        if o is not None:
            pass             # conv + relu + frequency pooling ran as one kernel per direction (qk_conv_relu_pool_*)
        elif relu_form:
            o = self.pool(c(x))
        elif fused_first:    # linear conv + PReLU + frequency pooling as ONE kernel per direction (qk_conv_prelu_pool_*)
            o = Fq.conv_prelu_pool(xl, c.kernel, c.bias, post0['alpha'], post0['alpha_axis'], 3, lay).movedim(-1, 1)
        else:
            o = Fq.quaternion_conv(x, c.kernel, c.bias, strides=c.strides, padding=c.padding, data_format='channels_first',
                                   dilation_rate=c.dilation_rate, activation=None, post=post0)
            o = self.pool(o)
        shape = tuple(o.shape)
        layers = []
        for i, cv in enumerate(self.convs):
            if not cv.built:
                cv._build_device = x.device
                cv.build(shape)
            shape = cv.compute_output_shape(shape)
            layers.append((cv.kernel, cv.bias, dict(strides=cv.strides, padding=cv.padding, dilation_rate=cv.dilation_rate,
                                                    activation=None, post=self._post(1 + i, shape))))
        k = 1 + len(self.convs)
        d0 = self.dense[0].layer
        if not d0.built:
            d0._build_device = x.device
            d0.build((None, shape[1] * shape[2]))
        dl, link = self._head_link(shape, x.device, o.dtype, activation=None, post=self._post(k, (shape[0], shape[3], d0.r.shape[-1])))
        head_shape = (shape[0], dl.r.shape[-1], shape[3])                       # (B, units, T): TimeDistributed output, channels_first view
        layers.append(link)
        k += 1
        if relu_form and not L.dbg(L.QK_DBG_NO_DENSE_IN_CHAIN):
            # aact == 'none': the second and third TimeDistributed(QuaternionDense(256)) (interspeech_model.py:150-166) are links
            # of the SAME chain -- 1 x 1 conj-convolutions on the (B, 1, T, 256) tensor, relu (+ dropout behind the second) as the
            # producing kernel's post-op, its derivative in the consumer's backward-data epilogue: no separate activation /
            # dropout pass in either direction
            width = head_shape[1]
            for i in (1, 2):
                dn = self.dense[i].layer
                if not dn.built:
                    dn._build_device = x.device
                    dn.build((None, width))
                width = dn.r.shape[-1]
                layers.append((dn.r, dn.bias, dict(activation=None, post=self._post(k, (shape[0], shape[3], width), dropout=(i < 2)))))
                k += 1
            y = Fq.quaternion_conv_chain(o.movedim(1, -1), layers)              # (B, 1, T, units)
            return self._pred(y.reshape(y.shape[0], y.shape[2], y.shape[3]))
        y = Fq.quaternion_conv_chain(o.movedim(1, -1), layers)                  # (B, 1, T, units)
        o = y.reshape(y.shape[0], y.shape[2], y.shape[3])
        for i in (1, 2):
            dn = self.dense[i].layer
            b, t = o.shape[0], o.shape[1]
            if not dn.built:
                dn._build_device = x.device
                dn.build((None, o.shape[2]))
            po = self._post(k, (b, t, dn.r.shape[-1]), dropout=(i < 2))
            if relu_form and po['rate'] == 0.0:
                o = Fq.quaternion_dense(o.reshape(b * t, o.shape[2]), dn.r, dn.bias, activation='relu').reshape(b, t, -1)
            else:
                h = Fq.quaternion_dense(o.reshape(b * t, o.shape[2]), dn.r, dn.bias, activation=None).reshape(b, t, -1)
                o = Fq.prelu_dropout(h, po['alpha'], po['alpha_axis'], po['rate'], po['seed'])
            k += 1
        return self._pred(o)

    def _head_kernel(self, o_shape, device):
        """The first dense layer's weight r[(cq*F + f), :] viewed as the (F, 1, Cq, units) kernel of the
        equivalent 'valid' conj-convolution over (F, T) (see _head_as_conv)."""
        dl = self.dense[0].layer
        b, c, f, t = o_shape
        if not dl.built:
            dl._build_device = device
            dl.build((None, c * f))
        return dl, dl.r.view(c // 4, f, dl.r.shape[-1]).permute(1, 0, 2).unsqueeze(1).contiguous()

    def _head_link(self, o_shape, device, dtype, **kw):
        """The head as a chain link (kernel, bias, kwargs).  16-bit device tensors with matrix-core widths: the dense PARAMETER
        itself, read in place as a channel-major kernel (functional.quaternion_conv_chain: dense_kernel_size) -- the permuted
        copy of `_head_kernel`, its re-layout launches and the permuted gradient accumulation (8 small launches per step)
        go away; anything else: the (F, 1, Cq, units) copy."""
        dl, w = None, None
        b, c, f, t = o_shape
        d0 = self.dense[0].layer
        if not d0.built:
            d0._build_device = device
            d0.build((None, c * f))
        cq, fq = c // 4, d0.r.shape[-1] // 4
        if (dtype in (torch.bfloat16, torch.float16) and str(device).startswith('cuda') and cq % 32 == 0 and fq % 32 == 0
                and d0.r.is_contiguous()):
            return d0, (d0.r, d0.bias, dict(kw, conj=True, dense_kernel_size=(f, 1)))
        dl, w = self._head_kernel(o_shape, device)
        return dl, (w, dl.bias, dict(kw, strides=1, padding='valid', dilation_rate=1, conj=True))

    def _convs_as_chain(self, o):
        """The n body convolutions (interspeech_model.py:105-137 with no advanced activation and no active
        dropout) through functional.quaternion_conv_chain: same values and gradients as calling the layers
        one by one, but each relu derivative is applied where it is cheapest.
        With fuse_head the first TimeDistributed dense layer (as an (F, 1) convolution) is the last link
        of the chain, so the last body convolution's relu derivative also moves into a backward-data
        epilogue.  Returns (tensor, head_included)."""
        from .. import functional as Fq
        from ..keras_like import activations
        shape = tuple(o.shape)
        layers = []
        for c in self.convs:
            if not c.built:
                c._build_device = o.device
                c.build(shape)
            shape = c.compute_output_shape(shape)
            name = activations.serialize(c.activation)
            if name not in ('linear', 'relu'):
                return self._convs_one_by_one(o), False
            layers.append((c.kernel, c.bias, dict(strides=c.strides, padding=c.padding,
                                                  dilation_rate=c.dilation_rate, activation=name)))
        with_head = False
        if self.fuse_head:
            name = activations.serialize(self.dense[0].layer.activation)
            if name in ('linear', 'relu'):
                dl, link = self._head_link(shape, o.device, o.dtype, activation=name)
                layers.append(link)
                with_head = True
        y = Fq.quaternion_conv_chain(o.movedim(1, -1), layers)       # (B, F, T, C) channels-last buffer
        if with_head:
            return y.reshape(y.shape[0], y.shape[2], y.shape[3]), True   # (B, 1, T, units) -> (B, T, units)
        return y.movedim(-1, 1), False

    def _convs_one_by_one(self, o):
        for c in self.convs:
            o = c(o)
        return o

    def _head_as_conv(self, o):
        """Permute((3,1,2)) + reshape + TimeDistributed(QuaternionDense) (interspeech_model.py:141-149) on
        the conv output (B, C, F, T) WITHOUT the 367 MB transpose copy: feature c*F + f of time step t
        is o[b, c, f, t], so the dense layer is a quaternion convolution with an (F, 1) 'valid' kernel
        over (F, T) -- same GEMM (K = F*C), same conj(W) (x) x table -- whose kernel is the dense weight
        r[(cq*F + f), :] re-indexed to [f, 0, cq, :].  Parameters stay the reference's (in_q, units)."""
        from .. import functional as Fq
        from ..keras_like import activations
        b, c, f, t = o.shape
        dl, w = self._head_kernel(tuple(o.shape), o.device)
        name = activations.serialize(dl.activation)
        fused = name if name in ('linear', 'relu') else 'linear'
        y = Fq.quaternion_conv(o, w, dl.bias, 1, 'valid', 'channels_first', 1, fused, conj=True)
        if fused != name:
            y = dl.activation(y)
        return y.reshape(b, dl.r.shape[-1], t).permute(0, 2, 1)                      # (B, units, 1, T) -> (B, T, units)

    def ctc_loss(self, x, labels, input_length, label_length, loss_scale=1.0):
        """The model output of the reference: K.ctc_batch_cost per sample, shape (B, 1) (interspeech_model.py:178).
        loss_scale (float16 training): the gradient sent back through the network is multiplied by it, the returned cost is
        not; divide it out in the optimiser (functional.adam_step(grad_scale=1 / (world * loss_scale)))."""
        return ctc_batch_cost(self(x), labels, input_length, label_length, loss_scale=loss_scale)

    def ctc_mean_loss(self, x, labels, input_length, label_length, loss_scale=1.0):
        """mean over the batch of ctc_loss(...): what training minimises (the reference compiles the model with
        `loss={'ctc': lambda y_true, y_pred: y_pred}`, i.e. the mean of the per-sample costs of interspeech_model.py:178).  Same value
        and gradients as `self.ctc_loss(...).mean()`; on the device the output layer, the CTC cost and the mean are ONE autograd node
        (layers.dense_softmax_ctc_mean: no framework launch between the CTC kernel and the output layer's backward)."""
        from ..layers import dense_softmax_ctc_mean
        dense = self.pred.layer
        if x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and hasattr(self, '_pred'):
            self._features_only = True
            try:
                feats = self(x)
            finally:
                self._features_only = False
            if not dense.built:
                dense._build_device = x.device
                dense.build((None, feats.shape[-1]))
            loss = dense_softmax_ctc_mean(feats, dense, labels, input_length, label_length, loss_scale) if feats.dim() == 3 else None
            if loss is not None:
                return loss
            return ctc_batch_cost(self.pred(feats), labels, input_length, label_length, loss_scale=loss_scale).mean()
        return self.ctc_loss(x, labels, input_length, label_length, loss_scale=loss_scale).mean()

    def regularization_loss(self):
        """Sum of the kernel regularisers (l2(d.l2) on every conv / dense kernel, interspeech_model.py:63,68,173):
        the term Keras adds to the compiled model's loss on top of the CTC cost."""
        terms = [t for m in self.modules() if isinstance(m, Layer) for t in m.regularization_losses()]
        if not terms:
            return next(self.parameters()).new_zeros(())
        return torch.stack([t.float() for t in terms]).sum()

    def training_loss(self, x, labels, input_length, label_length, loss_scale=1.0):
        """What training the reference model minimises: mean CTC cost over the batch (the usual
        `loss={'ctc': lambda y_true, y_pred: y_pred}` compile) + the regularisation terms.  loss_scale changes GRADIENTS only:
        the regulariser goes through the same identity-forward / scaled-backward node as the CTC cost, so that ONE grad_scale in
        the optimiser undoes both and the value returned (what gets logged) is the unscaled loss."""
        reg = self.regularization_loss()
        if loss_scale != 1.0:
            from ..layers import _GradScale
            reg = _GradScale.apply(reg, float(loss_scale))
        return self.ctc_mean_loss(x, labels, input_length, label_length, loss_scale=loss_scale) + reg


class _RealConv2D(Layer):
    """keras Conv2D(filters, (3, 5), data_format='channels_first', padding='same') of the reference's `d.model == "real"`
    branch (interspeech_model.py:92-96,109-113,124-128): a stock real-valued layer, outside the quaternion hot path -- torch's
    own convolution with TensorFlow's 'same' rule (odd kernels: symmetric padding)."""

    def __init__(self, filters, kernel_size, activation=None, kernel_regularizer=None, **kwargs):
        super(_RealConv2D, self).__init__(**kwargs)
        from ..keras_like import activations, initializers
        self.filters, self.kernel_size = filters, tuple(kernel_size)
        self.activation = activations.get(activation)
        self.kernel_regularizer = regularizers.get(kernel_regularizer)
        self._init = initializers.get('random_uniform')
        self._zeros = initializers.get('zeros')

    def build(self, input_shape):
        self.add_weight('kernel', self.kernel_size + (input_shape[1], self.filters), initializer=self._init,
                        regularizer=self.kernel_regularizer)                 # Keras layout (kh, kw, in, out)
        self.add_weight('bias', (self.filters,), initializer=self._zeros)
        self.built = True

    def call(self, inputs):
        w = self.kernel.permute(3, 2, 0, 1).to(inputs.dtype)
        y = torch.nn.functional.conv2d(inputs, w, self.bias.to(inputs.dtype), padding=(self.kernel_size[0] // 2, self.kernel_size[1] // 2))
        return self.activation(y)

    def compute_output_shape(self, input_shape):
        return (input_shape[0], self.filters) + tuple(input_shape[2:])


class TimitRealCNN(torch.nn.Module):
    """The `d.model == "real"` branch of getTimitModel2D (interspeech_model.py:92-96,109-113,124-128,159-169): Conv2D stack
    on (B, 3, 41, T), the same frequency pooling, three TimeDistributed(Dense(1024)), Dense(62, softmax), CTC.  Stock
    real-valued layers throughout (torch ops): the comparison network of the paper, not part of the Hamilton hot path."""

    def __init__(self, num_layers=10, start_filter=32, act='relu', aact='none', dropout=0.0, l2=0.0):
        super(TimitRealCNN, self).__init__()
        n, sf = num_layers, start_filter
        if aact != 'none':
            act = 'linear'
        reg = regularizers.l2(l2) if l2 else None
        self.conv = _RealConv2D(sf, (3, 5), activation=act, kernel_regularizer=reg)
        self.pool = MaxPooling2D(pool_size=(1, 3), padding='same')
        widths = [sf] * (n // 2) + [2 * sf] * (n // 2)
        self.convs = torch.nn.ModuleList([_RealConv2D(w, (3, 5), activation=act, kernel_regularizer=reg) for w in widths])
        dense_args = dict(activation=act, kernel_regularizer=reg, kernel_initializer='random_uniform', bias_initializer='zeros', use_bias=True)
        self.dense = torch.nn.ModuleList([TimeDistributed(Dense(1024, **dense_args)) for _ in range(3)])
        self.prelu = torch.nn.ModuleList([PReLU(shared_axes=[1, 0]) for _ in range(1 + len(widths) + 3)]) if aact == 'prelu' else None
        self.drop = Dropout(dropout)
        self.pred = TimeDistributed(Dense(62, activation='softmax', kernel_regularizer=reg, use_bias=True,
                                          bias_initializer='zeros', kernel_initializer='random_uniform'))

    def _act(self, x, i):
        return self.prelu[i](x) if self.prelu is not None else x

    def forward(self, x):
        o = self.pool(self._act(self.conv(x), 0))
        k = 1
        for c in self.convs:
            o = self.drop(self._act(c(o), k))
            k += 1
        o = o.permute(0, 3, 1, 2)
        o = o.reshape(o.shape[0], o.shape[1], o.shape[2] * o.shape[3])
        for i, dl in enumerate(self.dense):
            o = self._act(dl(o), k)
            k += 1
            if i < 2:
                o = self.drop(o)
        return self.pred(o)

    ctc_loss = TimitQCNN.ctc_loss
    ctc_mean_loss = TimitQCNN.ctc_mean_loss
    regularization_loss = TimitQCNN.regularization_loss
    training_loss = TimitQCNN.training_loss


def getTimitModel2D(d):
    """(model, val_function) like the reference: `model(x)` gives the (B, T, 62) posteriors,
    `model.ctc_loss(...)` the CTC cost of interspeech_model.py:178, `model.training_loss(...)` that cost
    averaged over the batch plus the l2 terms Keras adds (d.l2); val_function(x) == model(x).
    d.model == 'quaternion' builds the engine's TimitQCNN on (B, 4, 41, T); d.model == 'real' the stock-layer comparison
    network on (B, 3, 41, T) (TimitRealCNN)."""
    kind = getattr(d, 'model', 'quaternion')
    if kind == 'real':
        m = TimitRealCNN(d.num_layers, d.start_filter, d.act, d.aact, d.dropout, getattr(d, 'l2', 0.0))
        return m, (lambda x: m(x))
    if kind != 'quaternion':
        raise ValueError("d.model must be 'quaternion' or 'real', got %r" % (kind,))
    m = TimitQCNN(d.num_layers, d.start_filter, d.act, d.aact, d.dropout, getattr(d, 'l2', 0.0),
                  getattr(d, 'quat_init', 'quaternion'))
    return m, (lambda x: m(x))
