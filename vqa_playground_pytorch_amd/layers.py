"""Host-side mirror of the reference's layer library for the CoR2 / ODA hot path.

Same class names, constructor arguments, parameter names/shapes and error
behaviour as the reference, so its checkpoints load and its call sites read the same:

  Linear, MutanFusion, bmul, bmatmul   <- putils/__init__.py:16-33, :205-241, :98-104, :89-95
  MyConv1d, MyLinear, MyATT            <- config/CoR2.py:56-157 (identical copies in config/ODA.py:73-174)

What differs is underneath: the python per-sample loops are gone; MutanFusion's region side, the
attention softmax+pooling, the region projections (MyConv1d: K5, dropout + GEMM + bias + relu in one
kernel) and (in the models) the pairwise relation / object-difference tensors run in hand-written
HIP kernels through libvqa_mi355x.so (ops.py); the [B,.]-sized MyLinear layers run as grouped
phases (head.py, K6) -- except, under VQA_HEAD=auto, CoR2's glimpse projections and ODA's [B,.]
layers, which are one batched library GEMM each with HIP epilogues (my_linears below).
On GPU tensors every layer takes the HIP path and raises if the library is missing; the plain torch
expressions in this file (bmul, bmatmul, _activation, the non-CUDA branches) exist for CPU tensors
only -- shape checks and the state_dict tests -- and are never part of a measured step.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def bmul(inputs1, inputs2):
    """putils/__init__.py:98-104: out[b] = inputs1[b] * inputs2[b] (inputs2 broadcasts over the
    middle axes of inputs1) -- one fused broadcast multiply instead of B launches + stack."""
    b = inputs1.size(0)
    shape = [b] + [1] * (inputs1.dim() - inputs2.dim()) + list(inputs2.shape[1:])
    return inputs1 * inputs2.reshape(shape)


def bmatmul(inputs1, inputs2):
    """putils/__init__.py:89-95: out[b] = inputs1[b] @ inputs2[b]."""
    return torch.matmul(inputs1, inputs2)


def _activation(x, af, dim):
    if not af:
        return x
    if af == "softmax":
        return F.softmax(x, dim=dim)
    if af == "sigmoid":
        return torch.sigmoid(x)
    if af == "tanh":
        return torch.tanh(x)
    return getattr(F, af)(x)


def _same_config(mods, keys):
    first = mods[0]
    return all(all(getattr(m, k) == getattr(first, k) for k in keys) for m in mods)


def my_linears(mods, x, group_first=False, predropped=False):
    """[m(x_g) for m in mods] for G same-shaped MyLinear / Linear modules as ONE batched GEMM: x is [B,K] (every module
    reads the same input, each with its own dropout draw, as when the reference calls them one after the other) or
    [B,G,K] (module g reads x[:, g, :]).  Returns [B,G,A], or [G,B,A] when group_first (each result contiguous).
    The modules keep their own parameters (state_dict names unchanged); weights are stacked per call.
    predropped: the producer of x ([B,G,K]) has already applied the modules' input dropout."""
    first = mods[0]
    lins = [m.linear for m in mods]
    p = getattr(first, "p", None)
    af = getattr(first, "af", None)
    if predropped and x.dim() != 3:
        raise ValueError("my_linears: predropped needs one input per module ([B,G,K])")
    ok = x.is_cuda and _same_config(mods, ("in_features", "out_features")) and \
        all(getattr(m, "p", None) == p and getattr(m, "af", None) == af for m in mods) and \
        all((l.bias is None) == (lins[0].bias is None) for l in lins)
    G = len(mods)
    if not ok:
        outs = [m(x if x.dim() == 2 else x[:, g, :]) for g, m in enumerate(mods)]
        return torch.stack(outs, 0 if group_first else 1)
    if x.size(-1) != first.in_features:
        raise ValueError(
            "[error] putils.Linear(%s, %s): last dimension of input(%s) should equal to in_features(%s)"
            % (first.in_features, first.out_features, x.size(-1), first.in_features))
    training = getattr(first, "training", False)
    if x.dim() == 2 and not group_first and not (p and training) and af in (None, "") and x.dtype == torch.float32:
        # G layers on ONE input without dropout or activation (Mutan's rank factors, putils/__init__.py:232-238): a
        # plain GEMM against the [G*A, K] stack -- the sum over the groups in the data gradient happens inside the GEMM
        # (no expand / reduce pair around a batched one), the bias gradient is one column sum
        w = ops.stack_params([l.weight for l in lins])
        if w.is_contiguous():
            b = None
            if lins[0].bias is not None:       # (through ops.linear: its bias gradient is the replay-safe column sum)
                b = ops.stack_params([l.bias for l in lins]).reshape(G * first.out_features)
            y = ops.linear(x, w.view(G * first.out_features, first.in_features), b)
            return y.view(x.size(0), G, first.out_features)
    drop = bool(p) and training and not predropped
    if x.dim() == 2 and drop and x.dtype == torch.float32:
        x = ops.dropout(x, p, groups=G).transpose(0, 1)              # G independent draws over the one input: [B,G,K] view
        drop = False
    elif x.dim() == 2:
        x = x.unsqueeze(1).expand(x.size(0), G, x.size(1))          # stride-0 group axis: no copy
    if drop:
        # one draw over [B,G,K]: G independent masks (hash mask on the GPU fp32 path; torch's generator otherwise)
        if x.dtype != torch.float32:
            x = F.dropout(x, p=p, training=True)
        elif x.dim() == 3 and not x.is_contiguous() and x.transpose(0, 1).is_contiguous():
            x = ops.dropout(x.transpose(0, 1), p).transpose(0, 1)    # a [G,B,K] tensor seen as [B,G,K]: mask it as stored
        else:
            x = ops.dropout(x, p)
    w = ops.stack_params([l.weight for l in lins])                   # [G,A,K]: a view of the flat parameter buffer
    b = ops.stack_params([l.bias for l in lins]) if lins[0].bias is not None else None   # when the trainer laid it out
    if af in (None, "", "relu", "sigmoid"):
        return ops.batched_linear(x, w, b, group_first, af or None)      # bias + activation in the GEMM's epilogue kernel
    return _activation(ops.batched_linear(x, w, b, group_first), af, None)


def linear_stack_groups(mods):
    """[[weights...], [biases...]] of same-shaped Linear / MyLinear modules that ``my_linears`` runs as one batched GEMM:
    the trainer places each list at equal spacing in its flat parameter buffer so that stacking them is a view."""
    lins = [m.linear for m in mods]
    groups = [[l.weight for l in lins]]
    if lins[0].bias is not None:
        groups.append([l.bias for l in lins])
    return groups


class Linear(nn.Module):
    """putils/__init__.py:16-33."""

    def __init__(self, in_features, out_features, bias=True, seed=None):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        if seed:
            torch.manual_seed(seed)
        self.linear = nn.Linear(in_features, out_features, bias=bias)

    def forward(self, x):
        if x.size()[-1] != self.in_features:
            raise ValueError(
                "[error] putils.Linear(%s, %s): last dimension of input(%s) should equal to in_features(%s)"
                % (self.in_features, self.out_features, x.size(-1), self.in_features))
        return ops.linear(x, self.linear.weight, self.linear.bias)


class MyLinear(nn.Module):
    """config/CoR2.py:94-122: af(linear(dropout_p(x)))."""

    def __init__(self, in_features, out_features, seed=None, p=None, af=None, dim=None):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.p = p
        self.af = af
        self.dim = dim
        if seed:
            torch.manual_seed(seed)
        self.linear = nn.Linear(in_features, out_features, bias=True)

    def forward(self, x):
        if x.size()[-1] != self.in_features:
            raise ValueError(
                "[error] putils.Linear(%s, %s): last dimension of input(%s) should equal to in_features(%s)"
                % (self.in_features, self.out_features, x.size(-1), self.in_features))
        if self.p and self.training:
            x = ops.dropout(x, self.p) if (x.is_cuda and x.dtype == torch.float32) else F.dropout(x, p=self.p, training=True)
        return _activation(ops.linear(x, self.linear.weight, self.linear.bias), self.af, self.dim)


class MyConv1d(nn.Module):
    """config/CoR2.py:56-91 with kernel_size 1 (the only use): af(conv1d(dropout_p(x)^T)^T) on
    [B,N,Cin].  The parameter keeps nn.Conv1d's (Cout,Cin,1) shape under the name ``conv``; the
    contraction itself is a row-major GEMM on the [B*N,Cin] view (no transposes)."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, seed=None, p=None, af=None,
                 dim=None):
        super().__init__()
        if kernel_size != 1 or stride != 1 or padding != 0:
            raise ValueError("MyConv1d: only kernel_size=1, stride=1, padding=0 is on the CoR2/ODA path")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.p = p
        self.af = af
        self.dim = dim
        if seed:
            torch.manual_seed(seed)
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride, padding=padding, dilation=1, groups=1,
                              bias=True)

    # bf16 path: "engine" = the hand-written bf16 MFMA GEMMs (ops.LinearBf16), "library" = hipBLASLt through F.linear
    bf16_gemm = os.environ.get("VQA_BF16_GEMM", "engine")

    def _linear_bf16(self, x, af, p=0.0, packed=None):
        """Mixed-precision form (x bf16, last dim possibly zero-padded past in_channels): bf16 operands, fp32
        accumulate, fp32 master weights.  Wide layers return the output padded to a multiple of 64 (pad = 0) so the
        next bf16 GEMM needs no tail; narrow ones (the G attention logits) come back as fp32.  p: this layer's input
        dropout rate in the current mode (0 in eval / when the producer dropped x already); at 0.5 it runs inside the GEMM
        kernels (ops.LinearBf16).  packed: the layer's shadows from the model's ShadowPlan, or None."""
        w = self.conv.weight.squeeze(-1)
        if self.out_channels >= 32 and self.bf16_gemm == "engine" and af in (None, "", "relu"):
            if p and (p != 0.5 or x.requires_grad):
                # the in-kernel mask exists at p = 0.5 and has no data gradient (compress_v reads the model input); an x
                # that needs one (compress_v2 behind the pairwise relation, relation_mode=0) is dropped beforehand
                x, p = F.dropout(x, p=p, training=True), 0.0
            return ops.linear_bf16(x, w, self.conv.bias, af, p, ops.next_dropout_seed() if p else 0,
                                   pregated=self.grad_pregated, packed=packed)
        if p:
            x = F.dropout(x, p=p, training=True)
        n_p = ops.pad_to(self.out_channels) if self.out_channels >= 32 else self.out_channels
        wp = ops.PackedWeightBf16.apply(w, n_p, x.size(-1))
        bp = F.pad(self.conv.bias, (0, n_p - self.out_channels)).to(torch.bfloat16)
        y = ops.linear(x, wp, bp)
        if self.out_channels < 32:
            return _activation(y.float(), af, self.dim)
        return _activation(y, af, self.dim)

    def pre_activation(self, x):
        if x.dim() != 3:
            raise ValueError("[error] putils.Conv1d(%s, %s, %s, %s): input_dim (%s) should equal to 3"
                             % (self.in_channels, self.out_channels, self.kernel_size, self.stride, x.dim()))
        if ops.attention_logits_supported(x, self.in_channels, self.out_channels):
            # the G attention logits of MyATT: dropout + 1x1 conv in one HIP kernel (K3a), fp32 out for either storage
            p = self.p if (self.training and self.p) else 0.0
            return ops.attention_logits(x, self.conv.weight.squeeze(-1), self.conv.bias, p,
                                        ops.next_dropout_seed() if p else 0)
        if x.dtype == torch.bfloat16:
            return self._linear_bf16(x, None, self.p if (self.training and self.p) else 0.0)
        if self.p:
            x = F.dropout(x, p=self.p, training=self.training)
        return ops.linear(x, self.conv.weight.squeeze(-1), self.conv.bias)

    # K5 (fused dropout+GEMM+bias+relu on the hand-written fp32 MFMA engines) vs the library GEMM + a separate dropout
    # pass: set per process with VQA_FUSED_LINEAR=0/1.  Default: fused -- since the register-tile engine
    # (csrc/gemm_f32_rt.hpp) the forward beats the library GEMM at these shapes and no dropout pass over
    # [B,36,2048] exists any more (see profiles/README.md)
    fused = os.environ.get("VQA_FUSED_LINEAR", "1") == "1"
    # Set by a model whose ONLY consumer of this layer's relu output hands the gradient back already multiplied by
    # relu'(output) (MutanFusion(..., relu_input=True)): the fused backward then skips its own gate.
    grad_pregated = False

    def _fused_ok(self, x):
        return self.fused and self.af in (None, "relu") and x.dim() == 3 and x.is_cuda and x.dtype == torch.float32 \
            and self.out_channels >= 32 and x.size(0) * x.size(1) >= 1024

    def forward(self, x, predropped=False, packed=None):
        """predropped=True: the producer of x has already applied this layer's input dropout (ops.relation_apply).
        packed (bf16 inputs): this layer's shadows from the model's ops.ShadowPlan (else they are packed here)."""
        if predropped:
            if x.dim() != 3:
                raise ValueError("[error] putils.Conv1d(%s, %s, %s, %s): input_dim (%s) should equal to 3"
                                 % (self.in_channels, self.out_channels, self.kernel_size, self.stride, x.dim()))
            if x.dtype == torch.bfloat16:
                return self._linear_bf16(x, self.af, 0.0, packed)
            if self._fused_ok(x):
                return ops.linear_act(x, self.conv.weight.squeeze(-1), self.conv.bias, self.af, 0.0, 0,
                                      pregated=self.grad_pregated)
            return self._linear_f32(x)
        if x.dtype == torch.bfloat16 and self.af in (None, "relu"):
            if x.dim() != 3:
                raise ValueError("[error] putils.Conv1d(%s, %s, %s, %s): input_dim (%s) should equal to 3"
                                 % (self.in_channels, self.out_channels, self.kernel_size, self.stride, x.dim()))
            return self._linear_bf16(x, self.af, self.p if (self.training and self.p) else 0.0, packed)
        if self._fused_ok(x):
            # large region-side projection (compress_v / compress_v2): dropout + GEMM + bias + relu in ONE kernel
            # on the fp32 MFMA tile engine (K5); the dropout mask is a counter hash keyed by a seed drawn from
            # torch's CPU generator, so torch.manual_seed governs it and no mask tensor exists
            p = self.p if (self.training and self.p) else 0.0
            seed = ops.next_dropout_seed() if p else 0
            return ops.linear_act(x, self.conv.weight.squeeze(-1), self.conv.bias, self.af, p, seed,
                                  pregated=self.grad_pregated)
        if self.af == "relu" and x.dim() == 3 and x.dtype == torch.float32 and x.is_cuda and \
                not ops.attention_logits_supported(x, self.in_channels, self.out_channels):
            if self.p:
                x = F.dropout(x, p=self.p, training=self.training)
            return self._linear_f32(x)
        return _activation(self.pre_activation(x), self.af, self.dim)

    def _linear_f32(self, x):
        """af(conv) with a relu riding in the library GEMM's epilogue (ops.LinearFn); other activations as torch ops."""
        w, b = self.conv.weight.squeeze(-1), self.conv.bias
        if self.af == "relu" and x.is_cuda:
            return ops.linear(x, w, b, act="relu")
        return _activation(ops.linear(x, w, b), self.af, self.dim)


class MutanFusion(nn.Module):
    """putils/__init__.py:205-241: sum_r Linear1_r(x1) * Linear2_r(x2) (x2 [B,in2] broadcasts over the
    region axis of x1 [B,N,in1]).  Region side + product + rank sum = HIP kernel K4 (fp32 MFMA)."""

    def __init__(self, input_dim1, input_dim2, hidden_dim, R, seed=None):
        super().__init__()
        self.input_dim1 = input_dim1
        self.input_dim2 = input_dim2
        self.hidden_dim = hidden_dim
        self.R = R
        self.list_linear1 = nn.ModuleList([Linear(input_dim1, hidden_dim) for _ in range(R)])
        self.list_linear2 = nn.ModuleList([Linear(input_dim2, hidden_dim) for _ in range(R)])

    def stack_groups(self):
        return linear_stack_groups(list(self.list_linear2)) + linear_stack_groups(list(self.list_linear1))

    def forward(self, inputs1, inputs2, relu_input=False, packed=None, h2=None):
        """relu_input: inputs1 is the relu output of the layer in front and the gradient returned for it may come back
        multiplied by (inputs1 > 0) already (that layer's backward then skips the gate: MyConv1d.grad_pregated).
        packed (bf16 inputs1): the region-side shadows from the model's ops.ShadowPlan.
        h2 [B,R,H]: the question-side rank factors Linear2_r(inputs2) when a grouped phase of the model has computed them
        already (head.GatesAndRankFactors); inputs2 is then not read."""
        # (bf16 region tensors carry the feature dim zero-padded to a multiple of 64: see ops.pad_to)
        want = ops.pad_to(self.input_dim1) if inputs1.dtype == torch.bfloat16 else self.input_dim1
        if inputs1.size(-1) != want:
            raise ValueError(
                "[error] putils.Linear(%s, %s): last dimension of input(%s) should equal to in_features(%s)"
                % (self.input_dim1, self.hidden_dim, inputs1.size(-1), self.input_dim1))
        if h2 is None:
            if inputs2.dim() != 2 or inputs2.size(0) != inputs1.size(0):
                raise ValueError("MutanFusion: inputs2 must be [B, input_dim2] with the batch of inputs1")
            # question side: R small [B,in2]x[in2,H] GEMMs (Linear's own check raises ValueError on a bad last dim)
            if inputs2.size(-1) != self.input_dim2:
                raise ValueError(
                    "[error] putils.Linear(%s, %s): last dimension of input(%s) should equal to in_features(%s)"
                    % (self.input_dim2, self.hidden_dim, inputs2.size(-1), self.input_dim2))
            h2 = my_linears(list(self.list_linear2), inputs2)                          # [B,R,H], one batched GEMM
        if inputs1.dim() == 2 and inputs1.is_cuda:
            # vector-vector fusion (fusion_final: B rows, not B*N): a few hundred MFLOP, far too little for the K4 tile
            # kernels (16-64 workgroups, each walking the whole K = 1240 twice: 85 us forward).  Both sides run as
            # library GEMMs and the rank sum is one fused multiply-reduce kernel (ops.rank_product).
            h1 = my_linears(list(self.list_linear1), inputs1)                          # [B,R,H]
            return ops.rank_product(h1, h2)
        weights = [lin.linear.weight for lin in self.list_linear1]
        biases = [lin.linear.bias for lin in self.list_linear1]
        return ops.lowrank_bilinear_fusion(inputs1, h2, weights, biases, gate_dx=relu_input, packed=packed)


class MyATT(nn.Module):
    """config/CoR2.py:125-157: alpha = softmax over regions of conv_att(fuse); pooled = alpha^T @ inputs;
    one MyLinear per glimpse; concat.  Returns (x_v [B,att_dim], tuple of G tensors [B,N,1]) like the
    reference.  Softmax + pooling = HIP kernel K3."""

    def __init__(self, fuse_dim, glimpses, inputs_dim, att_dim, seed=None, af="tanh"):
        super().__init__()
        assert att_dim % glimpses == 0
        self.glimpses = glimpses
        self.inputs_dim = inputs_dim
        self.att_dim = att_dim
        self.conv_att = MyConv1d(fuse_dim, glimpses, 1, 1, seed=seed, p=0.5, af="softmax", dim=1)
        self.list_linear_v_fusion = nn.ModuleList(
            [MyLinear(inputs_dim, int(att_dim / glimpses), p=0.5, af=af) for _ in range(glimpses)])
        self.af = af

    def stack_groups(self):
        return linear_stack_groups(list(self.list_linear_v_fusion))

    def glimpse_dropout(self):
        """Input dropout rate of the per-glimpse MyLinear list in the current mode (0.0 in eval)."""
        first = self.list_linear_v_fusion[0]
        return float(first.p) if (self.training and getattr(first, "p", None)) else 0.0

    def _fused_dropout_ok(self):
        """The G glimpse layers share one dropout rate (else each draws its own mask at its own rate: the batched form and
        the fused store do not apply)."""
        mods = list(self.list_linear_v_fusion)
        return os.environ.get("VQA_FUSE_POOL_DROPOUT", "1") == "1" and all(getattr(m, "p", None) == mods[0].p for m in mods)

    def glimpse_projection(self, pooled, predropped=False, grouped=False):
        """cat_g MyLinear_g(pooled[:, g, :]) (config/CoR2.py:143-147).  The G layers have one shape, so they run as ONE
        batched GEMM over the [B,G,D] tensor (one dropout draw over all of it, one bias add, one activation) instead of
        G x {slice, dropout, GEMM, activation} and, backward, G slice gradients that autograd zero-fills and adds.
        grouped: the caller runs the grouped head; under VQA_HEAD=grouped the projections are a phase of it
        (head.GlimpseProjections) -- ONLY when the consumer of the result is the head's fusion phase, which hands the
        gradient back gated by the relu (gating it again, as the batched form does, changes nothing)."""
        from . import head
        if grouped and head.glimpses_grouped():
            mods = list(self.list_linear_v_fusion)
            pd = self.glimpse_dropout()
            if pd and not predropped:
                pooled = ops.dropout(pooled, pd)
            return head.GlimpseProjections.apply(pooled.contiguous(), *[m.linear.weight for m in mods],
                                                 *[m.linear.bias for m in mods])
        y = my_linears(list(self.list_linear_v_fusion), pooled, predropped=predropped)   # [B,G,A]
        return y.reshape(y.size(0), -1)

    def grouped_ok(self, inputs):
        """Can the glimpse projections run as a grouped phase?  (relu layers of one shape, fp32 GPU tensors)"""
        mods = list(self.list_linear_v_fusion)
        return inputs.is_cuda and self.af == "relu" and self.inputs_dim % 2 == 0 and \
            all(getattr(m, "af", None) == "relu" and m.out_features == mods[0].out_features and m.p == mods[0].p for m in mods)

    def attend(self, inputs, logits, pooled_map=None, return_pooled=False, grouped=False):
        """logits [B,N,G] (pre-softmax) -> (x_v, list_att, alpha [B,N,G][, pooled[:, 0] [B,D]]).  pooled_map (optional):
        ``pooled_map(pooled, p) -> tensor`` transforms the pooled features before the glimpse projections AND applies
        their input dropout at rate p (so the two share one pass).  (Nothing that carries an autograd graph is kept on
        the module: a tensor stashed across steps would pin the previous step's graph.)"""
        first = None
        pd = self.glimpse_dropout()
        if pooled_map is None and inputs.is_cuda and (pd > 0 or return_pooled) and self._fused_dropout_ok():
            # the glimpse projections' input dropout rides in the pooling kernel's store (one pass fewer each way), and
            # glimpse 0 comes back undropped as its own tensor for the caller (CoR2's relation step)
            res = ops.softmax_attention_pool_drop(logits, inputs, pd, ops.next_dropout_seed() if pd else 0, return_pooled)
            alpha, pooled = res[0], res[1]
            x_v = self.glimpse_projection(pooled, predropped=True, grouped=grouped)
            if return_pooled:
                return x_v, torch.split(alpha, 1, dim=2), alpha, res[2]
            return x_v, torch.split(alpha, 1, dim=2), alpha
        alpha, pooled = ops.softmax_attention_pool(logits, inputs)                     # [B,N,G], [B,G,D]
        if return_pooled:       # glimpse 0 for the caller (CoR2's relation step), the whole tensor for the projections
            pooled, first = ops.with_first_group(pooled)
        if pooled_map is None:
            x_v = self.glimpse_projection(pooled, grouped=grouped)
        else:
            x_v = self.glimpse_projection(pooled_map(pooled, self.glimpse_dropout()), predropped=True, grouped=grouped)
        if return_pooled:
            return x_v, torch.split(alpha, 1, dim=2), alpha, first
        return x_v, torch.split(alpha, 1, dim=2), alpha

    def forward(self, inputs, fuse):
        x_v, list_att, _ = self.attend(inputs, self.conv_att.pre_activation(fuse))
        return x_v, list_att


class SideOutputs(dict):
    """The ``alpha_dict`` side output of a model (read by visu.py:198-207 after a forward) with entries that may be computed
    on first access: a value stored as a zero-argument callable is replaced by its result when it is read.  Only a
    TRAINING forward stores such an entry (``alpha_dict['feature']``: a training step never reads it, so the kernel that
    would materialise it does not run in the step; read after a graph-replayed step it reflects the latest replay); an
    eval forward -- the visualisation path -- stores plain tensors like the reference.  Every way of getting the values
    out (indexing, get, items, values, iteration-based copies, dict(), ``**``, copy, pickling) resolves them first."""

    def _resolve(self, key):
        value = dict.__getitem__(self, key)
        if callable(value) and not isinstance(value, torch.Tensor):
            value = value()
            dict.__setitem__(self, key, value)
        return value

    def _resolve_all(self):
        for k in list(dict.keys(self)):
            self._resolve(k)

    def __getitem__(self, key):
        return self._resolve(key)

    def get(self, key, default=None):
        return self._resolve(key) if key in self else default

    def items(self):
        self._resolve_all()
        return dict.items(self)

    def values(self):
        self._resolve_all()
        return dict.values(self)

    def keys(self):          # dict(d) / {**d} walk keys() and then index: resolve up front so C-level fast paths see tensors
        self._resolve_all()
        return dict.keys(self)

    def __iter__(self):
        self._resolve_all()
        return dict.__iter__(self)

    def copy(self):
        self._resolve_all()
        return SideOutputs(dict.copy(self))

    def __reduce__(self):
        self._resolve_all()
        return (SideOutputs, (dict(dict.items(self)),))


class QuestionVectorInput(nn.Module):
    """Stand-in for the question encoder slot ``seq2vec`` (putils.SkipThoughts, putils/__init__.py:878-985,
    is upstream of the hot path and needs weight files that are not available offline): the sample's
    'q_idxes' entry already holds the 2400-d question vector and is passed through."""

    def __init__(self, dim=2400):
        super().__init__()
        self.dim = dim

    def forward(self, q):
        if not torch.is_floating_point(q) or q.size(-1) != self.dim:
            raise ValueError("seq2vec slot holds no encoder: pass the %d-d question vector as sample['q_idxes'] "
                             "(or construct the Model with seq2vec=<your encoder>)" % self.dim)
        return q


def question_feature(seq2vec, q):
    """What ``Model.forward`` feeds the head: ``seq2vec(q_idxes)`` for int64 token ids [B,T] as the reference does
    (config/CoR2.py:205, fed by datasets.py:928-969), and -- north_star's "(region-feature, question-embedding)"
    interface -- a floating [B,2400] tensor taken as the question vector itself, whatever the slot holds."""
    if torch.is_floating_point(q) and q.dim() == 2 and q.size(-1) == 2400:
        return q
    return seq2vec(q)
