"""bf16-aware restatement of the CoR2 head for BASELINE configs[4] ("CoR2 bf16, 100x2048 dense regions").
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- never imported by the product package.

The reference (config/CoR2.py:160-237) is fp32-only, so for the mixed-precision configuration the checker is its
restatement (oracle/reference_faithful.py, pinned against the reference's own outputs) with every tensor the product
STORES in bf16 rounded to bf16 at the same point, forward and backward -- everything else in float64:

  forward   v; the packed shadows of compress_v / compress_v2 / fusion_vq{1,2}.list_linear1 weights; the outputs of
            compress_v, compress_v2 (after relu), fusion_vq1, fusion_vq2; the relation tensor v2 handed to compress_v2
  backward  the gradients of those same activations (d fuse from the attention-logit kernel, d x from K4's data-gradient
            GEMM, d v2 from compress_v2's); K4 in its R-GEMM form (the product's default): the rank-scaled gradient g * h2_r (its
            GEMM operand) and the saved h1 that d h2 is contracted against; in its rank-folded form (k4_form="fold", round 5): the
            per-sample folded weight sum_r diag(h2_r) W1_r, the bf16 operand of its forward and data-gradient products

With ``rounding=False`` the class computes the reference's function exactly (closed form of the relation step, see
below) and tests/test_oracle_golden.py checks it against reference_faithful.CoR2Oracle -- that anchors this file to the
pinned oracle; with rounding on, a relu gate or a value can differ from the product's only where fp32 and float64
accumulation straddle a rounding boundary, so the GPU comparison runs at 2e-2 of each tensor's scale instead of the
cosine / 15 % Frobenius bar a plain fp32 oracle needs (tests/test_gpu_bf16.py).

Relation step: the reference builds out[b,i,j,:] = v_i q1 + v_j q2 and reduces it with alpha1[:, :, 0] (config/CoR2.py:
191-199, :216).  That equals q1 * sum_i alpha_i v_i + (sum_i alpha_i) q2 * v_j; a softmax alpha sums to 1, and the product
takes sum_i alpha_i v_i from the first attention's pooled glimpse 0 -- this class does the same (B = 128, N = 100 would
need 82 MB per sample for the literal tensor).
"""
import torch
import torch.nn.functional as F

from . import reference_faithful as RF


def _bf(x):
    return x.float().to(torch.bfloat16).to(x.dtype)


class _Round(torch.autograd.Function):
    """y = bf16(x) forward; the gradient is rounded too when ``both`` (the tensor's gradient is stored in bf16 as well)."""

    @staticmethod
    def forward(ctx, x, both):
        ctx.both = both
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return (_bf(g) if ctx.both else g), None


class _RankProduct(torch.autograd.Function):
    """out[b,n,:] = sum_r h1[b,n,r,:] * h2[b,r,:]  (putils/__init__.py:232-238) with the product's backward roundings:
    d h1_r = bf16(g * h2_r) (the operand of K4's data / weight gradient GEMMs), d h2 = sum_n g * bf16(h1) (h1 is saved in
    bf16)."""

    @staticmethod
    def forward(ctx, h1, h2, rounding):
        ctx.save_for_backward(h1, h2)
        ctx.rounding = rounding
        return (h1 * h2.unsqueeze(1)).sum(2)

    @staticmethod
    def backward(ctx, g):
        h1, h2 = ctx.saved_tensors
        gs = g.unsqueeze(2) * h2.unsqueeze(1)
        h1s = h1
        if ctx.rounding:
            gs, h1s = _bf(gs), _bf(h1)
        return gs, (g.unsqueeze(2) * h1s).sum(1), None


class CoR2MixedOracle(RF.CoR2Oracle):
    """reference_faithful.CoR2Oracle's parameters (same names: one seeded state_dict loads into both) with the forward
    described in the module docstring.  Eval mode needs nothing else.  Training mode (dropout at every Drop* layer) is a
    comparison with the product's exported masks: the caller switches the DropLinear modules' own F.dropout off and masks their
    inputs by forward pre-hooks (as tests/test_gpu_models.py does for the float64 restatement), and hands the masks of the four
    1x1-convolution sites -- which this forward evaluates without calling the modules -- in sample["site_masks"] =
    {"compress_v", "compress_v2": [b,N,2048]; "att1.conv_att", "att2.conv_att": [b,N,510]} (keep / (1 - p) values: 0 or 2).
    sample["forced_gates"] (eval mode, optional) = {"compress_v", "compress_v2": bool [b,N,310]; "att1.glimpses",
    "att2.glimpses": bool [b,620]}: the product's relu decisions at the sites a bf16 rounding can flip (see _gate)."""

    def __init__(self, *args, rounding=True, k4_form="rgemm", **kw):
        super().__init__(*args, **kw)
        self.rounding = rounding
        self.k4_form = k4_form          # which of the product's two K4 forms is restated ("rgemm" = its default, "fold")

    def _r(self, x, both=False):
        return _Round.apply(x, both) if self.rounding else x

    def _gate(self, site, pre, forced):
        """relu(pre) -- or, with the product's own gates handed in (sample["forced_gates"][site], a bool tensor: "the
        product's output of this relu is > 0"), pre * gate.  The units where the two sides decide differently are counted in
        self.gate_flips[site] = (units that differ, units, largest |pre| among them / rms(pre)): a test that forces the gates
        asserts with these numbers that only knife-edge units (|pre| within rounding of 0) ever differ."""
        if forced is None:
            return F.relu(pre)
        own = pre.detach() > 0
        diff = own != forced
        rms = float(pre.detach().pow(2).mean().sqrt())
        n, edge = self.gate_flips.get(site, (0, 0, 0.0))[0], self.gate_flips.get(site, (0, 0, 0.0))[2]
        units = self.gate_flips.get(site, (0, 0, 0.0))[1]
        worst = float(pre.detach()[diff].abs().max()) / max(rms, 1e-30) if bool(diff.any()) else 0.0
        self.gate_flips[site] = (n + int(diff.sum()), units + diff.numel(), max(edge, worst))
        return pre * forced.to(pre.dtype)

    def _region_linear(self, mod, x, mask=None, site=None, forced=None):
        """relu(drop(x) W^T + b) with the bf16 shadow of W; the output and its gradient are bf16.  (The product zeroes the
        dropped elements of the bf16 operand and applies the factor 2 to the accumulator: the same numbers, 2 is exact.)"""
        w = self._r(mod.conv.weight.squeeze(-1))
        if mask is not None:
            x = x * mask
        return self._r(self._gate(site, F.linear(x, w, mod.conv.bias), forced), both=True)

    def _fusion(self, mf, x_low, q_low):
        h2 = torch.stack([lin(q_low) for lin in mf.list_linear2], 1)                                     # [B,R,H] fp32
        if self.k4_form == "rgemm":      # the product's R-GEMM form (csrc/bf16_path.hip, its default)
            h1 = torch.stack([F.linear(x_low, self._r(lin.linear.weight), lin.linear.bias) for lin in mf.list_linear1], 2)
            return self._r(_RankProduct.apply(h1, h2, self.rounding), both=True)                          # [B,N,H] bf16
        # the product's rank-folded form (csrc/bilinear_fold_bf16.hip, VQA_K4_BF16_FORM=fold): the same sum reassociated --
        #   sum_r (x W1_r^T + b1_r) h2_r  =  x Wb^T + sum_r h2_r b1_r,   Wb = sum_r diag(h2_r[b]) W1_r  per sample --
        # with Wb folded from the bf16 shadows of W1_r in fp32 and ROUNDED to bf16 (it is the MFMA operand of the forward and of
        # the data gradient; the rounding is a straight-through step for the gradients of W1_r and h2, which the product forms
        # from P_b = g_b^T x_b in fp32).  With rounding off this is the reference's value exactly.
        w1 = torch.stack([self._r(lin.linear.weight) for lin in mf.list_linear1], 0)                      # [R,H,L]
        b1 = torch.stack([lin.linear.bias for lin in mf.list_linear1], 0)                                 # [R,H]
        wb = self._r(torch.einsum("brh,rhl->bhl", h2, w1))                                                # [B,H,L] bf16
        out = torch.einsum("bnl,bhl->bnh", x_low, wb) + torch.einsum("brh,rh->bh", h2, b1).unsqueeze(1)
        return self._r(out, both=True)                                                                     # [B,N,H] bf16

    @staticmethod
    def _attend(att, fuse, v, mask=None):
        if mask is not None:
            fuse = fuse * mask
        logits = att.conv_att.conv(fuse.transpose(1, 2)).transpose(1, 2)                                  # [B,N,G]
        alpha = F.softmax(logits, dim=1)
        return alpha, torch.matmul(alpha.transpose(1, 2), v)                                              # pooled [B,G,D]

    def _glimpses(self, att, pooled, site=None, forced=None):
        if forced is None:
            return torch.cat([att.list_linear_v_fusion[g](pooled[:, g, :]) for g in range(att.glimpses)], dim=1)
        if self.training:
            raise NotImplementedError("forced gates are an eval-mode comparison (the glimpse layers' dropout is not restated here)")
        pre = torch.cat([F.linear(pooled[:, g, :], att.list_linear_v_fusion[g].linear.weight, att.list_linear_v_fusion[g].linear.bias)
                         for g in range(att.glimpses)], dim=1)
        return self._gate(site, pre, forced)

    def forward(self, sample):
        sm = sample.get("site_masks")
        if self.training and sm is None:
            raise NotImplementedError("CoR2MixedOracle in training mode needs the product's masks (see the class docstring)")
        sm = sm or {}
        fg = sample.get("forced_gates") or {}      # the product's relu gates (bool, "output > 0"): see _gate
        if not hasattr(self, "gate_flips") or sample.get("reset_gate_flips", True):
            self.gate_flips = {}
        v = sample["v"]
        q = sample["q"] if "q" in sample else sample["q_idxes"]
        b, n = v.size(0), v.size(1)
        v = self._r(v.contiguous().view(b, n, -1))
        q_low = self.compress_q(q)
        q1 = self.expand_q_1(self.compress_q_1(q))
        q2 = self.expand_q_2(self.compress_q_2(q))
        v_low = self._region_linear(self.compress_v, v, sm.get("compress_v"), "compress_v", fg.get("compress_v"))
        fuse1 = self._fusion(self.fusion_vq1, v_low, q_low)
        alpha1, pooled1 = self._attend(self.att1, fuse1, v, sm.get("att1.conv_att"))
        v1_att = self._glimpses(self.att1, pooled1, "att1.glimpses", fg.get("att1.glimpses"))
        t = q1 * pooled1[:, 0, :]
        v2 = self._r(t.unsqueeze(1) + q2.unsqueeze(1) * v, both=True)                                     # relation tensor
        v2_low = self._region_linear(self.compress_v2, v2, sm.get("compress_v2"), "compress_v2", fg.get("compress_v2"))
        fuse2 = self._fusion(self.fusion_vq2, v2_low, q_low)
        alpha2, pooled2_v = self._attend(self.att2, fuse2, v, sm.get("att2.conv_att"))
        pooled2 = t.unsqueeze(1) + q2.unsqueeze(1) * pooled2_v               # = alpha2^T v2 for a softmax alpha2
        v2_att = self._glimpses(self.att2, pooled2, "att2.glimpses", fg.get("att2.glimpses"))
        self.alpha_dict = {"alpha1": torch.split(alpha1, 1, dim=2), "alpha2": torch.split(alpha2, 1, dim=2),
                           "feature": v2[:, [0, 1], :]}
        self.taps = {"v2_feature": v2, "fusion_vq1": fuse1, "fusion_vq2": fuse2, "compress_v": v_low, "compress_v2": v2_low}
        x = self.fusion_final(torch.cat([v1_att, v2_att], dim=1), self.linear_q(q))
        return self.linear_classif(x)
