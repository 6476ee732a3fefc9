"""Torch-CPU restatement of the reference's CoR2 / ODA hot path.  TEST INFRASTRUCTURE
ONLY (see oracle/__init__.py) -- the checker for the HIP path and the ``cpu_baseline``
("port") of bench.py.  Never imported by the product package.

It follows the reference's *op sequence*, including the things that make the
reference slow on purpose, so that timing it is a fair stand-in for timing the
reference on the GPU box's host cores (the reference itself cannot travel):

  * per-sample python loops + ``torch.stack`` for broadcast multiply and batched
    matmul                                   (putils/__init__.py:89-104)
  * the materialised [B,N,N,D] pairwise tensor (config/CoR2.py:191-199, :216)
  * the N*N-iteration object-difference loop    (config/ODA.py:216-222)

Differences, all deliberate: the region count is read from the input instead of
being the literal 36; per-glimpse squeezes keep the batch axis, so B=1 works
(the reference raises IndexError there); the question
encoder is outside (the 2400-d vector is an input, key 'q' or 'q_idxes').
Parameter names and shapes equal the reference's so one
seeded state_dict loads into the reference, this oracle and the product model.

Pinned by tests/test_oracle_golden.py against tests/golden/*.npz (outputs of the
imported reference).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def per_sample_mul(x, y):
    """putils/__init__.py:98-104 (bmul): out[b] = x[b] * y[b], python loop + stack."""
    return torch.stack([x[b] * y[b] for b in range(x.size(0))], dim=0)


def per_sample_matmul(a, b):
    """putils/__init__.py:89-95 (bmatmul): out[i] = a[i] @ b[i], python loop + stack."""
    return torch.stack([torch.matmul(a[i], b[i]) for i in range(a.size(0))], dim=0)


def _activate(x, af, dim):
    if not af:
        return x
    if af == "softmax":
        return F.softmax(x, dim=dim)
    return getattr(torch, af)(x) if af in ("sigmoid", "tanh") else getattr(F, af)(x)


class CheckedLinear(nn.Module):
    """putils/__init__.py:16-33 (Linear): nn.Linear under the name ``linear`` + last-dim check."""

    def __init__(self, fin, fout):
        super().__init__()
        self.fin, self.fout = fin, fout
        self.linear = nn.Linear(fin, fout, bias=True)

    def forward(self, x):
        if x.size(-1) != self.fin:
            raise ValueError("Linear(%d,%d): last dim of input is %d" % (self.fin, self.fout, x.size(-1)))
        return self.linear(x)


class DropLinear(nn.Module):
    """config/CoR2.py:94-122 (MyLinear): af(linear(dropout_p(x)))."""

    def __init__(self, fin, fout, p=None, af=None, dim=None):
        super().__init__()
        self.fin, self.fout, self.p, self.af, self.dim = fin, fout, p, af, dim
        self.linear = nn.Linear(fin, fout, bias=True)

    def forward(self, x):
        if x.size(-1) != self.fin:
            raise ValueError("MyLinear(%d,%d): last dim of input is %d" % (self.fin, self.fout, x.size(-1)))
        if self.p:
            x = F.dropout(x, p=self.p, training=self.training)
        return _activate(self.linear(x), self.af, self.dim)


class DropConv1x1(nn.Module):
    """config/CoR2.py:56-91 (MyConv1d, kernel 1): af(conv1d(dropout_p(x)^T)^T) on [B,N,Cin]."""

    def __init__(self, cin, cout, p=None, af=None, dim=None):
        super().__init__()
        self.cin, self.cout, self.p, self.af, self.dim = cin, cout, p, af, dim
        self.conv = nn.Conv1d(cin, cout, 1, 1, padding=0, bias=True)

    def forward(self, x):
        if x.dim() != 3:
            raise ValueError("MyConv1d(%d,%d): input must be 3-D, got %d-D" % (self.cin, self.cout, x.dim()))
        if self.p:
            x = F.dropout(x, p=self.p, training=self.training)
        x = self.conv(x.transpose(1, 2)).transpose(1, 2)
        return _activate(x, self.af, self.dim)


class LowRankBilinear(nn.Module):
    """putils/__init__.py:205-241 (MutanFusion): sum_r Linear1_r(x1) * Linear2_r(x2), the product
    taken sample by sample so a [B,N,H] left factor broadcasts against a [B,H] right factor."""

    def __init__(self, d1, d2, hidden, rank):
        super().__init__()
        self.rank = rank
        self.list_linear1 = nn.ModuleList([CheckedLinear(d1, hidden) for _ in range(rank)])
        self.list_linear2 = nn.ModuleList([CheckedLinear(d2, hidden) for _ in range(rank)])

    def forward(self, x1, x2):
        total = 0
        for r in range(self.rank):
            total = total + per_sample_mul(self.list_linear1[r](x1), self.list_linear2[r](x2))
        return total


class GlimpseAttention(nn.Module):
    """config/CoR2.py:125-157 (MyATT): alpha = softmax over regions of a 1x1 conv of ``fuse``;
    pooled = alpha^T @ inputs per sample; one DropLinear per glimpse; concatenate."""

    def __init__(self, fuse_dim, glimpses, inputs_dim, att_dim, af="tanh"):
        super().__init__()
        assert att_dim % glimpses == 0
        self.glimpses = glimpses
        self.conv_att = DropConv1x1(fuse_dim, glimpses, p=0.5, af="softmax", dim=1)
        self.list_linear_v_fusion = nn.ModuleList(
            [DropLinear(inputs_dim, att_dim // glimpses, p=0.5, af=af) for _ in range(glimpses)])

    def forward(self, inputs, fuse):
        alpha = self.conv_att(fuse)                                   # [B,N,G]
        pooled = per_sample_matmul(alpha.transpose(1, 2), inputs)     # [B,G,D]
        parts = [self.list_linear_v_fusion[g](pooled[:, g, :]) for g in range(self.glimpses)]
        return torch.cat(parts, dim=1), torch.split(alpha, 1, dim=2)


class CoR2Oracle(nn.Module):
    """config/CoR2.py:160-237 (Model) minus seq2vec."""

    def __init__(self, num_ans=2000, feat=2048, qdim=2400, low=310, hidden=510, glimpses=4, att_dim=620):
        super().__init__()
        self.compress_v = DropConv1x1(feat, low, p=0.5, af="relu")
        self.compress_v2 = DropConv1x1(feat, low, p=0.5, af="relu")
        self.compress_q = DropLinear(qdim, low, p=0.5, af="relu")
        self.fusion_vq1 = LowRankBilinear(low, low, hidden, 2)
        self.att1 = GlimpseAttention(hidden, glimpses, feat, att_dim, af="relu")
        self.fusion_vq2 = LowRankBilinear(low, low, hidden, 2)
        self.att2 = GlimpseAttention(hidden, glimpses, feat, att_dim, af="relu")
        self.linear_q = DropLinear(qdim, low, p=0.5, af="relu")
        self.fusion_final = LowRankBilinear(2 * att_dim, low, hidden, 2)
        self.linear_classif = DropLinear(hidden, num_ans, p=0.5)
        self.compress_q_1 = DropLinear(qdim, low, p=0.5, af="relu")
        self.expand_q_1 = DropLinear(low, feat, p=0.5, af="sigmoid")
        self.compress_q_2 = DropLinear(qdim, low, p=0.5, af="relu")
        self.expand_q_2 = DropLinear(low, feat, p=0.5, af="sigmoid")
        self.alpha_dict = {}
        self.taps = {}

    def pairwise_tensor(self, v, q):
        """config/CoR2.py:191-199 (decare_cat): out[b,i,j,:] = v[b,i,:]*q1[b,:] + v[b,j,:]*q2[b,:],
        built the reference's way: two repeat() materialisations, two per-sample multiplies, one add."""
        b, n, d = v.size()
        left = v.view(b, n, 1, d).repeat(1, 1, n, 1)
        right = v.view(b, 1, n, d).repeat(1, n, 1, 1)
        q1 = self.expand_q_1(self.compress_q_1(q))
        q2 = self.expand_q_2(self.compress_q_2(q))
        return per_sample_mul(left, q1) + per_sample_mul(right, q2)

    def forward(self, sample):
        v = sample["v"]
        q = sample["q"] if "q" in sample else sample["q_idxes"]
        b, n = v.size(0), v.size(1)
        v = v.contiguous().view(b, n, -1)
        q_low = self.compress_q(q)
        v_low = self.compress_v(v)
        fuse1 = self.fusion_vq1(v_low, q_low)
        v1_att, alpha1 = self.att1(v, fuse1)
        cat = self.pairwise_tensor(v, q)                                        # [B,N,N,D]
        v2 = (alpha1[0].contiguous().view(b, n, 1, 1) * cat).sum(1)             # glimpse 0 only (CoR2.py:216)
        v2_low = self.compress_v2(v2)
        fuse2 = self.fusion_vq2(v2_low, q_low)
        v2_att, alpha2 = self.att2(v2, fuse2)
        self.alpha_dict = {"alpha1": alpha1, "alpha2": alpha2, "feature": v2[:, [0, 1], :]}
        self.taps = {"v2_feature": v2, "fusion_vq1": fuse1, "fusion_vq2": fuse2, "compress_v": v_low,
                     "att1.x_v": v1_att, "att2.x_v": v2_att}
        x = self.fusion_final(torch.cat([v1_att, v2_att], dim=1), self.linear_q(q))
        return self.linear_classif(x)


class ODAOracle(nn.Module):
    """config/ODA.py:177-240 (Model) minus seq2vec."""

    def __init__(self, num_ans=3000, regions=36, feat=2048, qdim=2400, low=310, hidden=510, glimpses=4,
                 att_dim=620):
        super().__init__()
        self.compress_v = DropConv1x1(feat, low, p=0.5, af="relu")
        self.compress_q = DropLinear(qdim, low, p=0.5, af="relu")
        self.att = GlimpseAttention(regions * low, glimpses, feat, att_dim, af="relu")
        self.linear_q = DropLinear(qdim, low, p=0.5, af="relu")
        self.fusion_final = LowRankBilinear(att_dim, low, hidden, 5)
        self.linear_classif = DropLinear(hidden, num_ans, p=0.5)
        self.alpha_dict = {}
        self.taps = {}

    @staticmethod
    def difference_tensor(v_low, q_low):
        """config/ODA.py:216-222: vq[b,i,j*L+d] = (v_low[b,i,d]-v_low[b,j,d])*q_low[b,d], built by the
        reference's N*N python loop over region slices, stack, transpose, view."""
        b, n, _ = v_low.size()
        rows = [v_low[:, i, :] for i in range(n)]
        parts = []
        for vi in rows:
            for vj in rows:
                parts.append((vi - vj) * q_low)
        return torch.stack(parts, dim=0).transpose(0, 1).contiguous().view(b, n, -1)

    def forward(self, sample):
        v = sample["v"]
        q = sample["q"] if "q" in sample else sample["q_idxes"]
        b, n = v.size(0), v.size(1)
        v = v.contiguous().view(b, n, -1)
        v_low = self.compress_v(v)
        q_low = self.compress_q(q)
        vq = self.difference_tensor(v_low, q_low)
        v_final, alphas = self.att(v, vq)
        self.alpha_dict = {"alphas": alphas[0]}
        self.taps = {"compress_v": v_low, "compress_q": q_low, "att.x_v": v_final}
        x = self.fusion_final(v_final, self.linear_q(q))
        return self.linear_classif(x)


def kld_sum_loss(logits, target):
    """train.py:536-544: KLDivLoss(size_average=False)(log_softmax(logits, dim=1), target) -- a SUM
    over batch and classes (soft targets, datasets.py:963-969)."""
    return F.kl_div(F.log_softmax(logits, dim=1), target, reduction="sum")


def train_steps(model, batches, lr=1e-4, clip=0.25, gamma=0.5 ** (1 / 50000)):
    """train.py:41-107 + :286-299 step order: forward, loss, scheduler.step(), zero_grad, backward,
    clip_grad_norm_(0.25), Adam step.  Returns (losses, pre-clip grad norms, final lr)."""
    import warnings

    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=lr)
    sch = torch.optim.lr_scheduler.ExponentialLR(opt, gamma)
    losses, norms = [], []
    for v, q, a in batches:
        loss = kld_sum_loss(model({"v": v, "q": q}), a)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sch.step()
        opt.zero_grad()
        loss.backward()
        norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), clip)))
        opt.step()
        losses.append(loss.item())
    return losses, norms, opt.param_groups[0]["lr"]
