"""ODA ("Object-Difference Attention") head on the MI355X kernels -- drop-in for config/ODA.py:177-240.

Same constructor / forward / ``alpha_dict`` / state_dict names as the reference ``Model``
(``att.conv_att.conv.weight`` keeps its (4, N*310, 1) shape).  The [B,N,N*310] difference tensor and
its dropout mask are never built: HIP kernel K2 contracts them against the attention filter on the fly.
"""
import os

import torch
import torch.nn as nn

from . import head, ops
from .layers import MutanFusion, MyATT, MyConv1d, MyLinear, QuestionVectorInput, linear_stack_groups, my_linears, question_feature


class Model(nn.Module):
    def __init__(self, vocab_words=None, num_ans=None, seq2vec=None, regions=36):
        super().__init__()
        self.vocab_words = vocab_words
        self.num_classes = num_ans
        self.regions = regions

        if seq2vec == "skipthoughts":      # the reference's encoder (config/CoR2.py:166), randomly initialised here
            from .encoder import SkipThoughts
            seq2vec = SkipThoughts(vocab_list=vocab_words, gru="BayesianGRU", return_last=True, af="relu")
        if seq2vec == "vector":             # explicit pass-through slot: sample['q_idxes'] holds the 2400-d question vector
            seq2vec = None
        self.seq2vec = seq2vec if seq2vec is not None else QuestionVectorInput(2400)
        self.compress_v = MyConv1d(2048, 310, 1, 1, p=0.5, af="relu")
        self.compress_q = MyLinear(2400, 310, p=0.5, af="relu")
        self.att = MyATT(fuse_dim=regions * 310, glimpses=4, inputs_dim=2048, att_dim=620, af="relu")
        self.linear_q = MyLinear(2400, 310, p=0.5, af="relu")
        self.fusion_final = MutanFusion(620, 310, 510, 5)
        self.linear_classif = MyLinear(510, self.num_classes, p=0.5)
        self.alpha_dict = {}
        self._mask_step = 0
        # compress_v's relu output feeds K2 and nothing else: K2's backward holds it in registers (for d q) and hands the
        # gradient back already multiplied by (output > 0), so the projection's weight-gradient kernel runs ungated
        # (VQA_ODA_GATE_IN_K2=0: the gate in the projection's own backward, as in round 2)
        self.compress_v.grad_pregated = os.environ.get("VQA_ODA_GATE_IN_K2", "1") == "1"

    def stack_groups(self):
        return linear_stack_groups([self.compress_q, self.linear_q])

    def difference_logits(self, v_feature_low, q_feature_low):
        """config/ODA.py:216-222 + the dropout/1x1-conv of conv_att (config/ODA.py:149), fused (K2)."""
        conv = self.att.conv_att
        p = conv.p if (self.training and conv.p) else 0.0
        seed = 0
        if p:
            # fresh mask every call, drawn from torch's generator so torch.manual_seed governs it
            seed = ops.next_dropout_seed()
        w = conv.conv.weight.view(conv.out_channels, -1)
        return ops.object_difference_attention(v_feature_low, q_feature_low, w, conv.conv.bias, p, seed,
                                               gate_dvl=self.compress_v.grad_pregated)

    def _grouped_head_ok(self, q_feature, cut):
        """As cor2.Model._grouped_head_ok: the [B,.]-sized layers as grouped phases (head.py)."""
        return (cut is None and q_feature.is_cuda and q_feature.dtype == torch.float32 and not q_feature.requires_grad
                and self.compress_q.af == "relu" and self.linear_q.af == "relu" and self.compress_q.p == self.linear_q.p
                and self.linear_classif.af in (None, "") and self.att.grouped_ok(q_feature)
                and head.supported("oda", q_feature.size(1), self.compress_q.out_features, self.fusion_final.hidden_dim,
                                   self.fusion_final.input_dim1, self.num_classes))

    def late_parameters(self):
        """Parameters behind the attention (fusion_final, the classifier: 3.9 M of the 7.3 M): their gradients are complete
        before backward enters the attention and the region projection (see cor2.CoR2Model.late_parameters)."""
        return [p for m in (self.fusion_final, self.linear_classif) for p in m.parameters()]

    def forward_with_cut(self, sample):
        """-> (logits, outs, ins) as cor2.CoR2Model.forward_with_cut: the cut is (attended features, q_final)."""
        cut = []
        logits = self(sample, cut)
        return logits, cut[0], cut[1]

    def forward(self, sample, _cut=None):
        v = sample["v"]
        b = v.size(0)
        v_feature = v.contiguous().view(b, -1, 2048)
        if v_feature.dtype == torch.bfloat16 and v_feature.is_cuda:      # the feed's bf16 transport format, widened exactly (ops.widen_bf16)
            v_feature = ops.widen_bf16(v_feature)
        if v_feature.size(1) != self.regions:
            raise ValueError("ODA.Model was built for %d regions, input has %d" % (self.regions, v_feature.size(1)))
        q_feature = question_feature(self.seq2vec, sample["q_idxes"] if "q_idxes" in sample else sample["q"])

        v_feature_low = self.compress_v(v_feature)
        grouped = self._grouped_head_ok(q_feature, _cut)
        if grouped:
            # the [B,.]-sized layers as grouped phases (head.py): the two question projections (the object-difference kernel
            # reads the first one: its gradient comes back ungated), then fusion_final's question-side rank factors
            proj = [self.compress_q, self.linear_q]
            p_in = float(proj[0].p) if (self.training and proj[0].p) else 0.0
            q_feature_low, q_final = head.QuestionProjections.apply(
                q_feature.contiguous(), p_in, ops.next_dropout_seed() if p_in else 0, (), 0.0, 0, (0,),
                *[m.linear.weight for m in proj], *[m.linear.bias for m in proj])
            lin2 = list(self.fusion_final.list_linear2)
            (h2_final,) = head.GatesAndRankFactors.apply(2, (), ((1, self.fusion_final.R),), (1.0, 1.0), q_feature_low, q_final,
                                                         *[l.linear.weight for l in lin2], *[l.linear.bias for l in lin2])
        else:
            # the two MyLinear(2400 -> 310) on the question vector (config/ODA.py:185,193 applied at :207,:233): one batched GEMM
            q_both = my_linears([self.compress_q, self.linear_q], q_feature, group_first=True)   # [2,B,310]
            q_feature_low, q_final = ops.split_groups(q_both, (1, 1))                              # views; one cat kernel backward
        logits = self.difference_logits(v_feature_low, q_feature_low)
        v_final, alphas, _ = self.att.attend(v_feature, logits, grouped=grouped)

        self.alpha_dict = {"alphas": alphas[0].detach()}
        if grouped:
            p_c = float(self.linear_classif.p) if (self.training and self.linear_classif.p) else 0.0
            seed = ops.next_dropout_seed() if p_c else 0
            lin1 = list(self.fusion_final.list_linear1)
            x = head.VectorFusion.apply(h2_final, p_c, seed, 1, v_final, *[l.linear.weight for l in lin1],
                                        *[l.linear.bias for l in lin1])
            return head.Classifier.apply(x, self.linear_classif.linear.weight, self.linear_classif.linear.bias, p_c, seed)

        if _cut is not None:
            outs = [v_final, q_final]
            ins = [t.detach().requires_grad_(t.requires_grad) for t in outs]
            _cut.extend([outs, ins])
            v_final, q_final = ins
        x = self.fusion_final(v_final, q_final)
        return self.linear_classif(x)
