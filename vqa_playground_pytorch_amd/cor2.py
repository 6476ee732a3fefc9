"""CoR2 ("Chain of Reasoning") head on the MI355X kernels -- drop-in for config/CoR2.py:160-237.

Same constructor, ``forward(sample) -> logits [B,num_ans]``, ``alpha_dict`` side output and
state_dict names as the reference ``Model``; the region count is read from the input instead of
being the literal 36, and per-replica batch 1 works (the reference raises IndexError there).
"""
import os

import torch
import torch.nn as nn

from . import head, ops
from .layers import MutanFusion, MyATT, MyConv1d, MyLinear, QuestionVectorInput, SideOutputs, linear_stack_groups, my_linears, \
    question_feature


class Model(nn.Module):
    def __init__(self, vocab_words=None, num_ans=None, seq2vec=None, relation_mode=1, compute_dtype=None):
        super().__init__()
        self.vocab_words = vocab_words
        self.num_classes = num_ans
        # None / torch.float32: the reference's fp32 everywhere.  torch.bfloat16 (BASELINE configs[4]): the region-side
        # tensors (v, v2, compressed regions, fusion outputs) are stored in bf16 and contracted on the bf16 MFMA engine
        # with fp32 accumulation; parameters stay fp32 masters, the question side and everything [B,*]-sized stays fp32.
        if compute_dtype in ("bf16", "bfloat16"):
            compute_dtype = torch.bfloat16
        if compute_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("compute_dtype must be None, torch.float32 or torch.bfloat16, got %r" % (compute_dtype,))
        self.compute_dtype = compute_dtype or torch.float32
        # 0 = pairwise (every (i,j) term summed from the LDS tile), 1 = factored (same value, one pass)
        self.relation_mode = relation_mode

        if seq2vec == "skipthoughts":      # the reference's encoder (config/CoR2.py:166), randomly initialised here
            from .encoder import SkipThoughts
            seq2vec = SkipThoughts(vocab_list=vocab_words, gru="BayesianGRU", return_last=True, af="relu")
        if seq2vec == "vector":             # explicit pass-through slot: sample['q_idxes'] holds the 2400-d question vector
            seq2vec = None
        self.seq2vec = seq2vec if seq2vec is not None else QuestionVectorInput(2400)
        self.compress_v = MyConv1d(2048, 310, 1, 1, p=0.5, af="relu")
        self.compress_v2 = MyConv1d(2048, 310, 1, 1, p=0.5, af="relu")
        self.compress_q = MyLinear(2400, 310, p=0.5, af="relu")

        self.fusion_vq1 = MutanFusion(310, 310, 510, 2)
        self.att1 = MyATT(fuse_dim=510, glimpses=4, inputs_dim=2048, att_dim=620, af="relu")

        self.fusion_vq2 = MutanFusion(310, 310, 510, 2)
        self.att2 = MyATT(fuse_dim=510, glimpses=4, inputs_dim=2048, att_dim=620, af="relu")

        self.linear_q = MyLinear(2400, 310, p=0.5, af="relu")
        self.fusion_final = MutanFusion(1240, 310, 510, 2)
        self.linear_classif = MyLinear(510, self.num_classes, p=0.5)

        self.compress_q_1 = MyLinear(2400, 310, p=0.5, af="relu")
        self.expand_q_1 = MyLinear(310, 2048, p=0.5, af="sigmoid")
        self.compress_q_2 = MyLinear(2400, 310, p=0.5, af="relu")
        self.expand_q_2 = MyLinear(310, 2048, p=0.5, af="sigmoid")
        self.alpha_dict = {}
        # compress_v / compress_v2 feed their relu output to fusion_vq1 / fusion_vq2 and nothing else, so the fusion's data
        # gradient can apply the relu gate in its store (ops.lowrank_bilinear_fusion, gate_dx) and the projections' backward
        # kernels run ungated.  Measured at B = 512 (profiles/README.md): what the projections save (25 us) the fusion's
        # epilogue pays back in dependent loads of the gate (2 x 13 us) -- off by default, VQA_FUSE_RELU_GATE=1 turns it on.
        # (bf16: the gate rides in the 16-byte row stores of K4's data-gradient GEMM for free -- always on there)
        self.compress_v.grad_pregated = self.compress_v2.grad_pregated = \
            os.environ.get("VQA_FUSE_RELU_GATE", "0") == "1" or self.compute_dtype == torch.bfloat16
        self._plan = None

    def _bf16_shadows(self):
        """The bf16 / padded-fp32 shadows of the region-side masters (compress_v, compress_v2, the two region fusions), all
        packed by ONE kernel (ops.ShadowPlan): -> {layer: the `packed` tuple its op takes}.  Called at the top of every
        mixed-precision forward; the backward of the same step reads the transposed ones."""
        if self._plan is None or self._plan_dev != self.compress_v.conv.weight.device:
            plan, shad = ops.ShadowPlan(), {}
            for name in ("compress_v", "compress_v2"):
                conv = getattr(self, name).conv
                out_f, in_f = conv.weight.shape[0], conv.weight.shape[1]
                Np, Kp = ops.pad_to(out_f), ops.pad_to(in_f)
                wp = plan.add(conv.weight, ops._shadow(conv.weight, (Np, Kp), "nk"), Kp)
                bp = plan.add(conv.bias, ops._shadow(conv.bias, (Np,), "b", torch.float32), Np)
                wpt = None
                if name == "compress_v2":       # (compress_v reads the model input: no data gradient, no W^T)
                    wpt = plan.add(conv.weight, ops._shadow(conv.weight, (Kp, Np), "kn"), 1, Np)
                shad[name] = (wp, bp, wpt)
            for name in ("fusion_vq1", "fusion_vq2"):
                mf = getattr(self, name)
                R, H, L = mf.R, mf.hidden_dim, mf.input_dim1
                Hp, Lp = ops.pad_to(H, 256), ops.pad_to(L)
                w0 = mf.list_linear1[0].linear.weight
                w1p = ops._shadow(w0, (R, Hp, Lp), "k4")
                b1p = ops._shadow(mf.list_linear1[0].linear.bias, (R, Hp), "k4b", torch.float32)
                w1t = ops._shadow(w0, (Lp, R * Hp), "k4t")
                for r, lin in enumerate(mf.list_linear1):
                    plan.add(lin.linear.weight, w1p, Lp, 1, offset=r * Hp * Lp)
                    plan.add(lin.linear.bias, b1p, Hp, 1, offset=r * Hp)
                    plan.add(lin.linear.weight, w1t, 1, R * Hp, offset=r * Hp)      # w1t[l, r*Hp + h] = W1_r[h, l]
                shad[name] = (w1p, b1p, w1t)
            self._plan, self._shad, self._plan_dev = plan, shad, self.compress_v.conv.weight.device
        self._plan.pack()
        return self._shad

    def late_parameters(self):
        """Parameters of the second reasoning step.  Their gradients are complete once backward has walked from the loss
        to the tensors ``forward_with_cut`` reports -- before the first step's layers are touched -- so a data-parallel
        trainer can start reducing them while the rest of backward runs (trainer.DataParallelTrainer, overlap=True)."""
        mods = [self.compress_v2, self.fusion_vq2, self.att2, self.fusion_final, self.linear_classif]
        return [p for m in mods for p in m.parameters()]

    def forward_with_cut(self, sample):
        """-> (logits, outs, ins): ``outs`` are the first reasoning step's results that the second one reads, ``ins`` the
        detached aliases it actually consumed.  d loss / d ins fed as grad_outputs of ``outs`` completes the backward."""
        cut = []
        logits = self(sample, cut)
        return logits, cut[0], cut[1]

    def stack_groups(self):
        return linear_stack_groups([self.compress_q, self.linear_q, self.compress_q_1, self.compress_q_2]) + \
            linear_stack_groups([self.expand_q_1, self.expand_q_2])

    def question_projections(self, q_feature):
        """The four MyLinear(2400 -> 310) that read the question vector (config/CoR2.py:170,180,183,186, applied at
        :205,:193-194,:230) as one batched GEMM, then the two sigmoid gates expand_q_{1,2}(.) as another.
        -> (q_feature_low, q_final, q_gate_1 [B,2048], q_gate_2 [B,2048])."""
        low = my_linears([self.compress_q, self.linear_q, self.compress_q_1, self.compress_q_2], q_feature,
                         group_first=True)                                                          # [4,B,310]
        q_feature_low, q_final, low_gates = ops.split_groups(low, (1, 1, 2))      # views; one cat kernel backward
        gates = my_linears([self.expand_q_1, self.expand_q_2], low_gates.transpose(0, 1), group_first=True)   # [2,B,2048]
        q_gate_1, q_gate_2 = ops.split_groups(gates, (1, 1))
        return q_feature_low, q_final, q_gate_1, q_gate_2

    def _grouped_head_ok(self, q_feature, cut):
        """The [B,.]-sized layers run as grouped phases (head.py) whenever the tensors are fp32 on the GPU, the question
        vector needs no gradient (no trainable encoder in front), the backward is not cut in two (the cut crosses the
        phases) and the layer widths are even (8-byte operand loads)."""
        mods = [self.compress_q, self.linear_q, self.compress_q_1, self.compress_q_2, self.expand_q_1, self.expand_q_2]
        return (cut is None and q_feature.is_cuda and q_feature.dtype == torch.float32 and not q_feature.requires_grad
                and all(m.af == ("sigmoid" if i >= 4 else "relu") for i, m in enumerate(mods))
                and all(m.p == mods[0].p for m in mods) and self.linear_classif.af in (None, "")
                and self.att1.grouped_ok(q_feature) and self.att2.grouped_ok(q_feature)
                and head.supported("cor2", q_feature.size(1), self.compress_q.out_features, self.expand_q_1.out_features,
                                   self.fusion_vq1.hidden_dim, self.fusion_final.hidden_dim, self.fusion_final.input_dim1,
                                   self.att1.att_dim, self.num_classes))

    def _question_phases(self, q_feature):
        """Phases 1 and 2 of the grouped head: the four question projections, then everything that reads them -- the two
        sigmoid gates and the question-side rank factors of all three Mutan fusions.
        -> (q_gate_1, q_gate_2, h2 of fusion_vq1, of fusion_vq2, of fusion_final)."""
        proj = [self.compress_q, self.linear_q, self.compress_q_1, self.compress_q_2]
        p_in = float(proj[0].p) if (self.training and proj[0].p) else 0.0
        p_g = float(self.expand_q_1.p) if (self.training and self.expand_q_1.p) else 0.0
        lows = head.QuestionProjections.apply(q_feature.contiguous(), p_in, ops.next_dropout_seed() if p_in else 0, (2, 3), p_g,
                                              ops.next_dropout_seed() if p_g else 0, (),
                                              *[m.linear.weight for m in proj], *[m.linear.bias for m in proj])
        s = 1.0 / (1.0 - p_g) if p_g else 1.0
        fus = [self.fusion_vq1, self.fusion_vq2, self.fusion_final]
        params = [self.expand_q_1.linear.weight, self.expand_q_1.linear.bias, self.expand_q_2.linear.weight,
                  self.expand_q_2.linear.bias]
        for mf in fus:
            params += [lin.linear.weight for lin in mf.list_linear2] + [lin.linear.bias for lin in mf.list_linear2]
        return head.GatesAndRankFactors.apply(4, (2, 3), ((0, fus[0].R), (0, fus[1].R), (1, fus[2].R)), (1.0, 1.0, s, s),
                                              *lows, *params)

    def _final_phases(self, v1_att, v2_att, h2_final):
        """Phases 5 and 6: fusion_final (its first input is cat(v1_att, v2_att), never materialised) with the rank product and
        the classifier's input dropout in its epilogue, then the classifier."""
        p_c = float(self.linear_classif.p) if (self.training and self.linear_classif.p) else 0.0
        seed = ops.next_dropout_seed() if p_c else 0
        lins = list(self.fusion_final.list_linear1)
        x = head.VectorFusion.apply(h2_final, p_c, seed, 2, v1_att, v2_att, *[l.linear.weight for l in lins],
                                    *[l.linear.bias for l in lins])
        return head.Classifier.apply(x, self.linear_classif.linear.weight, self.linear_classif.linear.bias, p_c, seed)

    def relation_reduce(self, v_feature, q_gate_1, q_gate_2, alpha):
        """config/CoR2.py:191-199 (decare_cat) + :216 fused: v2[b,j] = sum_i alpha[b,i,0] *
        (v[b,i]*q1[b] + v[b,j]*q2[b]); the [B,N,N,D] tensor is never built (HIP kernel K1)."""
        return ops.pairwise_relation_reduce(v_feature, q_gate_1, q_gate_2, alpha, glimpse=0,
                                            mode=self.relation_mode, dual=True)

    def forward(self, sample, _cut=None):
        v = sample["v"]
        b = v.size(0)
        v_feature = v.contiguous().view(b, -1, 2048)
        if v_feature.dtype != self.compute_dtype:
            # bf16 regions into the fp32 path: the feed's transport format (feed.store_batches(region_dtype=torch.bfloat16): half
            # the host -> device bytes), widened exactly by one HIP pass; everything downstream is the fp32 step
            if v_feature.is_cuda and v_feature.dtype == torch.bfloat16 and self.compute_dtype == torch.float32:
                v_feature = ops.widen_bf16(v_feature)
            else:
                v_feature = v_feature.to(self.compute_dtype)
        q_feature = question_feature(self.seq2vec, sample["q_idxes"] if "q_idxes" in sample else sample["q"])

        shad = self._bf16_shadows() if v_feature.dtype == torch.bfloat16 and v_feature.is_cuda else {}
        grouped = self._grouped_head_ok(q_feature, _cut)
        h2_1 = h2_2 = h2_final = q_feature_low = q_final = None
        if grouped:
            q_gate_1, q_gate_2, h2_1, h2_2, h2_final = self._question_phases(q_feature)
        else:
            q_feature_low, q_final, q_gate_1, q_gate_2 = self.question_projections(q_feature)
        v_feature_low = self.compress_v(v_feature, packed=shad.get("compress_v"))
        fuse1 = self.fusion_vq1(v_feature_low, q_feature_low, relu_input=self.compress_v.grad_pregated,
                                packed=shad.get("fusion_vq1"), h2=h2_1)
        v1_att, alpha1, alpha1_full, pooled1_first = self.att1.attend(v_feature, self.att1.conv_att.pre_activation(fuse1),
                                                                return_pooled=True, grouped=grouped)
        if _cut is not None:
            # everything the second reasoning step (relation, compress_v2, fusion_vq2, att2, fusion_final, classifier)
            # reads from the first.  The second step runs on detached aliases of these tensors, so a backward pass from the
            # loss ends at the aliases (some of the originals are ancestors of others -- q_feature_low of v1_att -- and
            # would otherwise drag the first step's layers into it); a second pass then carries the aliases' gradients
            # from the originals to the inputs (see late_parameters / trainer.DataParallelTrainer overlap)
            outs = [q_feature_low, q_final, q_gate_1, q_gate_2, pooled1_first, v1_att, alpha1_full]
            ins = [t.detach().requires_grad_(t.requires_grad) for t in outs]
            _cut.extend([outs, ins])
            q_feature_low, q_final, q_gate_1, q_gate_2, pooled1_first, v1_att, alpha1_full = ins

        cv2 = self.compress_v2
        w2 = cv2.conv.weight.squeeze(-1)
        fused_node = cv2.af == "relu" and cv2.fused and b * v_feature.size(1) >= 1024 and v_feature.dtype == torch.float32 and \
            ops.relation_projection_supported(v_feature, w2)
        if self.relation_mode == 1 or (fused_node and ops.pairwise_projection_supported(v_feature)):
            # Closed form of the relation step (config/CoR2.py:191-199 + :216).  Only glimpse 0 of alpha1 weights it, and
            # sum_i alpha1[i,0] v_i is exactly glimpse 0 of the pooled features att1 has just produced, while a softmax
            # alpha sums to 1 -- so v2[b,n] = t[b] + q2[b] * v[b,n] with t = q1 * pooled1[:,0].  (The two unit sums only
            # reach the attention maps as a per-glimpse constant, which the softmax backward cancels exactly.)  v2 is
            # materialised once, already dropped out for compress_v2 (K1 apply kernel); the second attention pools v
            # itself and maps the result: sum_n alpha2[n] v2[n] = t + q2 * sum_n alpha2[n] v[n].
            # (t, c2) for the relation / projection node and (t_b, c2_b) -- the same values -- for the pooled map of the second
            # attention: two handles, so that both gradients meet in ONE backward kernel (ops.RelationGates)
            if q_gate_1.is_cuda and q_gate_1.dtype == torch.float32 and q_gate_1.numel() % 4 == 0:
                t, c2, t_b, c2_b = ops.relation_gates(q_gate_1, q_gate_2, pooled1_first)
            else:
                t = t_b = q_gate_1 * pooled1_first
                c2 = c2_b = q_gate_2
            p = self.compress_v2.p if (self.training and self.compress_v2.p) else 0.0
            if fused_node:
                # relation step + projection as one autograd node: backward reduces the projection's data gradient to
                # d_t / d_c2 inside the GEMM tile (csrc/relation_dgrad.hip) instead of writing and re-reading [B,N,2048].
                # relation_mode 0: the node's forward builds the relation tensor with the PAIRWISE kernel (every (i, j) term,
                # the reference's structure); everything else -- the backward through (t, c2), the second attention pooling v
                # itself -- is shared with the closed form, which it equals.
                pairwise = (q_gate_1, q_gate_2, alpha1_full, 0) if self.relation_mode != 1 else None
                v2_feature_low = ops.relation_projection(v_feature, t, c2, w2, cv2.conv.bias, p,
                                                         ops.next_dropout_seed() if p else 0, cv2.grad_pregated, pairwise)
            else:
                v2_dropped = ops.relation_apply(v_feature, t, c2, p, ops.next_dropout_seed() if p else 0)
                v2_feature_low = cv2(v2_dropped, predropped=True, packed=shad.get("compress_v2"))
            fuse2 = self.fusion_vq2(v2_feature_low, q_feature_low, relu_input=self.compress_v2.grad_pregated,
                                    packed=shad.get("fusion_vq2"), h2=h2_2)
            v2_att, alpha2, _ = self.att2.attend(v_feature, self.att2.conv_att.pre_activation(fuse2),
                                                 lambda pooled, pd: ops.relation_apply(
                                                     pooled, t_b, c2_b, pd, ops.next_dropout_seed() if pd else 0),
                                                 grouped=grouped)
            # the reference's v2_feature[:, [0, 1], :] (visu.py:198-207 reads it after an eval forward).  Eval: computed here,
            # like the reference does.  Training: a step never looks at it, so it is computed when read, from copies of the
            # two region rows and of t / c2 rows that belong to THIS forward only when it ran eagerly -- under graph replay
            # t and c2 are the graph's own buffers and hold the latest replay's values (documented on SideOutputs).  Only
            # the [:, 0:2] slice of v is kept alive, not the [B,N,2048] input.
            # (the slice keeps its storage dtype until it is read: for bf16 regions .float() is a conversion kernel, and a
            #  training step never reads the feature)
            t_d, c2_d, v_d = t.detach(), c2.detach(), v_feature.detach()[:, 0:2, :]
            if self.training:
                feature = lambda: torch.addcmul(t_d.unsqueeze(1), c2_d.unsqueeze(1), v_d.float())  # noqa: E731
            else:
                feature = torch.addcmul(t_d.unsqueeze(1), c2_d.unsqueeze(1), v_d.float())
        else:
            # pairwise form: every (i, j) term of the relation tensor summed in the kernel, as the reference structures it;
            # v2 has two consumers, each gets its own alias (see ops.pairwise_relation_reduce)
            v2_feature, v2_for_pooling = self.relation_reduce(v_feature, q_gate_1, q_gate_2, alpha1_full)
            v2_feature_low = self.compress_v2(v2_feature, packed=shad.get("compress_v2"))
            fuse2 = self.fusion_vq2(v2_feature_low, q_feature_low, relu_input=self.compress_v2.grad_pregated,
                                    packed=shad.get("fusion_vq2"), h2=h2_2)
            v2_att, alpha2, _ = self.att2.attend(v2_for_pooling, self.att2.conv_att.pre_activation(fuse2), grouped=grouped)
            feature = v2_feature[:, 0:2, :].detach().float()

        # side output read by visu.py:198-207; detached so it does not pin the autograd graph of the step
        # (feature = the reference's v2_feature[:, [0, 1], :])
        self.alpha_dict = SideOutputs({"alpha1": tuple(t_.detach() for t_ in alpha1), "alpha2": tuple(t_.detach() for t_ in alpha2),
                                       "feature": feature})

        if grouped:
            return self._final_phases(v1_att, v2_att, h2_final)
        v_f = torch.cat([v1_att, v2_att], dim=1)
        x = self.fusion_final(v_f, q_final)
        return self.linear_classif(x)
