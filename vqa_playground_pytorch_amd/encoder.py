"""Question encoder slot ``seq2vec``: SkipThoughts = embedding(620) + BayesianGRU(620 -> 2400), the step in front of the
hot path (putils/__init__.py:878-985 SkipThoughts, :604-746 BayesianGRUCell / BayesianGRU, :503-539 SequentialDropout).

Same module tree and parameter names as the reference (``embedding.weight``, ``gru.gru_cell.weight_{ir,ii,in}.{weight,
bias}``, ``gru.gru_cell.weight_{hr,hi,hn}.weight``), so a reference checkpoint's ``seq2vec.*`` entries load.  The
uni-skip weight files the reference downloads at construction (putils/__init__.py:902-911) are not fetched here: pass
``pretrained={'utable': ..., 'dictionary': [...], 'uni_skip': {...}}`` to load them, otherwise the module is randomly
initialised.  Pure torch ops (library GEMMs) -- upstream of the path north_star names; what is restructured is the data
flow: the three input-side projections of all T steps are three [B*T,620]x[620,2400] GEMMs instead of 3*T small ones,
and the last valid state is gathered instead of masked-and-summed.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def _af(name):
    return {"tanh": torch.tanh, "relu": F.relu, "sigmoid": torch.sigmoid}[name]


class BayesianGRUCell(nn.Module):
    """putils/__init__.py:604-646 (+ AbstractGRUCell :566-584): six nn.Linear; bias only on the input side."""

    def __init__(self, input_size, hidden_size, bias_ih=True, bias_hh=False, dropout=0.25, af="tanh"):
        super().__init__()
        self.input_size, self.hidden_size, self.dropout, self.af = input_size, hidden_size, dropout, af
        self.weight_ir = nn.Linear(input_size, hidden_size, bias=bias_ih)
        self.weight_ii = nn.Linear(input_size, hidden_size, bias=bias_ih)
        self.weight_in = nn.Linear(input_size, hidden_size, bias=bias_ih)
        self.weight_hr = nn.Linear(hidden_size, hidden_size, bias=bias_hh)
        self.weight_hi = nn.Linear(hidden_size, hidden_size, bias=bias_hh)
        self.weight_hn = nn.Linear(hidden_size, hidden_size, bias=bias_hh)


class BayesianGRU(nn.Module):
    """putils/__init__.py:672-746.  The six dropout masks are drawn once per sequence and shared by all time steps
    (SequentialDropout); ``forward(x [B,T,in], lengths [B]) -> [B,hidden]`` = the state at step lengths-1."""

    def __init__(self, input_size, hidden_size, bias_ih=True, bias_hh=False, dropout=0.25, return_last=True, af="tanh"):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        self.dropout, self.return_last, self.af = dropout, return_last, af
        self.gru_cell = BayesianGRUCell(input_size, hidden_size, bias_ih, bias_hh, dropout=dropout, af=af)
        self.all_hiddens = None

    def _mask(self, like):
        if not (self.training and self.dropout > 0):
            return None
        return torch.bernoulli(torch.full_like(like, 1.0 - self.dropout)) / (1.0 - self.dropout)

    def stack_groups(self):
        c = self.gru_cell
        inp, hid = (c.weight_ir, c.weight_ii, c.weight_in), (c.weight_hr, c.weight_hi, c.weight_hn)
        groups = [[m.weight for m in inp], [m.weight for m in hid]]
        if inp[0].bias is not None:
            groups.append([m.bias for m in inp])
        return groups

    def _forward_hip(self, x, lengths):
        """GPU form: the three input projections of all T steps as one batched GEMM, then ops.GruSequence (per step one
        batched recurrent GEMM + one gate kernel; csrc/gru.hip)."""
        c = self.gru_cell
        B, T, K = x.shape
        inp, hid = (c.weight_ir, c.weight_ii, c.weight_in), (c.weight_hr, c.weight_hi, c.weight_hn)
        xg = x.reshape(B * T, 1, K).expand(B * T, 3, K)                        # stride-0 gate axis
        if self.training and self.dropout > 0:
            mx = torch.stack([self._mask(x[:, :1, :]) for _ in range(3)], 2)   # [B,1,3,K]: shared over time
            xg = (x.unsqueeze(2) * mx).reshape(B * T, 3, K)
            mh = torch.stack([self._mask(x.new_zeros(B, self.hidden_size)) for _ in range(3)])   # [3,B,H]
        else:
            mh = None
        w_in = ops.stack_params([m.weight for m in inp])
        b_in = ops.stack_params([m.bias for m in inp]) if inp[0].bias is not None else None
        gi = ops.batched_linear(xg, w_in, b_in, group_first=True).view(3, B, T, self.hidden_size)
        out = ops.gru_sequence(gi, ops.stack_params([m.weight for m in hid]), mh, self.af)      # [T,B,H]
        if not self.return_last:
            return out.transpose(0, 1)
        self.all_hiddens = out.detach().transpose(0, 1)
        idx = (lengths.long() - 1) % T
        return out[idx, torch.arange(B, device=x.device)]

    def forward(self, x, lengths=None):
        c, af = self.gru_cell, _af(self.af)
        B, T, _ = x.shape
        if x.is_cuda and self.af in ("relu", "tanh") and c.weight_hr.bias is None and self.hidden_size % 4 == 0:
            return self._forward_hip(x, lengths)
        mx = [self._mask(x[:, :1, :]) for _ in range(3)]                       # [B,1,in], shared over time
        h = x.new_zeros(B, self.hidden_size)
        mh = [self._mask(h) for _ in range(3)]                                 # [B,hidden]
        gi = [ops.linear(x if m is None else x * m, lin.weight, lin.bias)                # (replay-safe bias gradient)
              for lin, m in zip((c.weight_ir, c.weight_ii, c.weight_in), mx)]
        outs = []
        for t in range(T):
            hr, hi, hn = (h if m is None else h * m for m in mh)
            r = torch.sigmoid(gi[0][:, t] + c.weight_hr(hr))
            i = torch.sigmoid(gi[1][:, t] + c.weight_hi(hi))
            n = af(gi[2][:, t] + r * c.weight_hn(hn))
            h = (1 - i) * n + i * h
            outs.append(h)
        output = torch.stack(outs, dim=1)                                      # [B,T,hidden]
        if not self.return_last:
            return output
        self.all_hiddens = output.detach()
        idx = (lengths.long() - 1) % T                                         # length 0 picks the last step, as mask[i][-1] does
        return output[torch.arange(B, device=x.device), idx]


class SkipThoughts(nn.Module):
    """putils/__init__.py:878-985: ``forward(q_idxes int64 [B,T], 0 = PAD) -> [B,2400]``."""

    def __init__(self, vocab_list, data_dir=None, gru="BayesianGRU", return_last=True, af="tanh", pretrained=None):
        super().__init__()
        if gru != "BayesianGRU":
            raise ValueError("only the BayesianGRU encoder of config/CoR2.py:166 / config/ODA.py:183 is provided")
        self.vocab_list, self.data_dir, self.af = vocab_list, data_dir, af
        self.embedding = nn.Embedding(num_embeddings=len(vocab_list), embedding_dim=620, padding_idx=0)
        self.gru = BayesianGRU(input_size=620, hidden_size=2400, dropout=0.25, return_last=return_last, af=af)
        if pretrained is not None:
            self.load_pretrained(pretrained)

    def load_pretrained(self, pre):
        """pre = {'dictionary': list of words, 'utable': [n,620] array, 'uni_skip': mapping with encoder_W/Wx/b/bx/U/Ux}
        (the three files of putils/__init__.py:902-904); the mapping onto parameters follows :913-973."""
        word_to_vec = {w: pre["utable"][i] for i, w in enumerate(pre["dictionary"])}
        rows = [word_to_vec[w] if w in word_to_vec else word_to_vec["UNK"] for w in self.vocab_list]
        sk = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in pre["uni_skip"].items()}
        with torch.no_grad():
            self.embedding.weight.copy_(torch.as_tensor(rows, dtype=torch.float32).reshape(len(rows), 620))
            c = self.gru.gru_cell
            c.weight_ir.weight.copy_(sk["encoder_W"].t()[:2400])
            c.weight_ii.weight.copy_(sk["encoder_W"].t()[2400:])
            c.weight_in.weight.copy_(sk["encoder_Wx"].t())
            c.weight_ir.bias.copy_(sk["encoder_b"][:2400])
            c.weight_ii.bias.copy_(sk["encoder_b"][2400:])
            c.weight_in.bias.copy_(sk["encoder_bx"])
            c.weight_hr.weight.copy_(sk["encoder_U"].t()[:2400])
            c.weight_hi.weight.copy_(sk["encoder_U"].t()[2400:])
            c.weight_hn.weight.copy_(sk["encoder_Ux"].t())

    def forward(self, x, return_hidden=False):
        if x.dtype != torch.long:
            raise ValueError("SkipThoughts expects int64 token ids [B,T] (0 = PAD)")
        emb = ops.embedding(self.embedding.weight, x, self.embedding.padding_idx) if x.is_cuda else self.embedding(x)
        lengths = x.size(1) - x.eq(0).sum(1)
        out = self.gru(emb, lengths)
        return (out, self.gru.all_hiddens) if return_hidden else out
