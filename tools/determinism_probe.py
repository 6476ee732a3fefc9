#!/usr/bin/env python3
"""Is one forward + backward of the model bitwise reproducible?  Two fresh models with the same seeded parameters on the same
inputs (eval mode and training mode with the same torch seed); prints every tensor that differs between the runs.
    python tools/determinism_probe.py [cor2|oda] [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import reference_faithful as RF  # noqa: E402  (a tool, not the product)
from oracle import seeded  # noqa: E402
from vqa_playground_pytorch_amd import CoR2Model, ODAModel  # noqa: E402


def run(cls, nans, v, q, a, train, bf16=False):
    dev = torch.device("cuda:0")
    ref = seeded.load_state({"cor2": RF.CoR2Oracle, "oda": RF.ODAOracle}[cls](nans), 0)
    model = {"cor2": CoR2Model, "oda": ODAModel}[cls](["PAD", "UNK"], nans)
    model.load_state_dict(ref.state_dict())
    model.to(dev)
    model.train(train)
    torch.manual_seed(5)
    vin = torch.from_numpy(v).to(torch.bfloat16) if bf16 else torch.from_numpy(v).to(torch.bfloat16).float()
    logits = model({"v": vin.to(dev), "q_idxes": torch.from_numpy(q).to(dev)})
    RF.kld_sum_loss(logits, torch.from_numpy(a).to(dev)).backward()
    torch.cuda.synchronize()
    return logits.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}


def main():
    cls = sys.argv[1] if len(sys.argv) > 1 else "cor2"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    nans = 2000 if cls == "cor2" else 3000
    v, q, a = seeded.seeded_inputs(B, answers=nans, seed=77)
    for train in (False, True):
        base = run(cls, nans, v, q, a, train)
        for tag, other in (("same input again", run(cls, nans, v, q, a, train)), ("bf16 transport", run(cls, nans, v, q, a, train, bf16=True))):
            diff = [] if torch.equal(base[0], other[0]) else ["logits"]
            for n in base[1]:
                if not torch.equal(base[1][n], other[1][n]):
                    d = (base[1][n] - other[1][n]).abs().max().item() / max(base[1][n].abs().max().item(), 1e-30)
                    diff.append("%s (%.1e)" % (n, d))
            print("[%s train=%s] %s: %s" % (cls, train, tag, "bit-identical" if not diff else "DIFFERS: " + ", ".join(diff)), flush=True)


if __name__ == "__main__":
    main()
