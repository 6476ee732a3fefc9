"""Drop-in for the reference's ``config/ODA.py``: the same module-level hyper-parameters
(config/ODA.py:8-70 of the reference) and the names ``MyConv1d / MyLinear / MyATT / Model`` --
backed by the MI355X HIP kernels.

    python train.py --cf config.ODA
"""
import os

from vqa_playground_pytorch_amd.encoder import BayesianGRU, SkipThoughts  # noqa: F401
from vqa_playground_pytorch_amd.layers import (MutanFusion, MyATT, MyConv1d, MyLinear,  # noqa: F401
                                               bmatmul, bmul)
from vqa_playground_pytorch_amd import oda as _impl

# Preprocess (relative paths by default, like the reference's fall-through branch; override with VQA_DATA_DIR)
YOUR_DATA_DIR = os.environ.get("VQA_DATA_DIR", "")
data_dir = os.path.join(YOUR_DATA_DIR, "data/VQA/download")
process_dir = os.path.join(YOUR_DATA_DIR, "data/VQA/preprocess")
log_dir = os.path.join(YOUR_DATA_DIR, "data/VQA/logs")
analyze_dir = os.path.join(YOUR_DATA_DIR, "data/VQA/analyze")

version = 2
samplingans = False
loss_metric = "KLD"
vgenome = False
version1_multiple_choices = False
# Process_img
arch = "rcnn"
size = 224

# Process_qa
nans = 3000
splitnum = 2
mwc = 0
mql = 26

# Train
target_list = ["v", "q_id", "q_idxes"]
epochs = 100

resume = True
print_freq = 10
lr = 0.0001
load_mem = "DB"
batch_size = 256
clip_grad = True
# if test_dev is None, skip
test_dev_range = range(epochs, epochs + 5, 5)
test_range = None
debug = False

method_name = os.path.splitext(os.path.basename(__file__))[0]
if splitnum == 2:
    method_name += "_VAL"
log_dir = os.path.join(log_dir, method_name)
analyze_dir = os.path.join(analyze_dir, method_name)

# Question encoder of the two-argument constructor ``Model(vocab_words, num_ans)`` (train.py:515):
#   "skipthoughts" -- what the reference builds (config/ODA.py:183): embedding(620) + BayesianGRU(2400) over int64 token ids
#                     [B,26]; randomly initialised here (the uni-skip weight files are downloaded by the reference and
#                     are not available offline: load them with model.seq2vec.load_pretrained(...) or a checkpoint);
#   "vector"       -- no encoder: sample['q_idxes'] already holds the 2400-d question vector (benchmarks, BASELINE.json).
# Either way a floating [B,2400] 'q_idxes' is taken as the question vector itself.
question_encoder = os.environ.get("VQA_SEQ2VEC", "skipthoughts")


class Model(_impl.Model):
    def __init__(self, vocab_words=None, num_ans=None, seq2vec=None, **kwargs):
        super().__init__(vocab_words, num_ans, seq2vec=question_encoder if seq2vec is None else seq2vec, **kwargs)
