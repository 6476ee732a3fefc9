"""The repo-root ``config.CoR2`` / ``config.ODA`` modules expose the reference's module surface
(train.py does importlib.import_module(args.cf) and reads these attributes)."""
import importlib

import pytest

ATTRS = ["data_dir", "process_dir", "log_dir", "analyze_dir", "method_name", "version", "samplingans",
         "loss_metric", "vgenome", "version1_multiple_choices", "arch", "size", "nans", "splitnum", "mwc", "mql",
         "target_list", "epochs", "resume", "print_freq", "lr", "load_mem",
         "batch_size", "clip_grad", "test_dev_range", "test_range", "debug", "Model", "MyConv1d", "MyLinear", "MyATT"]


@pytest.mark.parametrize("name,nans,epochs,bs", [("config.CoR2", 2000, 70, 100), ("config.ODA", 3000, 100, 256)])
def test_config_module_surface(name, nans, epochs, bs):
    cf = importlib.import_module(name)
    for a in ATTRS:
        assert hasattr(cf, a), a
    if name == "config.CoR2":  # ODA leaves these two to train.py's defaults (train.py:323-447)
        assert cf.restart_epoch is None and cf.keeping_epoch == 40 and cf.load_mem is None
    else:
        assert cf.load_mem == "DB" and list(cf.test_dev_range) == [100]
    assert cf.nans == nans and cf.epochs == epochs and cf.batch_size == bs
    assert cf.lr == 1e-4 and cf.loss_metric == "KLD" and cf.arch == "rcnn" and cf.splitnum == 2
    assert cf.method_name.endswith("_VAL") and cf.log_dir.endswith("_VAL") and cf.analyze_dir.endswith("_VAL")
    assert not hasattr(cf, "sgd")  # train.py:402-406 requires this
    m = cf.Model(["PAD", "UNK"], cf.nans)
    assert hasattr(m, "seq2vec") and hasattr(m, "alpha_dict")
