"""End-to-end training loop (reference train.py semantics) on the shrunken synthetic cohort: the loss
must be finite and go down; eval metrics must be produced."""
import math
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_training_loop_reduces_the_loss():
    from conftest import PKG
    sys.path.insert(0, PKG)
    import train_harness as th
    cfg = os.path.join(os.path.dirname(__file__), "golden", "gbm_like.yaml")
    args = th.parse_opts(["--config", cfg, "--small", "--patients", "128", "--epochs", "6", "--batch_size", "16",
                          "--lr", "0.003", "--head_dim", "32", "--dropout", "0.0"])
    args.feature_drop = False
    hist = th.run(args)
    assert len(hist) == 6
    assert all(math.isfinite(h["train_loss"]) and math.isfinite(h["valid_loss"]) for h in hist)
    assert hist[-1]["train_loss"] < hist[0]["train_loss"]
    assert 0.0 <= hist[-1]["valid_acc"] <= 1.0 and hist[-1]["graphs_per_s"] > 0
