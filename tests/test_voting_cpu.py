"""Multi-source voting (multi_source_voting.py:281-424): the golden vectors were produced by executing that block of
the reference itself (oracle/capture_fixtures.py) on stub loaders/models.  The oracle restatement and the batched
tensor op are both held to them; the op is device-agnostic tensor code, so this runs on the CPU (the GPU suite
repeats it on the device together with the K-model eval forward)."""
import os

import numpy as np
import pytest
import torch

from feature_level_style_transfer_for_tsc_amd import voting
from oracle import restatement as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def votes():
    return dict(np.load(os.path.join(GOLDEN, "voting_small.npz"), allow_pickle=False))


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_oracle_restatement_matches_the_reference_block(votes, case):
    w, scores, pred, acc = R.multi_source_vote(votes[f"{case}.train_logits"], votes[f"{case}.train_labels"],
                                               votes[f"{case}.test_logits"], votes[f"{case}.test_labels"])
    np.testing.assert_allclose(w, votes[f"{case}.weights"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(scores, votes[f"{case}.scores"], rtol=2e-6)
    assert np.array_equal(pred, votes[f"{case}.pred"]) and abs(acc - float(votes[f"{case}.acc"])) < 1e-12
    if case == "c":
        assert (w[:, 4] == 0).all()                                      # never-predicted class: 0/0 -> nan -> 0


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_batched_op_matches_the_reference_block(votes, case):
    t = lambda k: torch.tensor(votes[f"{case}.{k}"])
    w, scores, pred, acc = voting.multi_source_vote(t("train_logits"), t("train_labels"), t("test_logits"), t("test_labels"))
    np.testing.assert_allclose(w.numpy(), votes[f"{case}.weights"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(scores.numpy(), votes[f"{case}.scores"], rtol=2e-5)
    assert np.array_equal(pred.numpy(), votes[f"{case}.pred"]) and abs(acc - float(votes[f"{case}.acc"])) < 1e-12
