"""Checkpoint compatibility with the reference's ``.tar`` format (utils.py:9-25), on the CPU: state_dicts captured from
the REFERENCE's modules (the golden fixtures) are written in its format, loaded into the drop-in modules, saved again
by the drop-in helpers and compared key by key — both directions without a GPU (no kernel runs)."""
import json
import os

import numpy as np
import pytest
import torch

import feature_level_style_transfer_for_tsc_amd as fst

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _reference_states():
    g = dict(np.load(os.path.join(GOLDEN, "joint_small.npz"), allow_pickle=False))
    meta = json.loads(str(g["meta"]))
    sub = lambda prefix: {k[len(prefix):]: torch.tensor(v) for k, v in g.items() if k.startswith(prefix)}
    tup = lambda lp: [[tuple(t) for t in l] for l in lp]
    return meta, sub, tup


def test_reference_checkpoints_round_trip(tmp_path):
    meta, sub, tup = _reference_states()
    ref_t = {"epoch": 7, "feature_extraction_state_dict": sub("sd0.fe_t."), "classification_state_dict": sub("sd0.clf_t.")}
    ref_s = {"epoch": 9, "feature_extraction_state_dict": sub("sd0.fe_s."),
             "source_to_target_feature_trans": sub("sd0.dimunif."), "classification_state_dict": sub("sd0.clf_s.")}
    pt, ps = str(tmp_path / "epoch_7.tar"), str(tmp_path / "epoch_9_source.tar")
    torch.save(ref_t, pt); torch.save(ref_s, ps)                      # what the reference's utils.py writes

    C = sum(t[1] for t in meta["lp_t"][-1])
    C_s = sum(t[1] for t in meta["lp_s"][-1])
    fe_t, clf_t = fst.OS_CNN_res(tup(meta["lp_t"])), fst.OS_CNN(tup(meta["lp_clf"]), meta["ncls_t"])
    fe_s, clf_s = fst.OS_CNN_res(tup(meta["lp_s"])), fst.OS_CNN(tup(meta["lp_clf"]), meta["ncls_s"])
    du = fst.DimensionUnification(C_s, C, meta["L_s"], meta["L_t"])
    assert fst.load_target_classification_modules(pt, fe_t, clf_t) == 7
    assert fst.load_source_classification_modules(ps, fe_s, du, clf_s) == 9

    out_t = fst.save_target_classification_modules(fe_t, clf_t, 7, path=str(tmp_path / "out" / "epoch_7.tar"))
    out_s = fst.save_source_classification_modules(fe_s, du, clf_s, 9, path=str(tmp_path / "out" / "epoch_9_source.tar"))
    for ref, path in ((ref_t, out_t), (ref_s, out_s)):
        ck = torch.load(path, map_location="cpu", weights_only=True)
        assert list(ck) == list(ref)                                   # same top-level keys, same order
        for part, sd in ref.items():
            if part == "epoch":
                assert ck["epoch"] == sd
                continue
            assert list(ck[part]) == list(sd), part                    # same state_dict keys, same order
            for k, v in sd.items():
                assert ck[part][k].dtype == v.dtype and torch.equal(ck[part][k], v), (part, k)


def test_wrong_checkpoint_is_rejected(tmp_path):
    meta, sub, tup = _reference_states()
    p = str(tmp_path / "bad.tar")
    torch.save({"epoch": 1, "feature_extraction_state_dict": sub("sd0.fe_t.")}, p)
    fe_t, clf_t = fst.OS_CNN_res(tup(meta["lp_t"])), fst.OS_CNN(tup(meta["lp_clf"]), meta["ncls_t"])
    with pytest.raises(KeyError):
        fst.load_target_classification_modules(p, fe_t, clf_t)
    sd = sub("sd0.clf_t.")
    sd.pop("hidden.bias")
    torch.save({"epoch": 1, "feature_extraction_state_dict": sub("sd0.fe_t."), "classification_state_dict": sd}, p)
    with pytest.raises(RuntimeError):
        fst.load_target_classification_modules(p, fe_t, clf_t)
