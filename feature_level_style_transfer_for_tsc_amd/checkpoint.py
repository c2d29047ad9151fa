"""Checkpoints in the reference's ``.tar`` format (utils.py:9-25) and the evaluation pass (utils.py:27-183).

The drop-in modules keep the reference's ``state_dict`` keys, so a checkpoint written by the reference loads here
and vice versa: a ``torch.save``d dict with ``epoch`` and ``feature_extraction_state_dict`` /
``classification_state_dict`` (+ ``source_to_target_feature_trans`` on the source side).  Non-persisted state of the
reference stays non-persisted here too (omni-scale masks are rebuilt from the layer spec, WaveGlow's cached
``W_inverse`` — quirk Q2 — is recomputed on first ``infer``).
"""
from __future__ import annotations

import os
from typing import Iterable, Optional, Tuple

import torch

TARGET_KEYS = ("epoch", "feature_extraction_state_dict", "classification_state_dict")
SOURCE_KEYS = ("epoch", "feature_extraction_state_dict", "source_to_target_feature_trans", "classification_state_dict")


def _cpu_state(module) -> dict:
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def save_target_classification_modules(target_feature_extraction_module, target_classification_module, cur_epoch,
                                       path: Optional[str] = None) -> str:
    """utils.py:9-16; default path ``train_log/epoch_<n>.tar`` as in the reference."""
    path = path or os.path.join("train_log", f"epoch_{cur_epoch}.tar")
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save({"epoch": cur_epoch,
                "feature_extraction_state_dict": _cpu_state(target_feature_extraction_module),
                "classification_state_dict": _cpu_state(target_classification_module)}, path)
    return path


def save_source_classification_modules(source_feature_extraction_module, source_to_target_feature_trans,
                                       source_classification_module, cur_epoch, path: Optional[str] = None) -> str:
    """utils.py:19-26; default path ``train_log/epoch_<n>_source.tar``."""
    path = path or os.path.join("train_log", f"epoch_{cur_epoch}_source.tar")
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save({"epoch": cur_epoch,
                "feature_extraction_state_dict": _cpu_state(source_feature_extraction_module),
                "source_to_target_feature_trans": _cpu_state(source_to_target_feature_trans),
                "classification_state_dict": _cpu_state(source_classification_module)}, path)
    return path


def load_target_classification_modules(path: str, target_feature_extraction_module, target_classification_module) -> int:
    """Load a target-side checkpoint (written here or by the reference); returns its epoch.  Key mismatches raise
    (``strict=True``) — a silently half-loaded model would void every parity claim."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    missing = [k for k in TARGET_KEYS if k not in ck]
    if missing:
        raise KeyError(f"{path}: not a target classification checkpoint (missing {missing})")
    target_feature_extraction_module.load_state_dict(ck["feature_extraction_state_dict"], strict=True)
    target_classification_module.load_state_dict(ck["classification_state_dict"], strict=True)
    return int(ck["epoch"])


def load_source_classification_modules(path: str, source_feature_extraction_module, source_to_target_feature_trans,
                                       source_classification_module) -> int:
    ck = torch.load(path, map_location="cpu", weights_only=True)
    missing = [k for k in SOURCE_KEYS if k not in ck]
    if missing:
        raise KeyError(f"{path}: not a source classification checkpoint (missing {missing})")
    source_feature_extraction_module.load_state_dict(ck["feature_extraction_state_dict"], strict=True)
    source_to_target_feature_trans.load_state_dict(ck["source_to_target_feature_trans"], strict=True)
    source_classification_module.load_state_dict(ck["classification_state_dict"], strict=True)
    return int(ck["epoch"])


@torch.no_grad()
def eval_accuracy(feature_extraction_module, classification_module, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]],
                  feature_trans=None) -> Tuple[float, torch.Tensor]:
    """Accuracy of ``argmax(classifier(extractor(x)))`` over ``batches`` of (x, y) — the loop of utils.py:27-183 with
    predictions kept on the device and one host read at the end.  The modules must already be in eval mode, as the
    reference puts them before calling (train_and_test.py:175-176); ``feature_trans`` is the source side's
    DimensionUnification.  Returns (accuracy, predictions)."""
    from . import ops
    dev = next(classification_module.parameters()).device
    preds, hits, n = [], torch.zeros((), device=dev, dtype=torch.int64), 0
    with ops.pack_cache():            # eval-mode BatchNorm folded into the convs; packed weight images shared by all batches
        for x, y in batches:
            f = feature_extraction_module(x.float().to(dev))
            if feature_trans is not None:
                f = feature_trans(f)
            p = classification_module(f)[0].argmax(dim=1)
            preds.append(p)
            hits += (p == y.to(dev)).sum()
            n += int(y.numel())
    if n == 0:
        raise ValueError("eval_accuracy: no samples")
    return float(hits.item()) / n, torch.cat(preds)
