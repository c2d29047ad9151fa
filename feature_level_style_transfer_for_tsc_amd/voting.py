"""Multi-source voting (multi_source_voting.py:281-424) as batched on-device ops.

K checkpoints of the target-side pipeline (one per source domain) vote on the test set: each model's class
probabilities are sharpened by its confidence, ``p·(1 + 120·e^{−H(p)})``, scaled per class by ``9^{w_k[c]}`` where
``w_k[c]`` is the model's train-set precision for class c relative to the mean over models, and summed.  The
reference concatenates logits batch by batch on the host and loops over samples in numpy; here logits stay on the
device as one ``[K, N, C]`` tensor and the whole vote is a handful of batched tensor ops.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch


@torch.no_grad()
def precision_weights(train_logits: torch.Tensor, train_labels: torch.Tensor) -> torch.Tensor:
    """[K, N, C] logits + [N] labels -> [K, C] weights (:292-367): per-class precision of each model's train-set
    predictions (0 for a class it never predicts), divided by the mean over models; 0/0 -> 0."""
    K, N, C = train_logits.shape
    pred = train_logits.argmax(dim=2)                                             # [K, N]
    onehot = torch.nn.functional.one_hot(pred, C).to(torch.float64)                # [K, N, C]
    predicted = onehot.sum(dim=1)                                                  # [K, C]
    correct = (onehot * (pred == train_labels.view(1, N)).unsqueeze(2)).sum(dim=1)
    w = torch.where(predicted > 0, correct / predicted.clamp(min=1), torch.zeros_like(correct))
    return torch.nan_to_num(w / w.mean(dim=0, keepdim=True), nan=0.0, posinf=0.0, neginf=0.0)


@torch.no_grad()
def vote_scores(test_logits: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """[K, M, C] logits + [K, C] weights -> [M, C] summed scores (:406-422)."""
    p = torch.softmax(test_logits.float(), dim=2)
    H = -(torch.where(p > 0, p * torch.log(p), torch.zeros_like(p))).sum(dim=2, keepdim=True)
    scale = torch.pow(torch.tensor(9.0, dtype=torch.float64, device=p.device), weights.to(torch.float64)).unsqueeze(1)
    return (p.double() * (1.0 + 120.0 * torch.exp(-H.double())) * scale).float().sum(dim=0)


@torch.no_grad()
def multi_source_vote(train_logits: torch.Tensor, train_labels: torch.Tensor, test_logits: torch.Tensor,
                      test_labels: Optional[torch.Tensor] = None):
    """Returns (weights [K,C], scores [M,C], predictions [M], accuracy or None)."""
    w = precision_weights(train_logits, train_labels)
    scores = vote_scores(test_logits, w)
    pred = scores.argmax(dim=1)
    acc = None if test_labels is None else float((pred == test_labels.to(pred.device)).double().mean().item())
    return w, scores, pred, acc


@torch.no_grad()
def collect_logits(models: Sequence[Tuple[torch.nn.Module, torch.nn.Module]],
                   batches: Iterable[Tuple[torch.Tensor, torch.Tensor]]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eval-mode logits of K (feature extractor, classifier) pairs over the same batches: ([K, N, C], labels [N]),
    everything kept on the device (the loops at :283-291 / :370-404).  The modules must be in eval mode."""
    from . import ops
    dev = next(models[0][1].parameters()).device
    per_model: List[List[torch.Tensor]] = [[] for _ in models]
    labels: List[torch.Tensor] = []
    # eval mode + no autograd: every conv → BatchNorm (→ add → ReLU) of the K pipelines is ONE launch with the
    # normalisation folded into the weights (os_cnn._FoldedBN); the packed weight images are shared by all batches
    with ops.pack_cache():
        for x, y in batches:
            x = x.float().to(dev)
            labels.append(y.to(dev))
            for k, (fe, clf) in enumerate(models):
                per_model[k].append(clf(fe(x))[0])
    return torch.stack([torch.cat(v) for v in per_model]), torch.cat(labels)


def multi_source_voting(models, train_batches, test_batches):
    """The whole script block: weights from the train set, vote on the test set."""
    tr_logits, tr_labels = collect_logits(models, train_batches)
    te_logits, te_labels = collect_logits(models, test_batches)
    return multi_source_vote(tr_logits, tr_labels, te_logits, te_labels)
