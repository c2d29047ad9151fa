"""Input pipeline of the reference without sktime (DataSource.py:9-68, train_and_test.py:134-137).

``load_ts`` parses the UCR/UEA ``.ts`` text format into what ``sktime.datasets.load_from_tsfile(...,
return_data_type="numpy3d")`` hands the reference: a float64 array ``[N, C, L]`` and an array of label strings.
``TrainData`` / ``TestData`` keep the reference's constructor signature, attributes and label-dictionary semantics,
including its quirks: the dictionary is shared and mutated by the training set (labels numbered in order of first
appearance), ``num_class`` counts only the labels THIS dataset added, the test set never adds labels (an unseen one is
reported and its sample gets no label) and its ``num_class`` stays 0.

Parity note: the file format itself is restated from its public description (sktime is not in the image, so the
parser has no reference-generated vectors: "parity unpinned" for ``load_ts``); the dataset classes are pinned by
``tests/golden/datasource_small.npz``, produced by the REFERENCE's classes running on top of this parser.

``DeviceLoader`` is the H2D side: batches are cast to float32, staged in pinned memory and copied on a side stream one
batch ahead, so the step never waits for the host at B=256.
"""
from __future__ import annotations

import os
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset


class TsFormatError(ValueError):
    pass


def load_ts(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Parse an equal-length ``.ts`` file -> (x float64 [N, C, L], y str [N]).  ``?`` is a missing value (NaN).
    Dimensions of a case are separated by ``:``, values by ``,``, the class label is the last ``:`` field."""
    with open(path, "r", encoding="utf-8") as f:
        return parse_ts(f.read(), where=path)


def parse_ts(text: str, where: str = "<string>") -> Tuple[np.ndarray, np.ndarray]:
    has_labels, in_data, timestamps = None, False, False
    cases: List[List[np.ndarray]] = []
    labels: List[str] = []
    for ln, raw in enumerate(text.splitlines(), 1):
        line = raw.strip()
        if not line or line.startswith("#"):
            continue
        if not in_data:
            low = line.lower()
            if low.startswith("@data"):
                in_data = True
                if has_labels is None:
                    raise TsFormatError(f"{where}:{ln}: @data before @classLabel")
            elif low.startswith("@classlabel"):
                tok = line.split()
                if len(tok) < 2 or tok[1].lower() not in ("true", "false"):
                    raise TsFormatError(f"{where}:{ln}: malformed @classLabel")
                has_labels = tok[1].lower() == "true"
            elif low.startswith("@timestamps"):
                timestamps = line.split()[-1].lower() == "true"
            elif not low.startswith("@"):
                raise TsFormatError(f"{where}:{ln}: expected a header line starting with '@'")
            continue
        if timestamps:
            raise TsFormatError(f"{where}: time-stamped .ts files are not supported")
        fields = line.split(":")
        if has_labels:
            if len(fields) < 2:
                raise TsFormatError(f"{where}:{ln}: case without a class label")
            labels.append(fields[-1].strip())
            fields = fields[:-1]
        dims = []
        for d in fields:
            vals = [v.strip() for v in d.split(",")]
            try:
                dims.append(np.array([np.nan if v == "?" else float(v) for v in vals], dtype=np.float64))
            except ValueError as e:
                raise TsFormatError(f"{where}:{ln}: {e}") from None
        cases.append(dims)
    if not in_data or not cases:
        raise TsFormatError(f"{where}: no @data section / no cases")
    C, L = len(cases[0]), len(cases[0][0])
    for i, dims in enumerate(cases):
        if len(dims) != C or any(len(v) != L for v in dims):
            raise TsFormatError(f"{where}: case {i} is not {C} x {L} (numpy3d needs equal dimensions and lengths)")
    x = np.stack([np.stack(dims) for dims in cases])
    return x, np.array(labels if has_labels else [""] * len(cases))


class TrainData(Dataset):
    """DataSource.py:9-36 — ``temp_dict`` is filled in place, in order of first appearance."""

    def __init__(self, file_path_begin, file_path_end, temp_dict: Dict[str, int]):
        super().__init__()
        train_x, train_y = load_ts(os.path.join(file_path_begin, file_path_end))
        self.len = train_x.shape[0]
        self.in_channel = train_x.shape[1]
        self.time_length = train_x.shape[-1]
        self.train_x = torch.from_numpy(train_x)                       # float64 like the reference; cast at use (:151)
        label, class_label = [], 0
        for i in train_y:
            if i not in temp_dict:
                temp_dict[i] = class_label
                class_label += 1
            label.append(temp_dict[i])
        self.num_class = class_label                                   # only the labels added here (reference quirk)
        self.train_y = torch.tensor(label).long()

    def __len__(self):
        return self.len

    def __getitem__(self, idx):
        return self.train_x[idx, :, :], self.train_y[idx]


class TestData(Dataset):
    """DataSource.py:38-64 — never adds labels; an unseen one is reported and skipped; ``num_class`` stays 0."""

    def __init__(self, file_path_begin, file_path_end, temp_dict: Dict[str, int]):
        super().__init__()
        test_x, test_y = load_ts(os.path.join(file_path_begin, file_path_end))
        self.len = test_x.shape[0]
        self.in_channel = test_x.shape[1]
        self.time_length = test_x.shape[-1]
        self.test_x = torch.from_numpy(test_x)
        label = []
        self.unseen_labels: List[str] = []
        for i in test_y:
            if i in temp_dict:
                label.append(temp_dict[i])
            else:
                self.unseen_labels.append(str(i))
                print("label of the test set missing from the training set: stop the training", i)
        self.num_class = 0
        self.test_y = torch.tensor(label).long()

    def __len__(self):
        return self.len

    def __getitem__(self, idx):
        return self.test_x[idx, :, :], self.test_y[idx]


class DeviceLoader:
    """Iterate (x float32 on device, y on device) batches with the next batch's host->device copy in flight:
    float32 cast and pinning on the host, ``non_blocking`` copies on a side stream, an event per batch so the consumer
    stream waits only for its own batch.  Order is the dataset's unless ``generator`` is given (then a permutation
    per epoch, like ``DataLoader(shuffle=True)``)."""

    def __init__(self, x: torch.Tensor, y: torch.Tensor, batch_size: int, device, generator: Optional[torch.Generator] = None,
                 drop_last: bool = False):
        if x.size(0) != y.size(0):
            raise ValueError(f"{x.size(0)} series but {y.size(0)} labels")
        self.device = torch.device(device)
        pin = self.device.type == "cuda"
        self.x = x.to(torch.float32).contiguous()
        self.y = y.contiguous()
        if pin:
            self.x, self.y = self.x.pin_memory(), self.y.pin_memory()
        self.batch_size, self.generator, self.drop_last = batch_size, generator, drop_last
        self._copy = torch.cuda.Stream(self.device) if pin else None

    def __len__(self) -> int:
        n = self.x.size(0)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _stage(self, idx):
        if self._copy is None:
            return self.x[idx], self.y[idx], None
        # gather on the host into fresh pinned buffers (index_select output is not pinned), then async copy
        xb = torch.empty((len(idx),) + tuple(self.x.shape[1:]), dtype=torch.float32).pin_memory()
        yb = torch.empty((len(idx),) + tuple(self.y.shape[1:]), dtype=self.y.dtype).pin_memory()
        torch.index_select(self.x, 0, idx, out=xb)
        torch.index_select(self.y, 0, idx, out=yb)
        with torch.cuda.stream(self._copy):
            xd, yd = xb.to(self.device, non_blocking=True), yb.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy)
        return xd, yd, (ev, xb, yb)                                  # keep the pinned buffers alive until consumed

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        n = self.x.size(0)
        order = torch.randperm(n, generator=self.generator) if self.generator is not None else torch.arange(n)
        chunks = [order[i: i + self.batch_size] for i in range(0, n, self.batch_size)]
        if self.drop_last and chunks and len(chunks[-1]) < self.batch_size:
            chunks.pop()
        nxt = self._stage(chunks[0]) if chunks else None
        for i in range(len(chunks)):
            cur = nxt
            nxt = self._stage(chunks[i + 1]) if i + 1 < len(chunks) else None
            xd, yd, hold = cur
            if hold is not None:
                torch.cuda.current_stream(self.device).wait_event(hold[0])
                xd.record_stream(torch.cuda.current_stream(self.device))
                yd.record_stream(torch.cuda.current_stream(self.device))
            yield xd, yd
