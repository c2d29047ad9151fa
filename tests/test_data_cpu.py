"""Input pipeline (DataSource.py): the .ts parser on hand-written files, and TrainData / TestData against the fixture
the REFERENCE's classes produced on top of this parser (shared label dictionary, num_class quirks, unseen test label)."""
import json
import os

import numpy as np
import pytest
import torch

import feature_level_style_transfer_for_tsc_amd as fst
from feature_level_style_transfer_for_tsc_amd.data import TsFormatError

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

UNI = """# comment
@problemName uni
@timeStamps false
@missing true
@univariate true
@equalLength true
@seriesLength 5
@classLabel true 1 2

@data
0.5,1,-2e-1,?,3:1
1,2,3,4,5:2
"""


def test_parse_univariate_with_missing_values():
    x, y = fst.parse_ts(UNI)
    assert x.shape == (2, 1, 5) and x.dtype == np.float64 and list(y) == ["1", "2"]
    assert np.isnan(x[0, 0, 3]) and x[0, 0, 2] == -0.2 and x[1, 0, 4] == 5.0


@pytest.mark.parametrize("bad,msg", [
    ("@classLabel true a\n@data\n1,2:3,4:a\n1,2:a\n", "not 2 x 2"),
    ("@data\n1,2:a\n", "@data before @classLabel"),
    ("@classLabel true a\n@data\n1,x:a\n", "could not convert"),
    ("@classLabel true a\n@timeStamps true\n@data\n1,2:a\n", "time-stamped"),
    ("@classLabel true a\n", "no @data"),
])
def test_malformed_files_are_rejected(bad, msg):
    with pytest.raises(TsFormatError, match=msg):
        fst.parse_ts(bad)


def test_dataset_classes_match_the_reference(tmp_path, capsys):
    g = dict(np.load(os.path.join(GOLDEN, "datasource_small.npz"), allow_pickle=False))
    for k in ("target_TRAIN", "target_TEST", "source_TRAIN"):
        (tmp_path / f"{k}.ts").write_text(str(g[f"text.{k}"]))
    d = {}
    a = fst.TrainData(str(tmp_path), "target_TRAIN.ts", d)
    assert d == json.loads(str(g["dict_after_target_train"])) and list(d) == ["run", "walk", "sit"]   # first-appearance order
    b = fst.TestData(str(tmp_path), "target_TEST.ts", d)
    assert "jump" in capsys.readouterr().out and b.unseen_labels == ["jump"]
    c = fst.TrainData(str(tmp_path), "source_TRAIN.ts", d)
    assert d == json.loads(str(g["dict_final"]))
    for name, obj, xk, yk in (("target_TRAIN", a, "train_x", "train_y"), ("target_TEST", b, "test_x", "test_y"),
                              ("source_TRAIN", c, "train_x", "train_y")):
        x, y = getattr(obj, xk), getattr(obj, yk)
        assert x.dtype == torch.float64 and y.dtype == torch.int64
        assert np.array_equal(x.numpy(), g[f"{name}.x"]) and np.array_equal(y.numpy(), g[f"{name}.y"])
        assert [obj.len, obj.in_channel, obj.time_length, obj.num_class] == list(g[f"{name}.meta"])
        assert np.array_equal(obj[1][0].numpy(), g[f"{name}.item1_x"]) and int(obj[1][1]) == int(g[f"{name}.item1_y"])
    assert a.num_class == 3 and b.num_class == 0 and c.num_class == 1                  # the reference's quirks
    assert len(b.test_y) == len(b) - 1                                                  # the unseen sample got no label


def test_device_loader_on_cpu_covers_every_sample_once():
    x = torch.arange(10 * 2 * 3, dtype=torch.float64).view(10, 2, 3)
    y = torch.arange(10)
    ld = fst.DeviceLoader(x, y, 4, "cpu")
    got = list(ld)
    assert len(ld) == 3 and [b[0].shape[0] for b in got] == [4, 4, 2] and got[0][0].dtype == torch.float32
    assert torch.equal(torch.cat([b[1] for b in got]), y) and torch.equal(torch.cat([b[0] for b in got]), x.float())
    sh = fst.DeviceLoader(x, y, 4, "cpu", generator=torch.Generator().manual_seed(0), drop_last=True)
    ys = torch.cat([b[1] for b in sh])
    assert len(sh) == 2 and len(ys) == 8 and len(set(ys.tolist())) == 8
    for xb, yb in sh:
        assert torch.equal(xb, x.float()[yb])
