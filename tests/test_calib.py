"""CPU tests of the calibration side-input format (reorder_indices.py / main.py contract)."""
import os

import pytest
import torch

from micromix_amd import calib


def test_split_rule_and_roundtrip(tmp_path):
    g = torch.Generator().manual_seed(0)
    x = torch.randn((512, 1024), generator=g)
    x[:, :13] *= 40          # a few outlier channels
    order, p4, p6, p8 = calib.split_from_activations(x, lamda=1.0)
    assert p4 + p6 + p8 == 1024 and p6 % 128 == 0 and p8 % 128 == 0 and p4 % 128 == 0
    assert sorted(order.tolist()) == list(range(1024))
    # ascending mean |x|: the 13 outlier channels end up last, i.e. in the highest-precision segment
    assert set(order[-13:].tolist()) == set(range(13)) and p8 >= 128
    keys = list(calib.llama_keys(2))
    assert keys[0] == "layers.0.self_attn.q_proj.input" and len(keys) == 14
    ri = {k: order for k in keys}
    p6s = {k: p6 for k in keys}
    p8s = {k: p8 for k in keys}
    calib.save_calibration(str(tmp_path), "Llama-3-8B", ri, p6s, p8s)
    assert sorted(os.listdir(tmp_path)) == sorted(calib.file_names("Llama-3-8B"))
    ri2, p6b, p8b = calib.load_calibration(str(tmp_path), "Llama-3-8B")
    assert torch.equal(ri2[keys[3]], order) and p6b[keys[5]] == p6 and p8b[keys[6]] == p8
    with pytest.raises(FileNotFoundError, match="reorder index file not found"):
        calib.load_calibration(str(tmp_path), "missing-model")


def test_validation():
    idx = torch.arange(256)
    with pytest.raises(ValueError):
        calib.validate({"a": idx}, {"a": 100}, {"a": 128})
    with pytest.raises(ValueError):
        calib.validate({"a": torch.zeros(256, dtype=torch.long)}, {"a": 128}, {"a": 128})
    with pytest.raises(KeyError):
        calib.validate({"a": idx}, {}, {"a": 128})
    calib.validate({"a": idx}, {"a": 0}, {"a": 256})
