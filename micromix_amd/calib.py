"""Calibration side inputs of the hot path: the on-disk format the reference's
`reorder_indices.py` writes and `model/main.py` reads, and the rule that produces it.

Files (reorder_indices.py:149-151, main.py:114-124), all `torch.save`d dicts keyed by
`layers.{i}.self_attn.{q,k,v,o}_proj.input` / `layers.{i}.mlp.{gate,up,down}_proj.input`
(Mixtral: `layers.{i}.block_sparse_moe.experts.{j}.{w1,w2,w3}.input`):
    saved/{model}_reorder_index_wikitext2.pt   key -> LongTensor[K]  (ascending argsort of the channel score)
    saved/{model}_p6_num_wikitext2.pt          key -> int (multiple of 128)
    saved/{model}_p8_num_wikitext2.pt          key -> int (multiple of 128)

`split_from_activations` restates reorder_indices.py:41-49,64-69,98-111 so that the same side inputs can be produced
from any activation sample (the reference needs HF weights + wikitext2, neither available offline)."""
from __future__ import annotations

import math
import os
from typing import Dict, Tuple

import torch

SUFFIXES = ("reorder_index", "p6_num", "p8_num")


def file_names(model_name: str, dataset: str = "wikitext2") -> Tuple[str, str, str]:
    return tuple(f"{model_name}_{s}_{dataset}.pt" for s in SUFFIXES)


def load_calibration(directory: str, model_name: str, dataset: str = "wikitext2"):
    """-> (reorder_index, p6_nums, p8_nums) exactly as main.py:121-123 loads them."""
    paths = [os.path.join(directory, f) for f in file_names(model_name, dataset)]
    if not os.path.isfile(paths[0]):
        raise FileNotFoundError("reorder index file not found.")          # main.py:118
    reorder_index = torch.load(paths[0], weights_only=False)
    p6 = torch.load(paths[1], weights_only=False)
    p8 = torch.load(paths[2], weights_only=False)
    validate(reorder_index, p6, p8)
    return reorder_index, p6, p8


def save_calibration(directory: str, model_name: str, reorder_index: Dict[str, torch.Tensor], p6: Dict[str, int],
                     p8: Dict[str, int], dataset: str = "wikitext2"):
    validate(reorder_index, p6, p8)
    os.makedirs(directory, exist_ok=True)
    for f, obj in zip(file_names(model_name, dataset), (reorder_index, p6, p8)):
        torch.save(obj, os.path.join(directory, f))


def validate(reorder_index, p6, p8):
    for key, idx in reorder_index.items():
        k = idx.numel()
        if key not in p6 or key not in p8:
            raise KeyError(f"{key}: p6_num / p8_num entry missing")
        a, b = int(p6[key]), int(p8[key])
        if a % 128 or b % 128 or a < 0 or b < 0 or a + b > k or (k - a - b) % 128:
            raise ValueError(f"{key}: p6_num={a}, p8_num={b} must be multiples of 128 with p4 = K - p6 - p8 >= 0 (K={k})")
        if k > 32768:
            raise ValueError(f"{key}: K={k} does not fit the int16 reorder index of the kernels")
        if not torch.equal(torch.sort(idx.long()).values, torch.arange(k)):
            raise ValueError(f"{key}: reorder_index is not a permutation of range({k})")


def split_from_activations(x: torch.Tensor, lamda: float = 1.0):
    """x: [tokens, K] activations feeding one linear layer.  Returns (reorder_index LongTensor[K], p4, p6, p8).

    reorder_indices.py: channel score = mean |x| over tokens (:44, max over batches :47), ascending argsort (:66);
    per-token thresholds p4_thr = rowmax * 448/6/2^10 * lamda, p6_thr = rowmax * 448/28/2^6 * lamda (:103-104);
    ratios of elements below them (:106-108); p6/p8 rounded UP to multiples of 128 (:109-110)."""
    v = x.reshape(-1, x.shape[-1]).float().abs()
    k = v.shape[-1]
    order = torch.sort(v.mean(dim=0), descending=False).indices
    rowmax = v.max(dim=-1, keepdim=True)[0]
    p4_thr = rowmax * 448 / 6 / math.pow(2, 10) * lamda
    p6_thr = rowmax * 448 / 28 / math.pow(2, 6) * lamda
    p4_ratio = float((v < p4_thr).sum()) / v.numel()
    p6_ratio = float((v < p6_thr).sum()) / v.numel() - p4_ratio
    p8_ratio = 1 - p4_ratio - p6_ratio
    p6 = math.ceil(k * p6_ratio / 128) * 128
    p8 = math.ceil(k * p8_ratio / 128) * 128
    p6 = max(0, min(p6, k))
    p8 = max(0, min(p8, k - p6))
    return order, k - p6 - p8, p6, p8


def llama_keys(num_layers: int):
    """the dict keys QLlamaAttention / QLlamaMLP look up (qLlamaLayer.py:212-235, 336-355)."""
    for i in range(num_layers):
        for blk, projs in (("self_attn", ("q_proj", "k_proj", "v_proj", "o_proj")), ("mlp", ("gate_proj", "up_proj", "down_proj"))):
            for p in projs:
                yield f"layers.{i}.{blk}.{p}.input"
