"""TensorDict — nested dict of tensors that can be indexed/assigned like a tensor.

Same contract as sample_factory/algo/utils/tensor_dict.py:19-70: string keys address the dict, anything else
indexes every leaf; assignment with a non-string key copies into every matching leaf.
"""
from __future__ import annotations

import torch


class TensorDict(dict):
    def __getitem__(self, key):
        if isinstance(key, str):
            return dict.__getitem__(self, key)
        out = TensorDict()
        for k, v in self.items():
            out[k] = v[key]  # leaves: tensor indexing; sub-dicts: recursion through this method
        return out

    def __setitem__(self, key, value):
        if isinstance(key, str):
            dict.__setitem__(self, key, value)
            return
        _assign(self, key, value)


def _assign(dst, index, src):
    if isinstance(src, dict):
        for k, v in src.items():
            _assign(dict.__getitem__(dst, k), index, v)
    else:
        if not torch.is_tensor(src):
            src = torch.as_tensor(src)
        dst[index].copy_(src)


def clone_tensordict(d: TensorDict) -> TensorDict:
    out = TensorDict()
    for k, v in d.items():
        out[k] = clone_tensordict(v) if isinstance(v, dict) else v.clone().detach()
    return out


def to_device(d: TensorDict, device) -> TensorDict:
    out = TensorDict()
    for k, v in d.items():
        out[k] = to_device(v, device) if isinstance(v, dict) else v.to(device)
    return out
