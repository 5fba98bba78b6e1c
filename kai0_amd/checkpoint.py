"""safetensors I/O with the reference's conventions (train_pytorch.py:149-194, model.py:276-280).

`safetensors.torch.save_model/load_model` work on a plain model (and stay the recommended call for callers such as
model_arithmetic), but they refuse modules whose parameters are views into shared flat buffers — which is how the
sharded trainer stores them.  These two helpers produce/consume the SAME file contents: tied weights are written
once under the alphabetically first name (what safetensors' `_remove_duplicate_names` keeps), dtypes untouched."""

from __future__ import annotations

import torch
from safetensors.torch import load_file, save_file


def _tied_groups(model: torch.nn.Module):
    by_id: dict[int, list[str]] = {}
    for name, p in model.state_dict(keep_vars=True).items():
        by_id.setdefault(id(p), []).append(name)
    return [sorted(names) for names in by_id.values()]


def save_model_safetensors(model: torch.nn.Module, path: str) -> None:
    sd = model.state_dict()
    out, meta = {}, {"format": "pt"}
    for names in _tied_groups(model):
        keep = names[0]
        out[keep] = sd[keep].detach().clone().contiguous()
        for alias in names[1:]:
            meta[alias] = keep  # same convention as safetensors.torch.save_model
    save_file(out, path, metadata=meta)


def load_model_safetensors(model: torch.nn.Module, path: str, strict: bool = True) -> None:
    sd = load_file(path)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    # a missing name is forgiven only if it is tied to a tensor the file DID contain (what safetensors.torch.load_model
    # forgives: the aliases save_model dropped); a file with no member of a tied group leaves that weight unloaded -> error
    forgiven = {n for names in _tied_groups(model) if any(m in sd for m in names) for n in names}
    missing = [k for k in missing if k not in forgiven]
    if strict and (missing or unexpected):
        raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {list(unexpected)}")
