"""Executing the reference's own definitions at TEST time (build container only).

The reference package cannot be imported here (python 3.10, no flax / jax, transformers 5.x instead of the patched 4.53.2),
but the functions and classes the hot path is made of are plain torch: `lift` / `lift_method` take exactly those definitions
out of the reference source with `ast` and compile them into a namespace.  Nothing is copied into the repository — the source is
read from /root/reference when a test runs, and tests that need it skip where it is absent (the GPU box).  The committed
fixtures under tests/golden/ come from the same mechanism (tests/golden/make_reference_*_golden.py)."""

import ast
import math
import os
import typing

import torch
import torch.nn.functional as F  # noqa: N812
import typing_extensions
from torch import nn

REF = "/root/reference/src/openpi/models_pytorch"
GEMMA_PY = f"{REF}/transformers_replace/models/gemma/modeling_gemma.py"
SIGLIP_PY = f"{REF}/transformers_replace/models/siglip/modeling_siglip.py"
PI0_PY = f"{REF}/pi0_pytorch.py"
GEMMA_PT_PY = f"{REF}/gemma_pytorch.py"


def available() -> bool:
    return all(os.path.isfile(p) for p in (GEMMA_PY, SIGLIP_PY, PI0_PY, GEMMA_PT_PY))


def lift(path, names, ns):
    tree = ast.parse(open(path).read())
    found = set()
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
            found.add(node.name)
    assert found == set(names), set(names) - found
    return ns


def lift_method(path, cls, name, ns):
    """a method of a class, compiled as a free function taking `self` explicitly"""
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == name:
                    exec(compile(ast.Module([sub], []), path, "exec"), ns)
                    return ns[name]
    raise KeyError((cls, name))


def base_ns():
    from transformers.activations import ACT2FN  # un-vendored dependency: "gelu_pytorch_tanh" = F.gelu(approximate="tanh")

    return {"torch": torch, "nn": nn, "F": F, "math": math, "Optional": typing.Optional, "Union": typing.Union,
            "Callable": typing.Callable, "Tensor": torch.Tensor, "Cache": typing.Any, "GemmaConfig": typing.Any,
            "SiglipVisionConfig": typing.Any, "SiglipTextConfig": typing.Any, "FlashAttentionKwargs": dict,
            "Unpack": typing_extensions.Unpack, "GradientCheckpointingLayer": nn.Module, "ACT2FN": ACT2FN}  # fmt: skip
