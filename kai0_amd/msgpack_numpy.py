"""The serving wire format for arrays (SURVEY.md §8b "Serving wire"): msgpack with numpy arrays and scalars carried as small
tagged maps, byte-compatible with `packages/openpi-client/src/openpi_client/msgpack_numpy.py:21-57` so that existing robot
clients and this server can talk to each other:

    ndarray -> {b"__ndarray__": True, b"data": <raw bytes, C order>, b"dtype": dtype.str, b"shape": shape}
    scalar  -> {b"__npgeneric__": True, b"data": <python value>, b"dtype": dtype.str}

Void / object / complex dtypes are refused (nothing is ever pickled)."""

from __future__ import annotations

import functools

import msgpack
import numpy as np

_REFUSED_KINDS = ("V", "O", "c")


def pack_array(obj):
    is_np = isinstance(obj, (np.ndarray, np.generic))
    if is_np and obj.dtype.kind in _REFUSED_KINDS:
        raise ValueError(f"Unsupported dtype: {obj.dtype}")
    if isinstance(obj, np.ndarray):
        return {b"__ndarray__": True, b"data": obj.tobytes(), b"dtype": obj.dtype.str, b"shape": obj.shape}
    if isinstance(obj, np.generic):
        return {b"__npgeneric__": True, b"data": obj.item(), b"dtype": obj.dtype.str}
    return obj


def unpack_array(obj):
    if b"__ndarray__" in obj:
        return np.ndarray(buffer=obj[b"data"], dtype=np.dtype(obj[b"dtype"]), shape=obj[b"shape"])
    if b"__npgeneric__" in obj:
        return np.dtype(obj[b"dtype"]).type(obj[b"data"])
    return obj


Packer = functools.partial(msgpack.Packer, default=pack_array)
packb = functools.partial(msgpack.packb, default=pack_array)
Unpacker = functools.partial(msgpack.Unpacker, object_hook=unpack_array)
unpackb = functools.partial(msgpack.unpackb, object_hook=unpack_array)
