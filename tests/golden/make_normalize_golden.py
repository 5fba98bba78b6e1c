"""Generate tests/golden/normalize_golden.json by RUNNING THE REFERENCE (src/openpi/shared/normalize.py) in the build
container — it cannot travel to the GPU box, the vectors can.  `numpydantic` is not installed here; the reference only
uses it as a field annotation, so a one-line stand-in module is registered before the import.

    python tests/golden/make_normalize_golden.py   # needs /root/reference
"""
import importlib.util
import json
import os
import sys
import types
import typing

import numpy as np

REF = "/root/reference/src/openpi/shared/normalize.py"
stub = types.ModuleType("numpydantic")
stub.NDArray = typing.Any
sys.modules.setdefault("numpydantic", stub)
spec = importlib.util.spec_from_file_location("ref_normalize", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(20240925)
cases = []
for name, shape, scale, chunks in (("state14", (600, 14), 3.0, 9), ("actions", (10, 30, 7), 0.5, 5), ("tiny", (4, 3), 1.0, 4)):
    data = rng.normal(size=shape) * scale + rng.normal(size=shape[-1])
    if name == "tiny":
        data = np.arange(12, dtype=np.float64).reshape(4, 3)
    rs = ref.RunningStats()
    parts = np.array_split(data, chunks, axis=0)
    for p in parts:
        rs.update(p)
    st = rs.get_statistics()
    cases.append({"name": name, "data": data.tolist(), "chunks": chunks,
                  "mean": np.asarray(st.mean).tolist(), "std": np.asarray(st.std).tolist(),
                  "q01": np.asarray(st.q01).tolist(), "q99": np.asarray(st.q99).tolist()})
# (serialize_json cannot be run here: it relies on numpydantic's ndarray -> list serialiser, which the stand-in lacks; the
# JSON layout is pinned from normalize.py:125-146 in tests/test_normalize_cpu.py instead)
out = {"cases": cases}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "normalize_golden.json")
json.dump(out, open(path, "w"))
print("wrote", path, os.path.getsize(path), "bytes")
