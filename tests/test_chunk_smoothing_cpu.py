"""kai0_amd.chunk_smoothing against the deployment scripts' classes EXECUTED from the reference source (lifted with `ast`; the
scripts themselves import ROS and cannot be imported).  Needs /root/reference, which only exists in the build container:
there the random-schedule comparison runs live; everywhere the committed vectors (tests/golden/chunk_smoothing.npz, written
by this file when run as a script) are replayed."""
import ast
import os
import sys
import threading
from collections import deque

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kai0_amd import chunk_smoothing  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/train_deploy_alignment/inference/agilex/inference"
GOLD = os.path.join(HERE, "golden", "chunk_smoothing.npz")


def _lift(path, name):
    ns = {"np": np, "threading": threading, "deque": deque}
    for node in ast.parse(open(path).read()).body:
        if isinstance(node, ast.ClassDef) and node.name == name:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
            return ns[name]
    raise KeyError(name)


def _schedule(seed):
    """A random interleaving of chunk arrivals and control ticks, as (op, payload) pairs."""
    rng = np.random.default_rng(seed)
    ops = []
    for _ in range(60):
        if rng.random() < 0.3:
            n = int(rng.integers(1, 30))
            ops.append(("chunk", rng.standard_normal((n, 14)), int(rng.integers(0, 12)), int(rng.integers(1, 12)), int(rng.integers(-3, 4))))
        for _ in range(int(rng.integers(0, 9))):
            ops.append(("tick",))
    return ops


def _run_stream(buf, ops):
    out = []
    for op in ops:
        if op[0] == "chunk":
            buf.integrate_new_chunk(op[1], max_k=op[2], min_m=op[3])
        else:
            a = buf.pop_next_action()
            out.append(np.full(14, np.nan) if a is None else a)
    return np.array(out)


def _run_ensemble(buf, ops):
    out = []
    for op in ops:
        if op[0] == "chunk":
            buf.add_chunk(op[1], start_timestep=None if op[4] == 0 else buf.get_current_timestep() + op[4])
        else:
            a = buf.pop_next_action()
            out.append(np.full(14, np.nan) if a is None else a)
    return np.array(out)


def _ours(seed):
    ops = _schedule(seed)
    return (_run_stream(chunk_smoothing.StreamActionBuffer(), ops),
            _run_ensemble(chunk_smoothing.TemporalEnsemblingBuffer(exp_weight_m=0.05), ops))


@pytest.mark.parametrize("seed", range(6))
def test_against_committed_reference_vectors(seed):
    g = np.load(GOLD)
    s, e = _ours(seed)
    assert np.array_equal(s, g[f"stream.{seed}"], equal_nan=True) and np.array_equal(e, g[f"ensemble.{seed}"], equal_nan=True)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference sources only exist in the build container")
@pytest.mark.parametrize("seed", range(100, 110))
def test_live_against_the_reference_classes(seed):
    ops = _schedule(seed)
    ref_s = _lift(f"{REF}/agilex_inference_openpi_temporal_smoothing.py", "StreamActionBuffer")()
    ref_e = _lift(f"{REF}/agilex_inference_openpi_temporal_ensembling.py", "TemporalEnsemblingBuffer")(exp_weight_m=0.05)
    s, e = _ours(seed)
    assert np.array_equal(s, _run_stream(ref_s, ops), equal_nan=True)
    assert np.array_equal(e, _run_ensemble(ref_e, ops), equal_nan=True)


def test_documented_behaviour():
    b = chunk_smoothing.StreamActionBuffer()
    assert b.pop_next_action() is None and not b.has_any()
    b.integrate_new_chunk(np.arange(10, dtype=float)[:, None] * np.ones(14), max_k=5)
    assert b.has_any() and b.pop_next_action()[0] == 0.0 and b.k == 1
    b.integrate_new_chunk(np.full((10, 14), 100.0), max_k=5, min_m=1)  # first row of the new chunk dropped (k = 1); linear cross-fade
    first = b.pop_next_action()
    assert first[0] == 1.0  # overlap starts at 100 % old plan
    e = chunk_smoothing.TemporalEnsemblingBuffer(exp_weight_m=0.0)
    e.add_chunk(np.zeros((5, 14)))
    e.add_chunk(np.ones((5, 14)))
    assert np.allclose(e.pop_next_action(), 0.5) and e.get_current_timestep() == 1 and e.has_prediction()
    e.reset()
    assert e.pop_next_action() is None


if __name__ == "__main__":  # build container: write the committed vectors from the reference classes
    out = {}
    for seed in range(6):
        ops = _schedule(seed)
        out[f"stream.{seed}"] = _run_stream(_lift(f"{REF}/agilex_inference_openpi_temporal_smoothing.py", "StreamActionBuffer")(), ops)
        out[f"ensemble.{seed}"] = _run_ensemble(
            _lift(f"{REF}/agilex_inference_openpi_temporal_ensembling.py", "TemporalEnsemblingBuffer")(exp_weight_m=0.05), ops)
    np.savez_compressed(GOLD, **out)
    print("wrote", GOLD, {k: v.shape for k, v in out.items()})
