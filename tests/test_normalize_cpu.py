"""Known-answer tests of the normalisation layer.  The first three restate the reference's own tests
(src/openpi/shared/normalize_test.py:6-43 — the only tests near this path that pin values); the rest pin the
formulas of src/openpi/transforms.py:124-191 and the on-disk layout of normalize.py:125-146."""

import json

import numpy as np
import pytest

from kai0_amd import normalize as N


def test_running_stats_one_vector_at_a_time():  # normalize_test.py:6-16
    arr = np.arange(12).reshape(4, 3)
    rs = N.RunningStats()
    for i in range(len(arr)):
        rs.update(arr[i : i + 1])
    st = rs.get_statistics()
    assert np.allclose(st.mean, arr.mean(0)) and np.allclose(st.std, arr.std(0))


def test_json_round_trip():  # normalize_test.py:19-27
    rs = N.RunningStats()
    rs.update(np.arange(12).reshape(4, 3))
    a = {"test": rs.get_statistics()}
    b = N.deserialize_json(N.serialize_json(a))
    assert np.allclose(a["test"].mean, b["test"].mean) and np.allclose(a["test"].std, b["test"].std)
    assert np.allclose(a["test"].q01, b["test"].q01) and np.allclose(a["test"].q99, b["test"].q99)


def test_multiple_batch_dimensions():  # normalize_test.py:30-43
    arr = np.random.default_rng(0).random((2, 3, 4))
    rs = N.RunningStats()
    rs.update(arr)
    st = rs.get_statistics()
    flat = arr.reshape(-1, 4)
    assert np.allclose(st.mean, flat.mean(0)) and np.allclose(st.std, flat.std(0))


def test_quantiles_track_numpy_after_rebinning():
    rng = np.random.default_rng(1)
    data = rng.normal(size=(20000, 3)) * np.array([1.0, 5.0, 0.1]) + np.array([0.0, 2.0, -1.0])
    rs = N.RunningStats()
    for chunk in np.array_split(data, 17):  # growing range -> histograms are re-binned several times
        rs.update(chunk)
    st = rs.get_statistics()
    span = data.max(0) - data.min(0)
    assert np.all(np.abs(st.q01 - np.quantile(data, 0.01, axis=0)) < 0.01 * span)
    assert np.all(np.abs(st.q99 - np.quantile(data, 0.99, axis=0)) < 0.01 * span)


def test_errors():
    rs = N.RunningStats()
    rs.update(np.zeros((1, 3)))
    with pytest.raises(ValueError, match="less than 2"):
        rs.get_statistics()
    with pytest.raises(ValueError, match="does not match"):
        rs.update(np.zeros((2, 4)))


def test_file_layout_is_the_reference_layout(tmp_path):
    st = {"state": N.NormStats(mean=np.array([1.0, 2.0]), std=np.array([3.0, 4.0])),
          "actions": N.NormStats(mean=np.zeros(2), std=np.ones(2), q01=-np.ones(2), q99=np.ones(2))}  # fmt: skip
    N.save(tmp_path, st)
    raw = json.loads((tmp_path / "norm_stats.json").read_text())
    assert set(raw) == {"norm_stats"} and raw["norm_stats"]["state"] == {"mean": [1.0, 2.0], "std": [3.0, 4.0], "q01": None, "q99": None}
    back = N.load(tmp_path)
    assert back["state"].q01 is None and np.array_equal(back["actions"].q99, np.ones(2))
    with pytest.raises(FileNotFoundError):
        N.load(tmp_path / "missing")


def test_zscore_maps_and_dimension_rules():
    st = N.NormStats(mean=np.array([1.0, 2.0, 3.0]), std=np.array([2.0, 4.0, 8.0]))
    x = np.array([[3.0, 6.0]])  # vector shorter than the stats: stats are truncated
    assert np.allclose(N.normalize(x, st), (x - st.mean[:2]) / (st.std[:2] + 1e-6))
    y = np.array([[0.5, -0.5, 1.0, 7.0]])  # vector longer than the stats: mean 0 / std 1 for the tail
    out = N.unnormalize(y, st)
    assert np.allclose(out[0, :3], y[0, :3] * (st.std + 1e-6) + st.mean) and np.isclose(out[0, 3], 7.0 * (1 + 1e-6))
    z = np.array([[0.1, 0.2, 0.3]])
    assert np.allclose(N.unnormalize(N.normalize(z, st), st), z)


def test_quantile_maps_and_dimension_rules():
    st = N.NormStats(mean=np.zeros(2), std=np.ones(2), q01=np.array([-2.0, 0.0]), q99=np.array([2.0, 10.0]))
    x = np.array([[-2.0, 10.0], [0.0, 5.0]])
    n = N.normalize(x, st, use_quantiles=True)
    assert np.allclose(n, [[-1.0, 1.0], [0.0, 0.0]], atol=1e-5)
    assert np.allclose(N.unnormalize(n, st, use_quantiles=True), x, atol=1e-5)
    y = np.array([[1.0, -1.0, 42.0]])  # trailing dimension without stats passes through
    assert np.allclose(N.unnormalize(y, st, use_quantiles=True), [[2.0 + 2e-6, 0.0, 42.0]], atol=1e-4)
    with pytest.raises(ValueError, match="quantile stats"):
        N.normalize(x, N.NormStats(mean=np.zeros(2), std=np.ones(2)), use_quantiles=True)


def test_running_stats_against_vectors_generated_by_the_reference_itself():
    """tests/golden/normalize_golden.json was produced by running the reference's RunningStats in the build container
    (tests/golden/make_normalize_golden.py): same data, same chunking -> same mean / std / quantiles."""
    import os

    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "normalize_golden.json")))
    assert len(g["cases"]) == 3
    for c in g["cases"]:
        data = np.asarray(c["data"])
        rs = N.RunningStats()
        for part in np.array_split(data, c["chunks"], axis=0):
            rs.update(part)
        st = rs.get_statistics()
        for f in ("mean", "std", "q01", "q99"):
            assert np.allclose(getattr(st, f), np.asarray(c[f]), rtol=1e-12, atol=1e-12), (c["name"], f)
