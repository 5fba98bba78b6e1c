"""State / action normalisation on either side of the hot path: the statistics file a checkpoint
ships (`assets/<asset_id>/norm_stats.json`) and the two affine maps `Policy.infer` applies around `sample_actions`.

Mirrors the reference's behaviour — not its code — so files written by either side load in the other:
  * statistics container + running estimator: `src/openpi/shared/normalize.py:9-122` (mean / std from running first and
    second moments; 1 % / 99 % quantiles from 5000-bin per-dimension histograms that are re-binned when the range grows);
  * JSON layout `{"norm_stats": {key: {"mean": [...], "std": [...], "q01": [...]|null, "q99": [...]|null}}}`:
    `normalize.py:125-146`;
  * normalise / unnormalise formulas incl. the 1e-6 guards and the "stats shorter than the vector" rules:
    `src/openpi/transforms.py:124-191`.
"""

from __future__ import annotations

import dataclasses
import json
import pathlib

import numpy as np

EPS = 1e-6
QUANTILE_BINS = 5000


@dataclasses.dataclass
class NormStats:
    mean: np.ndarray
    std: np.ndarray
    q01: np.ndarray | None = None
    q99: np.ndarray | None = None

    def __post_init__(self):
        for f in ("mean", "std", "q01", "q99"):
            v = getattr(self, f)
            if v is not None:
                setattr(self, f, np.asarray(v))


class RunningStats:
    """Streaming mean / std / 1-99 % quantiles over vectors (all leading axes are batch axes)."""

    def __init__(self, bins: int = QUANTILE_BINS):
        self.bins = bins
        self.count = 0
        self.m1 = self.m2 = self.lo = self.hi = None
        self.hist: list[np.ndarray] = []
        self.edges: list[np.ndarray] = []

    def update(self, batch: np.ndarray) -> None:
        x = np.asarray(batch).reshape(-1, np.shape(batch)[-1])
        n, d = x.shape
        if self.count == 0:
            self.m1, self.m2 = x.mean(0), (x**2).mean(0)
            self.lo, self.hi = x.min(0), x.max(0)
            self.hist = [np.zeros(self.bins) for _ in range(d)]
            self.edges = [np.linspace(self.lo[i] - 1e-10, self.hi[i] + 1e-10, self.bins + 1) for i in range(d)]
        else:
            if d != self.m1.size:
                raise ValueError("The length of new vectors does not match the initialized vector length.")
            lo, hi = x.min(0), x.max(0)
            grew = bool(np.any(hi > self.hi) or np.any(lo < self.lo))
            self.lo, self.hi = np.minimum(self.lo, lo), np.maximum(self.hi, hi)
            if grew:  # re-bin what has been counted so far onto the wider range
                for i in range(d):
                    new_edges = np.linspace(self.lo[i], self.hi[i], self.bins + 1)
                    self.hist[i], _ = np.histogram(self.edges[i][:-1], bins=new_edges, weights=self.hist[i])
                    self.edges[i] = new_edges
        self.count += n
        w = n / self.count
        self.m1 = self.m1 + (x.mean(0) - self.m1) * w
        self.m2 = self.m2 + ((x**2).mean(0) - self.m2) * w
        for i in range(d):
            h, _ = np.histogram(x[:, i], bins=self.edges[i])
            self.hist[i] = self.hist[i] + h

    def _quantile(self, q: float) -> np.ndarray:
        target = q * self.count
        return np.array([e[np.searchsorted(np.cumsum(h), target)] for h, e in zip(self.hist, self.edges, strict=True)])

    def get_statistics(self) -> NormStats:
        if self.count < 2:
            raise ValueError("Cannot compute statistics for less than 2 vectors.")
        std = np.sqrt(np.maximum(0, self.m2 - self.m1**2))
        return NormStats(mean=self.m1, std=std, q01=self._quantile(0.01), q99=self._quantile(0.99))


# ------------------------------------------------------------------------------------------------ file format
def serialize_json(norm_stats: dict[str, NormStats]) -> str:
    def enc(v):
        return None if v is None else np.asarray(v).tolist()

    body = {k: {"mean": enc(s.mean), "std": enc(s.std), "q01": enc(s.q01), "q99": enc(s.q99)} for k, s in norm_stats.items()}
    return json.dumps({"norm_stats": body}, indent=2)


def deserialize_json(data: str) -> dict[str, NormStats]:
    body = json.loads(data)["norm_stats"]
    return {k: NormStats(**{f: (None if v.get(f) is None else np.asarray(v[f])) for f in ("mean", "std", "q01", "q99")})
            for k, v in body.items()}  # fmt: skip


def save(directory, norm_stats: dict[str, NormStats]) -> None:
    path = pathlib.Path(directory) / "norm_stats.json"
    path.parent.mkdir(parents=True, exist_ok=True)
    path.write_text(serialize_json(norm_stats))


def load(directory) -> dict[str, NormStats]:
    path = pathlib.Path(directory) / "norm_stats.json"
    if not path.exists():
        raise FileNotFoundError(f"Norm stats file not found at: {path}")
    return deserialize_json(path.read_text())


# ------------------------------------------------------------------------------------------------ the two maps
def _pad_last(v: np.ndarray, n: int, value: float) -> np.ndarray:
    if v.shape[-1] >= n:
        return v
    pad = [(0, 0)] * (v.ndim - 1) + [(0, n - v.shape[-1])]
    return np.pad(v, pad, constant_values=value)


def normalize(x: np.ndarray, stats: NormStats, use_quantiles: bool = False) -> np.ndarray:
    """Input side.  Stats longer than the vector are truncated to it."""
    d = x.shape[-1]
    if use_quantiles:
        if stats.q01 is None or stats.q99 is None:
            raise ValueError("quantile stats must be provided if use_quantile_norm is True")
        q01, q99 = stats.q01[..., :d], stats.q99[..., :d]
        return (x - q01) / (q99 - q01 + EPS) * 2.0 - 1.0
    return (x - stats.mean[..., :d]) / (stats.std[..., :d] + EPS)


def unnormalize(x: np.ndarray, stats: NormStats, use_quantiles: bool = False) -> np.ndarray:
    """Output side.  z-score: stats shorter than the vector are padded with mean 0 / std 1; quantiles: the extra trailing
    dimensions pass through unchanged."""
    d = x.shape[-1]
    if use_quantiles:
        if stats.q01 is None or stats.q99 is None:
            raise ValueError("quantile stats must be provided if use_quantile_norm is True")
        k = stats.q01.shape[-1]
        if k < d:
            head = (x[..., :k] + 1.0) / 2.0 * (stats.q99 - stats.q01 + EPS) + stats.q01
            return np.concatenate([head, x[..., k:]], axis=-1)
        return (x + 1.0) / 2.0 * (stats.q99 - stats.q01 + EPS) + stats.q01
    return x * (_pad_last(stats.std, d, 1.0) + EPS) + _pad_last(stats.mean, d, 0.0)
