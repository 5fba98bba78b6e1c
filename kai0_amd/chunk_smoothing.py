"""Client-side blending of consecutive action chunks — the step right after the hot path in deployment: the
robot loop publishes one action per control tick while new 50-step chunks keep arriving from `Policy.infer`.

Two schemes kai0's deployment scripts use, with the same public methods and numerics, as a plain numpy library (no ROS):

  StreamActionBuffer  (train_deploy_alignment/inference/agilex/inference/agilex_inference_openpi_temporal_smoothing.py:134-259)
    `integrate_new_chunk(chunk, max_k, min_m)`: drop the first min(k, max_k) actions of the new chunk (k = ticks published
    since the last chunk: the part of the new plan that is already in the past), cross-fade it with what is left of the old
    plan — weight of the old plan falls linearly 1 -> 0 over the overlap; an old plan shorter than min_m is first extended by
    repeating its last action (or the last popped action when it ran dry) — and append the new tail.
    `pop_next_action()`: next action, k += 1.

  TemporalEnsemblingBuffer  (…/agilex_inference_openpi_temporal_ensembling.py:55-172; ACT-style)
    every chunk votes for the timesteps it covers; the action at t is the average of all votes with weights exp(-m * i),
    i = 0 for the OLDEST chunk; timesteps without votes repeat the last action; votes older than t - 10 are dropped.

Thread-safety as in the scripts: one lock per buffer (an inference thread adds chunks, the control loop pops).
Pinned by tests/test_chunk_smoothing_cpu.py against the reference classes executed from source on random schedules."""

from __future__ import annotations

import threading

import numpy as np


class StreamActionBuffer:
    def __init__(self, max_chunks: int = 10, decay_alpha: float = 0.25, state_dim: int = 14, smooth_method: str = "temporal"):
        self.max_chunks, self.decay_alpha, self.state_dim, self.smooth_method = max_chunks, float(decay_alpha), state_dim, smooth_method
        self.lock = threading.Lock()
        self._plan = np.zeros((0, state_dim), dtype=float)  # actions still to publish, row 0 next
        self.k = 0                                          # ticks published since the plan was last replaced
        self.last_action = None                             # the action that emptied the plan

    # the reference keeps a deque of per-step arrays; expose the same view for callers that peek at it
    @property
    def cur_chunk(self):
        return list(self._plan)

    def integrate_new_chunk(self, actions_chunk, max_k: int, min_m: int = 8) -> None:
        with self.lock:
            if actions_chunk is None or len(actions_chunk) == 0:
                return
            drop = min(self.k, max(0, int(max_k)))
            min_m = max(1, int(min_m))
            if drop >= len(actions_chunk):
                return  # the whole chunk is already in the past
            new = np.array(actions_chunk[drop:], dtype=None, copy=True)
            old = self._plan
            if len(old) == 0:
                if self.last_action is None:
                    self._plan, self.k = new, 0
                    return
                old = np.repeat(np.asarray(self.last_action, dtype=float)[None], min_m, axis=0)
                self.last_action = None
            elif len(old) < min_m:
                old = np.concatenate([old, np.repeat(old[-1:], min_m - len(old), axis=0)])
            n = min(len(old), len(new))
            w_old = np.linspace(1.0, 0.0, n, dtype=float) if n > 1 else np.ones(1)
            head = w_old[:, None] * old[:n].astype(float) + (1.0 - w_old)[:, None] * new[:n].astype(float)
            self._plan = np.concatenate([head, new[n:]]) if len(new) > n else head
            self.k = 0

    def pop_next_action(self):
        with self.lock:
            if len(self._plan) == 0:
                return None
            act = np.asarray(self._plan[0], dtype=float).copy()
            if len(self._plan) == 1:
                self.last_action = act.copy()
            self._plan = self._plan[1:]
            self.k += 1
            return act

    def has_any(self) -> bool:
        with self.lock:
            return len(self._plan) > 0


class TemporalEnsemblingBuffer:
    KEEP_BEHIND = 10  # votes for timesteps older than current_t - 10 are discarded

    def __init__(self, max_timesteps: int = 10000, chunk_size: int = 50, state_dim: int = 14, exp_weight_m: float = 0.01):
        self.max_timesteps, self.chunk_size, self.state_dim, self.exp_weight_m = max_timesteps, chunk_size, state_dim, exp_weight_m
        self.lock = threading.Lock()
        self.reset()

    def reset(self) -> None:
        with self.lock:
            self.predictions: dict[int, list] = {}  # timestep -> [(chunk number, action)], in arrival (= chunk number) order
            self.current_t = 0
            self.inference_count = 0
            self.last_action = None

    def add_chunk(self, actions_chunk, start_timestep: int | None = None) -> None:
        with self.lock:
            if actions_chunk is None or len(actions_chunk) == 0:
                return
            t0 = self.current_t if start_timestep is None else start_timestep
            idx = self.inference_count
            self.inference_count += 1
            for i, action in enumerate(actions_chunk):
                if t0 + i >= 0:
                    self.predictions.setdefault(t0 + i, []).append((idx, np.array(action, copy=True)))
            floor = max(0, self.current_t - self.KEEP_BEHIND)
            for t in [t for t in self.predictions if t < floor]:
                del self.predictions[t]

    def _aggregate(self, t: int):
        votes = self.predictions.get(t)
        if not votes:
            return self.last_action
        if len(votes) == 1:
            out = votes[0][1].copy()
        else:
            acts = np.array([a for _, a in sorted(votes, key=lambda v: v[0])])
            w = np.exp(-self.exp_weight_m * np.arange(len(votes)))
            out = (acts * (w / w.sum())[:, None]).sum(axis=0)
        self.last_action = out.copy()
        return out

    def get_action(self, timestep: int | None = None):
        with self.lock:
            return self._aggregate(self.current_t if timestep is None else timestep)

    def pop_next_action(self):
        with self.lock:
            out = self._aggregate(self.current_t)
            self.current_t += 1
            return out

    def has_prediction(self, timestep: int | None = None) -> bool:
        with self.lock:
            return bool(self.predictions.get(self.current_t if timestep is None else timestep))

    def get_current_timestep(self) -> int:
        with self.lock:
            return self.current_t
