"""Prompt tokenisation for the pi0 / pi0.5 prefix: `PaligemmaTokenizer.tokenize`
(`src/openpi/models/tokenizer.py:14-47`) on top of a sentencepiece model.

The reference downloads `gs://big_vision/paligemma_tokenizer.model`; there is no network here, so the model file is an
explicit argument (path, bytes or a ready SentencePieceProcessor).  Everything around the sentencepiece call is what this
module owns and tests pin (tests/test_host_pipeline_cpu.py, against the reference's method executed from source on a tiny
sentencepiece model trained in the test):
  * prompt clean-up: strip, '_' -> ' ', newline -> ' ';
  * pi0.5: the state is discretised into 256 bins over [-1, 1] (`np.digitize` against 256 left edges, minus one) and written
    into the text: "Task: <prompt>, State: <b0 b1 ...>;\\nAction: " with BOS;
  * pi0: "<prompt>" with BOS, then the tokens of "\\n" (the start-of-answer marker);
  * right-padding with 0 / False to max_len, or truncation to max_len with a warning."""

from __future__ import annotations

import logging
import os

import numpy as np

STATE_BINS = 256


def discretize_state(state: np.ndarray) -> np.ndarray:
    """Bin index in [-1, 255]: values below -1 give -1, values >= the last edge give 255."""
    return np.digitize(state, bins=np.linspace(-1, 1, STATE_BINS + 1)[:-1]) - 1


class PaligemmaTokenizer:
    def __init__(self, max_len: int = 48, model=None):
        import sentencepiece

        self._max_len = max_len
        model = model if model is not None else os.environ.get("KAI0_PALIGEMMA_TOKENIZER")
        if model is None:
            raise FileNotFoundError("PaligemmaTokenizer needs the paligemma sentencepiece model: pass model=<path | bytes | "
                                    "SentencePieceProcessor> or set KAI0_PALIGEMMA_TOKENIZER (no download is attempted)")  # fmt: skip
        if isinstance(model, sentencepiece.SentencePieceProcessor):
            self._tokenizer = model
        elif isinstance(model, (bytes, bytearray)):
            self._tokenizer = sentencepiece.SentencePieceProcessor(model_proto=bytes(model))
        else:
            with open(model, "rb") as f:
                self._tokenizer = sentencepiece.SentencePieceProcessor(model_proto=f.read())

    def tokenize(self, prompt: str, state: np.ndarray | None = None) -> tuple[np.ndarray, np.ndarray]:
        text = prompt.strip().replace("_", " ").replace("\n", " ")
        sp = self._tokenizer
        if state is None:
            ids = sp.encode(text, add_bos=True) + sp.encode("\n")
        else:
            bins = " ".join(map(str, discretize_state(state)))
            ids = sp.encode(f"Task: {text}, State: {bins};\nAction: ", add_bos=True)
        n = len(ids)
        if n > self._max_len:
            logging.warning(f"Token length ({n}) exceeds max length ({self._max_len}), truncating. "
                            "Consider increasing the `max_token_len` in your model config if this happens frequently.")  # fmt: skip
        if n >= self._max_len:
            return np.asarray(ids[: self._max_len]), np.asarray([True] * self._max_len)
        fill = [False] * (self._max_len - n)  # the reference pads the id list with False as well (== 0)
        return np.asarray(ids + fill), np.asarray([True] * n + fill)
