"""``TokenizerWrapper`` / ``load_tokenizer``: the slice of ``mlx_lm.tokenizer_utils`` that the
reference's generation loop touches (``mlx_parallm/utils.py:24,449-471,498-531,1011-1068``).

mlx-lm is not installable here, so the wrapper is restated over a HuggingFace tokenizer:
attribute pass-through to ``_tokenizer`` plus a ``detokenizer`` with
``reset / add_token / finalize / last_segment / text`` (the naive streaming detokenizer:
decode everything generated so far and emit the new suffix, holding back an incomplete
UTF-8 character).
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional


class NaiveStreamingDetokenizer:
    def __init__(self, tokenizer):
        self._tokenizer = tokenizer
        self.reset()

    def reset(self):
        self.tokens: List[int] = []
        self._text = ""
        self._emitted = 0
        self._final = False

    def add_token(self, token: int):
        self.tokens.append(int(token))
        self._text = self._tokenizer.decode(self.tokens)

    def finalize(self):
        self._text = self._tokenizer.decode(self.tokens)
        self._final = True

    @property
    def text(self) -> str:
        if not self._final and self._text.endswith("�"):
            return self._text[:-1]
        return self._text

    @property
    def last_segment(self) -> str:
        t = self.text
        seg = t[self._emitted:]
        self._emitted = len(t)
        return seg


class TokenizerWrapper:
    """Pass-through wrapper; ``wrapper._tokenizer`` is the HF tokenizer (utils.py:511-516)."""

    def __init__(self, tokenizer, detokenizer_class=NaiveStreamingDetokenizer):
        object.__setattr__(self, "_tokenizer", tokenizer)
        object.__setattr__(self, "_detokenizer", detokenizer_class(tokenizer))

    def __getattr__(self, attr):
        if attr == "detokenizer":
            return self._detokenizer
        if attr.startswith("_"):
            return object.__getattribute__(self, attr)
        return getattr(self._tokenizer, attr)

    def __setattr__(self, attr, value):
        if attr in ("detokenizer", "_detokenizer"):
            object.__setattr__(self, "_detokenizer", value)
        elif attr == "_tokenizer":
            object.__setattr__(self, attr, value)
        else:
            setattr(self._tokenizer, attr, value)

    def __call__(self, *a, **kw):
        return self._tokenizer(*a, **kw)


def load_tokenizer(model_path, tokenizer_config_extra: Optional[dict] = None) -> TokenizerWrapper:
    """``AutoTokenizer.from_pretrained(local_dir)`` wrapped (utils.py:745).  Local paths only."""
    from transformers import AutoTokenizer

    model_path = Path(model_path)
    tok = AutoTokenizer.from_pretrained(str(model_path), **(tokenizer_config_extra or {}))
    return TokenizerWrapper(tok)
