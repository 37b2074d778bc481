"""CPU: the oracle reproduces the committed golden vectors (tests/golden/*.npz, made by
tests/golden/make_golden.py).  This pins the oracle against regressions; the GPU path is checked
against the same files in tests/test_gpu_golden.py."""
import json
import tempfile
from pathlib import Path

import numpy as np
import pytest

from mlx_parallm_amd.tiny_model import build_tiny_model
from oracle import ref_generate

GOLDEN = sorted(p for p in (Path(__file__).resolve().parent / "golden").glob("*.npz") if not p.stem.startswith("wide_"))
# production-width cases (make_golden_wide.py): minutes of oracle time each, so the CPU suite only checks the files
WIDE = sorted((Path(__file__).resolve().parent / "golden").glob("wide_*.npz"))


def test_golden_files_present():
    assert len(GOLDEN) >= 6


@pytest.mark.parametrize("path", GOLDEN, ids=[p.stem for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    g = np.load(path)
    spec = json.loads(str(g["spec"]))
    if path.stem == "tiny_default_greedy_b1":
        steps = 4          # 151936-token vocabulary: keep the CPU suite fast
    else:
        steps = min(spec["steps"], 6)
    with tempfile.TemporaryDirectory() as d:
        build_tiny_model(d, **spec["model"])
        ref = ref_generate.load(d, max_pos=256)
        gen = ref_generate.generate_step(g["prompts"], ref, temp=spec["temp"], top_p=spec["top_p"], paged=spec["paged"],
                                         uniforms_fn=lambda s: g["uniforms"][s], return_logits=True)
        for s, ((t, _p, logits, lp), _) in enumerate(zip(gen, range(steps))):
            assert np.array_equal(t[:, 0], g["tokens"][s]), (path.stem, s)
            assert np.allclose(lp, g["logprobs"][s], atol=1e-5)
            top = np.take_along_axis(logits, g["top_ids"][s].astype(np.int64), axis=-1)
            assert np.allclose(top, g["top_vals"][s], atol=1e-5)


@pytest.mark.parametrize("path", WIDE, ids=[p.stem for p in WIDE])
def test_wide_golden_files_are_consistent(path):
    """tests/golden/wide_*.npz (oracle at the Mistral-7B / Qwen3-14B layer shapes): spec and array shapes agree, greedy
    tokens are the argmax ids, logprobs are log-softmax values.  The oracle run itself takes minutes per case
    (make_golden_wide.py); the GPU side is tests/test_gpu_golden_wide.py."""
    g = np.load(path)
    spec = json.loads(str(g["spec"]))
    S, B = spec["steps"], spec["B"]
    assert g["tokens"].shape == (S, B) and g["logprobs"].shape == (S, B)
    assert g["top_ids"].shape == (S, B, 8) and g["top_vals"].shape == (S, B, 8) and g["margins"].shape == (S, B)
    assert np.all(np.diff(g["top_vals"], axis=-1) <= 0) and np.all(g["logprobs"] <= 0)
    assert np.allclose(g["margins"], g["top_vals"][..., 0] - g["top_vals"][..., 1])
    if spec["temp"] == 0.0:
        assert np.array_equal(g["tokens"], g["top_ids"][..., 0])
