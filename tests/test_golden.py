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

GOLDEN = sorted(p for p in (Path(__file__).resolve().parent / "golden").glob("*.npz")
                if not p.stem.startswith(("wide_", "serving_")))
# production-width cases (make_golden_wide.py): minutes of oracle time each, so the CPU suite only checks the files
WIDE = sorted(p for p in (Path(__file__).resolve().parent / "golden").glob("wide_*.npz") if not p.stem.endswith("_logits"))
SERVING = sorted((Path(__file__).resolve().parent / "golden").glob("serving_*.npz"))
ENVELOPE = Path(__file__).resolve().parent / "golden" / "envelope"


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


@pytest.mark.parametrize("path", SERVING, ids=[p.stem for p in SERVING])
def test_serving_golden_files_are_consistent(path):
    """tests/golden/serving_*.npz (make_golden_serving.py): the oracle's solo runs of the four sequences of the serving
    schedule -- shapes follow the schedule in the spec, greedy tokens are the arg-max ids."""
    g = np.load(path)
    spec = json.loads(str(g["spec"]))
    assert len(spec["prompt_lens"]) == 4 and sum(spec["chunks"]) == spec["prompt_lens"][3]
    for i in range(4):
        n = 1 + spec["steps"] if i < 3 else spec["steps"] - (spec["arrive_step"] + len(spec["chunks"])) + 1
        assert g[f"seq{i}_tokens"].shape == (n,) and g[f"seq{i}_top_ids"].shape == (n, 8)
        assert np.array_equal(g[f"seq{i}_tokens"], g[f"seq{i}_top_ids"][:, 0])
        assert np.all(g[f"seq{i}_logprobs"] <= 0) and np.all(g[f"seq{i}_margins"] >= 0)
    assert max(spec["prompt_lens"][:3]) + spec["steps"] > 1024          # the decode rows cross the 16-block boundary


def test_every_wide_case_has_its_accumulation_envelope():
    """tests/golden/envelope/<case>.npz (make_golden_wide.py --envelope): the oracle re-run with float32 accumulators in two
    summation orders.  The GPU tests take their tolerances from these files, so each must be there, cover the case's
    (step, row) grid and carry a summary that is what its arrays say."""
    missing = [p.stem for p in WIDE if not (ENVELOPE / p.name).exists()]
    assert not missing, f"run tests/golden/make_golden_wide.py --envelope {' '.join(missing)}"
    for p in WIDE:
        g, e = np.load(p), np.load(ENVELOPE / p.name)
        summ = json.loads(str(e["summary"]))
        for mode in ("f32_seq32", "f32_pairwise"):
            lp = e[f"lp_{mode}"]
            assert lp.shape == g["logprobs"].shape and e[f"top_lp_{mode}"].shape == g["top_vals"].shape
            d = np.abs(lp.astype(np.float64) - g["logprobs"].astype(np.float64))
            assert abs(float(d.max()) - summ[mode]["max_lp"]) <= 1e-6 and abs(float(d.mean()) - summ[mode]["mean_lp"]) <= 1e-6
            assert summ[mode]["id_flips"] == int((e[f"argmax_{mode}"] != g["top_ids"][:, :, 0]).sum())
        # the two orders are different arithmetic: they do not coincide, and neither is far from the exact oracle
        assert summ["seq32_vs_pairwise"]["max_lp"] > 0
        # (16-bit mode: errors are whole one-ulp flips of 16-bit values, and they add up like a random walk over the decoder
        # blocks -- the 2-block cases stay under 0.1, the full-depth ones (32 / 40 blocks, round 4) under 0.1 * sqrt(blocks / 2):
        # measured 0.16 / 0.19.  The float32-KV bound does not move with depth: 1.4e-3 at full depth.)
        spec = json.loads(str(g["spec"]))
        depth = (spec.get("layers", 2) / 2.0) ** 0.5
        assert max(summ["f32_seq32"]["max_lp"], summ["f32_pairwise"]["max_lp"]) <= (0.1 * depth if not spec["paged"] else 2e-2)


def test_full_logits_of_the_sampled_case_are_the_oracles():
    """wide_mistral_int4_topp_paged_logits.npz: full last-position logits of a few (step, row) pairs; their 8 largest values
    must be the ones stored in the case file."""
    f = Path(__file__).resolve().parent / "golden" / "wide_mistral_int4_topp_paged_logits.npz"
    assert f.exists(), "run tests/golden/make_golden_wide.py --logits"
    full, g = np.load(f), np.load(f.with_name("wide_mistral_int4_topp_paged.npz"))
    for (s_, b), lg in zip(full["pairs"].tolist(), full["logits"]):
        order = np.argsort(-lg, kind="stable")[:8]
        assert np.array_equal(order, g["top_ids"][s_, b]) and np.allclose(lg[order], g["top_vals"][s_, b], atol=1e-6)
