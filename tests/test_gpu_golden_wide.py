"""-m gpu: the HIP path against the oracle at PRODUCTION widths (tests/golden/wide_*.npz, made by
tests/golden/make_golden_wide.py): Mistral-7B and Qwen3-14B layer shapes truncated to 2 decoder blocks, bf16 /
int4-g64 / int4 + rank-16 LoRA on q,v, both KV modes, batch 8 with a 1024-token prompt decoded across KV length
1024 -> 1100 (the bench's regime: second 256-key round of the split-KV decode attention, K = 4096 / 5120 / 14336 /
17408 linears, V = 32000 / 151936 sampler rows), plus the batch-32 (config 4) and ragged batch-64 + LoRA (config 5)
decode steps.  The checkpoints are rebuilt here from the seeds in each file's spec (tests/wide_models.py), loaded
through utils.load_model / utils.load_adapters and driven through the C ABI (mi_step_enqueue / mi_step_wait).

Bar (BASELINE.json north_star), teacher-forced with the oracle's tokens so that one flip cannot cascade:
  * PagedKVCache (float32-KV) mode = the reference's default numerics.  Layer 0 of that mode still rounds q / k / v
    and the MLP input to bf16 (quirk Q2: everything before the first float32 cache is the 16-bit model), and a bf16
    rounding of a K = 4096-term float32 sum flips whenever the sum lies within its accumulation error of a rounding
    boundary (~0.2 % of the elements, each then off by 2^-8 relative).  The oracle rounds the exactly rounded sums,
    any float32-accumulating implementation (MLX's own included) rounds its own: measured with this build's EXACT
    float32 VALU kernels (tools/debug/wide_f32_error.py) that alone moves single logits by up to 2.5e-3 (rms 3e-4) at
    these widths.  So: chosen-token and top-8 logprobs within 1e-3 ON AVERAGE and 2e-2 at worst, greedy ids equal
    except where the oracle's own top-2 margin is <= 1e-2 (counted; at most two per case).  The 64..128-wide models of
    tests/test_gpu_engine.py, where such flips are rare, hold the 1e-3 bound on every value;
  * BatchedKVCache (model-dtype KV) mode: logits are bf16 values (ulp 0.0156-0.031 at |logit| 2-8) and the top of a
    32000 / 151936-way random-weight distribution is dense, so ids must match wherever the oracle's margin exceeds
    MODELKV_MARGIN, a differing id must be one of the oracle's three largest, and logprobs must stay within MODELKV_LP.
"""
import gc
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import wide_models  # noqa: E402
from mlx_parallm_amd import utils  # noqa: E402
from mlx_parallm_amd.engine import SampleArgs  # noqa: E402

WIDE = sorted((Path(__file__).resolve().parent / "golden").glob("wide_*.npz"))

EXACT_MARGIN, EXACT_LP, EXACT_LP_MEAN = 1e-2, 2e-2, 1e-3
MODELKV_MARGIN, MODELKV_LP = 0.13, 0.1


class _Checkpoints:
    """One checkpoint directory + engine at a time (files of up to 4.5 GB, rebuilt from seeds)."""

    def __init__(self, root):
        self.root, self.key, self.model, self.cfg, self.dir, self.adapted = root, None, None, None, None, False

    def _drop(self):
        if self.model is not None:
            self.model.engine.close()
        self.model = None
        gc.collect()

    def get(self, family, precision, seed, lora, adapter_seed):
        key = (family, precision, seed)
        if key != self.key:
            self._drop()
            if self.dir is not None:
                for f in Path(self.dir).rglob("*"):
                    if f.is_file():
                        f.unlink()
            self.dir = self.root / f"{family}-{precision}-{seed}"
            self.cfg = wide_models.build_checkpoint(self.dir, family, precision, seed)
            self.key, self.adapted = key, False
        if self.model is None or (self.adapted and not lora):
            self._drop()
            self.model = utils.load_model(str(self.dir), max_positions=wide_models.MAX_POS)
            self.adapted = False
        if lora and not self.adapted:
            ad = self.dir / "adapter"
            wide_models.build_adapter(ad, self.cfg, adapter_seed)
            utils.load_adapters(self.model, str(ad))
            self.adapted = True
        return self.model, self.cfg

    def close(self):
        self._drop()


@pytest.fixture(scope="module")
def checkpoints(tmp_path_factory):
    c = _Checkpoints(tmp_path_factory.mktemp("wide"))
    yield c
    c.close()


def test_wide_golden_files_present():
    assert len(WIDE) >= 17, "run tests/golden/make_golden_wide.py"


@pytest.mark.parametrize("path", WIDE, ids=[p.stem for p in WIDE])
def test_device_matches_oracle_at_production_width(checkpoints, path):
    g = np.load(path)
    spec = json.loads(str(g["spec"]))
    model, cfg = checkpoints.get(spec["family"], spec["precision"], spec["model_seed"], bool(spec.get("lora")),
                                 spec["adapter_seed"])
    B, steps, exact = spec["B"], spec["steps"], bool(spec["paged"])
    greedy = spec["temp"] == 0.0
    prompts = wide_models.prompts_for(spec, cfg["vocab_size"])
    kv = model.engine.new_kv(B, capacity=spec["L0"] + steps + 2, kv_dtype="float32" if exact else "model")
    margin_eps, lp_eps = (EXACT_MARGIN, EXACT_LP) if exact else (MODELKV_MARGIN, MODELKV_LP)
    y = prompts
    near, lp_err, top_err, decided, lp_sum, lp_n = 0, 0.0, 0.0, 0, 0.0, 0
    for s in range(steps):
        sp = SampleArgs(temp=spec["temp"], top_p=spec["top_p"], uniforms=None if greedy else g["uniforms"][s],
                        top_logprobs=8)
        res = model.engine.decode_sample(kv, y.astype(np.int32), sp)
        want = g["tokens"][s]
        # the oracle's 8 largest logprobs: log Z from the chosen token's logit and logprob (greedy: the largest logit)
        for b in range(B):
            gt, wt = int(res["tokens"][b]), int(want[b])
            ids8, vals8 = g["top_ids"][s, b], g["top_vals"][s, b]
            if gt != wt:
                near += 1
                if greedy:
                    assert g["margins"][s, b] <= margin_eps and gt in ids8[:3], (path.stem, s, b, gt, wt, float(g["margins"][s, b]))
                elif gt in ids8:               # a different draw that is among the oracle's 8 most likely: its logprob must be the oracle's
                    logz = float(vals8[list(ids8).index(wt)]) - float(g["logprobs"][s, b]) if wt in ids8 else None
                    if logz is not None:
                        top_err = max(top_err, abs(float(res["logprobs"][b]) - (float(vals8[list(ids8).index(gt)]) - logz)))
                continue
            if g["margins"][s, b] > margin_eps:
                decided += 1
            lp_err = max(lp_err, abs(float(res["logprobs"][b]) - float(g["logprobs"][s, b])))
            lp_sum, lp_n = lp_sum + abs(float(res["logprobs"][b]) - float(g["logprobs"][s, b])), lp_n + 1
            if wt in ids8:
                logz = float(vals8[list(ids8).index(wt)]) - float(g["logprobs"][s, b])
                dev = dict(zip(res["top_ids"][b].tolist(), res["top_logprobs"][b].tolist()))
                for i, v in zip(ids8.tolist(), vals8.tolist()):
                    if i in dev:
                        top_err = max(top_err, abs(dev[i] - (v - logz)))
        y = want[:, None]
    total = steps * B
    print(f"{path.stem}: mismatching ids {near}/{total} (oracle margins <= {margin_eps}: {int((g['margins'] <= margin_eps).sum())}), "
          f"|logprob - oracle| max {lp_err:.2e} mean {lp_sum / max(lp_n, 1):.2e}, max top-8 logprob error {top_err:.2e}")
    assert lp_err <= lp_eps and top_err <= lp_eps, (path.stem, lp_err, top_err)
    if exact:
        assert lp_sum / max(lp_n, 1) <= EXACT_LP_MEAN, (path.stem, lp_sum / max(lp_n, 1))
    if exact and greedy:
        assert near <= 2, (path.stem, near, total)
    elif greedy:
        assert near <= int((g["margins"] <= margin_eps).sum()), (path.stem, near, total)
        assert decided >= total // 3, (path.stem, decided, total)      # the id check really decided a good share of the steps
    else:
        # Sampled cases (config 3: top-p 0.9, T = 1): the nucleus of a 32000-way random-weight distribution holds thousands of
        # tokens of ~1e-4 probability each, so the inverse-CDF draw moves to a neighbouring token when the cumulative sum
        # shifts by 1e-4 -- i.e. under the float32 noise documented above (measured: 90 of 192 draws differ while every
        # logprob agrees to 1e-3).  The draw itself is therefore checked where it is checkable: against the oracle's sampler on
        # the SAME logits (tests/test_gpu_kernels.py, V up to 151936) and against the nucleus of the device's own logits at
        # full size (tests/test_gpu_fullsize.py::test_sampling_with_logprobs_config3).  Here: the distribution (top-8 logprobs,
        # chosen-token logprobs on equal draws) must be the oracle's, and the draws must not be degenerate.
        assert near < total, (path.stem, near, total)
    assert kv.offsets == [spec["L0"] + steps - 1] * B            # the prompt + (steps - 1) fed-back tokens
    kv.close()
