"""bench.py's output contract (-m gpu): ONE JSON line on stdout with the fields the driver reads, `roofline` for the
dominant kernel and `cpu_baseline` from the oracle's C restatement.  Run on the tiny shape so that it takes seconds."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_bench_prints_one_json_line_with_the_contract_fields():
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "tiny-bf16", "--steps", "6", "--warmup", "2",
                          "--context", "64", "--batch", "2"], cwd=str(ROOT), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["metric"] == "decode_tokens_per_sec" and j["unit"] == "tokens/s" and j["n_gpus"] == 1
    assert j["steps"] == 6 and j["warmup"] == 2 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 0 and abs(j["value"] - 2 * 6 / (j["ms_per_step"] * 6e-3)) / j["value"] < 0.01
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r and r["launches"] > 0
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "tokens/s" and c["sample"]


def test_bench_gpus_2_as_a_plain_command_runs_two_ranks_with_engines():
    """`python bench.py --gpus 2` (no torchrun): self-launch, rendezvous, bucketed weight broadcast, two engines decoding
    their own shards, barrier + max-over-ranks timing, one JSON line.  On this one-GPU box the ranks share cuda:0 and
    talk over gloo (--same-device --backend gloo); on a multi-GPU node the same command uses RCCL."""
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device",
                          "--workload", "tiny-bf16", "--steps", "6", "--warmup", "2", "--context", "64", "--batch", "2",
                          "--no-cpu-baseline"], cwd=str(ROOT), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["config"]["global_batch"] == 4 and j["scaling"] == "weak"
    assert j["broadcast_buckets"] >= 1 and j["broadcast_bytes"] > 0 and j["broadcast_seconds"] > 0
    assert j["value"] > 0 and abs(j["value"] - 2 * 2 * 6 / (j["ms_per_step"] * 6e-3)) / j["value"] < 0.01
    # the headline leg is the reference's numerics (float32 KV); the 16-bit-KV leg rides along as fast_mode
    assert "float32" in j["config"]["kv_dtype"] and "fast_mode" in j and j["fast_mode"]["value"] > 0
    assert "BatchedKVCache" in j["fast_mode"]["what"] and "reference_numerics" not in j
    pr = j["per_rank_tokens_per_sec"]                      # each rank's own K steps: min <= max, and the job total is at most the sum
    assert 0 < pr["min"] <= pr["max"] and j["value"] <= 2 * pr["max"] * 1.01


def test_dp_generate_two_ranks_equal_one_process(tiny_dirs, tmp_path):
    """mlx_parallm_amd.dp_generate: prompts sharded over two ranks (weights replicated from rank 0 by bucketed
    broadcast) give the responses of the single-process batch_generate on each shard, in prompt order."""
    from mlx_parallm_amd import utils
    from mlx_parallm_amd.distributed import shard_range

    d = tiny_dirs["llama_q4_f32"][0]
    prompts = ["first", "second prompt", "a third, longer prompt", "4th", "and the fifth one"]
    pf = tmp_path / "prompts.json"
    pf.write_text(json.dumps(prompts))
    res = subprocess.run([sys.executable, "-m", "mlx_parallm_amd.dp_generate", "--model-path", d, "--prompts-file", str(pf),
                          "--gpus", "2", "--backend", "gloo", "--same-device", "--max-tokens", "8", "--kv-dtype", "float32"],
                         cwd=str(ROOT), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    j = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["n_prompts"] == 5 and j["broadcast_bytes"] > 0
    utils._kv_pool.clear()
    import pytest as _pt

    mp = _pt.MonkeyPatch()
    mp.setattr(utils, "DEFAULT_KV_DTYPE", "float32")
    try:
        model, tok = utils.load(d)
        want = []
        for r in range(2):
            s, e = shard_range(len(prompts), r, 2)
            want += utils.batch_generate(model, tok, prompts[s:e], max_tokens=8, temp=0.0)
        model.engine.close()
    finally:
        mp.undo()
        utils._kv_pool.clear()
    assert j["responses"] == want
