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
