"""The server as the reference tests it (tests/helpers.py + test_server_basic.py + test_server_batching.py):
``python -m mlx_parallm_amd.cli`` in a child process on the GPU, tiny int4 model, real HTTP."""
import concurrent.futures
import json
import os
import signal
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np
import pytest
import requests

from oracle import ref_generate, ref_sample

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


@pytest.fixture(scope="module", params=["default", "continuous", "replicas"])   # replicas: two model copies ("--devices 0,0")
def server(request, tiny_dirs, tmp_path_factory):
    model_dir = tiny_dirs["llama_q4_f32"][0]
    port = _free_port()
    log_path = tmp_path_factory.mktemp("server") / "server.log"
    env = os.environ.copy()
    env["PYTHONPATH"] = str(ROOT) + os.pathsep + env.get("PYTHONPATH", "")
    args = [sys.executable, "-m", "mlx_parallm_amd.cli", "--model-path", model_dir, "--host", "127.0.0.1", "--port", str(port),
            "--max-batch-size", "8", "--batch-timeout", "0.2", "--diverse-mode", "false"]
    args += ["--devices", "0,0", "--max-batch-size", "4"] if request.param == "replicas" else ["--scheduler", request.param]
    with open(log_path, "w", buffering=1) as lf:
        proc = subprocess.Popen(args, cwd=str(ROOT), stdout=lf, stderr=subprocess.STDOUT, text=True, env=env)
    base = f"http://127.0.0.1:{port}"
    import fastapi, transformers, uvicorn  # noqa: F401,E401  (pages the child's imports in: a fresh box reads them cold)
    deadline = time.time() + 600
    ok = False
    while time.time() < deadline and proc.poll() is None:
        try:
            if requests.get(f"{base}/health", timeout=2).ok:
                ok = True
                break
        except Exception:
            time.sleep(0.5)
    if not ok:
        proc.kill()
        pytest.fail("server did not come up:\n" + log_path.read_text()[-3000:])
    yield base, model_dir, request.param
    proc.send_signal(signal.SIGTERM)
    try:
        proc.wait(timeout=20)
    except subprocess.TimeoutExpired:
        proc.kill()


def test_models_completion_chat_and_metrics(server):
    base, model_dir, mode = server
    data = requests.get(f"{base}/v1/models", timeout=10).json()["data"]
    assert any(m["id"] == model_dir and m["status"] == "loaded" for m in data)
    ref = ref_generate.load(model_dir, max_pos=512)
    from mlx_parallm_amd.tokenizer_utils import load_tokenizer

    tok = load_tokenizer(model_dir)
    prompt = "Say hello in one word."
    ids = np.asarray(tok.encode(prompt))[None]
    want = [int(t[0, 0]) for (t, _), _ in zip(ref_generate.generate_step(ids, ref, paged=False), range(8))]
    if tok.eos_token_id in want:
        want = want[:want.index(tok.eos_token_id)]
    r = requests.post(f"{base}/v1/completions", json={"model": model_dir, "prompt": prompt, "max_tokens": 8, "temperature": 0.0}, timeout=120)
    assert r.status_code == 200, r.text[:500]
    j = r.json()
    assert j["choices"][0]["text"] == tok.decode(want, skip_special_tokens=True)       # greedy ids == oracle's
    assert j["usage"]["prompt_tokens"] == ids.shape[1] and j["usage"]["completion_tokens"] == len(want)

    r = requests.post(f"{base}/v1/chat/completions", timeout=120, json={
        "model": model_dir, "messages": [{"role": "user", "content": "Return exactly one word."}], "max_tokens": 8,
        "temperature": 0.7, "top_p": 0.95, "n": 2})
    assert r.status_code == 200 and len(r.json()["choices"]) == 2

    before = requests.get(f"{base}/debug/metrics", timeout=5).json()["batches_processed"]
    payloads = [{"model": model_dir, "prompt": f"Request {i}: Say hello.", "max_tokens": 8, "temperature": 0.0} for i in range(8)]
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        outs = list(ex.map(lambda p: requests.post(f"{base}/v1/completions", json=p, timeout=120).json(), payloads))
    assert all("choices" in o for o in outs)
    m = requests.get(f"{base}/debug/metrics", timeout=5).json()
    assert m["decode_tps_last"] > 0 and m["decode_tokens_total"] > 0
    if mode == "default":
        assert before + 1 <= m["batches_processed"] <= before + 3     # the 8 concurrent requests shared windows
    else:
        assert m["batches_processed"] == 0                             # continuous: no windows, slots
    assert requests.post(f"{base}/v1/completions", json={"model": "nope", "prompt": "x", "max_tokens": 1}, timeout=30).status_code == 404


def test_logprobs_echo_perplexity_on_device(server):
    base, model_dir, mode = server
    from mlx_parallm_amd.tokenizer_utils import load_tokenizer

    tok = load_tokenizer(model_dir)
    ref = ref_generate.load(model_dir, max_pos=512)
    prompt = "Hello world"
    ids = np.asarray(tok._tokenizer([prompt], return_tensors="np")["input_ids"])
    lsm = ref_sample.log_softmax(np.asarray(ref(ids, cache=ref.make_cache(1, paged=False)), dtype=np.float32)[0])
    want = [float(lsm[i, ids[0, i + 1]]) for i in range(ids.shape[1] - 1)]
    r = requests.post(f"{base}/v1/completions", timeout=120, json={
        "model": model_dir, "prompt": prompt, "max_tokens": 0, "temperature": 0.0, "top_p": 1.0, "logprobs": 2, "echo": True})
    assert r.status_code == 200, r.text[:500]
    lp = r.json()["choices"][0]["logprobs"]
    np.testing.assert_allclose(lp["token_logprobs"], want, atol=1e-3)           # north_star: logprobs within 1e-3
    np.testing.assert_allclose([max(d.values()) for d in lp["top_logprobs"]], lsm[:-1].max(axis=1), atol=1e-3)
    r = requests.post(f"{base}/v1/completions", timeout=120, json={
        "model": model_dir, "prompt": prompt, "max_tokens": 4, "logprobs": 1, "echo": True})
    assert r.status_code == 200 and r.json()["usage"]["completion_tokens"] >= 1
    j = requests.post(f"{base}/v1/perplexity", json={"model": model_dir, "text": prompt}, timeout=120).json()
    assert j["token_count"] == len(want) and abs(j["avg_nll"] + np.mean(want)) < 1e-3


def test_streams_finish(server):
    base, model_dir, mode = server
    for url, payload in (("/v1/chat/completions", {"messages": [{"role": "user", "content": "In one sentence, describe a tree."}]}),
                         ("/v1/completions", {"prompt": "In one word, greet me."})):
        payload.update(model=model_dir, max_tokens=12, temperature=0.7, top_p=0.95, stream=True)
        with requests.post(base + url, json=payload, stream=True, timeout=120) as r:
            assert r.status_code == 200
            done = False
            for _, line in zip(range(400), r.iter_lines(decode_unicode=True)):
                if not line:
                    continue
                if line.strip() == "data: [DONE]":
                    done = True
                    break
                assert line.startswith("data: ")
                json.loads(line[len("data: "):])
            assert done
