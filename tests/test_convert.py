"""On-disk formats on CPU: MLX-format safetensors written by mlx_parallm_amd.convert load in the
oracle's loader (which reads what the reference reads) and quantise exactly as the oracle's restatement of
``nn.quantize`` does."""
import json

import numpy as np
import pytest
import torch
from safetensors import safe_open

from mlx_parallm_amd import convert as cv
from oracle import ref_generate, ref_quant


@pytest.fixture(scope="module")
def tiny(tmp_path_factory):
    from mlx_parallm_amd.tiny_model import build_tiny_model

    d = tmp_path_factory.mktemp("cv") / "tiny"
    cfg = build_tiny_model(d, seed=11, vocab_size=320, hidden_size=64, layers=2, heads=2, kv_heads=2,
                           intermediate_size=128, quantize_model=False, dtype="float32")
    return str(d), cfg


def test_make_shards_and_save_weights(tmp_path):
    w = {f"t{i}": torch.full((1 << 18,), float(i)) for i in range(5)}           # 1 MiB each
    shards = cv.make_shards(w, max_file_size_gb=0)                               # limit 0: one tensor per shard (after the first)
    assert [list(s) for s in shards][1:] == [["t0"], ["t1"], ["t2"], ["t3"], ["t4"]] and shards[0] == {}
    assert len(cv.make_shards(w)) == 1
    cv.save_weights(tmp_path / "one", w)
    idx = json.loads((tmp_path / "one" / "model.safetensors.index.json").read_text())
    assert idx["metadata"]["total_size"] == 5 << 20 and set(idx["weight_map"].values()) == {"model.safetensors"}
    assert list(idx["weight_map"]) == sorted(idx["weight_map"])
    with safe_open(str(tmp_path / "one" / "model.safetensors"), "pt") as f:
        assert f.metadata() == {"format": "mlx"} and set(f.keys()) == set(w)
    big = {f"t{i}": torch.zeros((3 << 28,), dtype=torch.uint8) for i in range(2)}    # 2 x 0.75 GiB against a 1 GiB limit
    assert [list(s) for s in cv.make_shards(big, max_file_size_gb=1)] == [["t0"], ["t1"]]


def test_convert_quantize_matches_oracle_quantiser_and_loads(tiny, tmp_path):
    src, cfg = tiny
    out = tmp_path / "q4"
    cv.convert(src, str(out), quantize=True, q_group_size=64, q_bits=4)
    conf = json.loads((out / "config.json").read_text())
    assert conf["quantization"] == {"group_size": 64, "bits": 4} and list(conf) == sorted(conf)
    assert (out / "tokenizer.json").exists() and (out / "tokenizer_config.json").exists()
    w = cv.load_weights_dir(out)
    dense = cv.load_weights_dir(src)
    name = "model.layers.1.mlp.down_proj"
    assert w[name + ".weight"].dtype in (torch.int32, torch.uint32) and w[name + ".scales"].dtype == torch.float16
    assert "model.embed_tokens.scales" in w and w["model.norm.weight"].dtype == torch.float16     # embedding quantised, norms not
    p, s, b = ref_quant.quantize(dense[name + ".weight"].to(torch.float16).float().numpy(), 64, 4, "float16")
    assert np.array_equal(w[name + ".weight"].numpy().view(np.uint32), p)
    assert np.array_equal(w[name + ".scales"].float().numpy(), s) and np.array_equal(w[name + ".biases"].float().numpy(), b)
    # the converted directory runs in the oracle (the reference's loader semantics), close to the dense model
    toks = np.array([[5, 6, 7, 8, 9]])
    q_logits = ref_generate.load(str(out))(toks, cache=None)
    d_logits = ref_generate.load(src)(toks, cache=None)
    assert q_logits.shape == d_logits.shape and 0 < np.abs(q_logits - d_logits).max()
    assert np.corrcoef(q_logits.ravel(), d_logits.ravel())[0, 1] > 0.9          # int4 noise, same function
    # and back: dequantize reproduces dequantised values, drops the quantization entry
    back = tmp_path / "deq"
    cv.convert(str(out), str(back), dequantize=True, dtype="float16")
    wb = cv.load_weights_dir(back)
    assert "quantization" not in json.loads((back / "config.json").read_text()) and name + ".scales" not in wb
    want = ref_quant.dequantize(p, s, b, 64, 4)
    assert np.allclose(wb[name + ".weight"].float().numpy(), want, atol=2e-3)
    with pytest.raises(ValueError):
        cv.convert(src, str(tmp_path / "x"), quantize=True, dequantize=True)
    with pytest.raises(NotImplementedError):
        cv.convert(src, str(tmp_path / "x"), upload_repo="someone/model")


def test_convert_dtype_only(tiny, tmp_path):
    src, _ = tiny
    cv.convert(src, str(tmp_path / "bf16"), dtype="bfloat16")
    w = cv.load_weights_dir(tmp_path / "bf16")
    assert all(t.dtype == torch.bfloat16 for t in w.values())
    assert "quantization" not in json.loads((tmp_path / "bf16" / "config.json").read_text())
