import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("TOKENIZERS_PARALLELISM", "false")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def tiny_dirs(tmp_path_factory):
    """Builds the tiny checkpoints used across tests once per session (name -> (dir, cfg))."""
    from mlx_parallm_amd.tiny_model import build_tiny_model

    root = tmp_path_factory.mktemp("tiny")
    out = {}

    def make(name, **kw):
        d = root / name
        cfg = build_tiny_model(d, **kw)
        out[name] = (str(d), cfg)

    # config 1 of BASELINE.json: scripts/build_tiny_model.py defaults (fp32 activations, int4 g64,
    # tied + quantised embedding) at a small vocabulary for speed
    make("llama_q4_f32", seed=0, vocab_size=512, dtype="float32", quantize_model=True)
    make("llama_f32", seed=1, vocab_size=512, dtype="float32", quantize_model=False)
    make("llama_bf16_gqa", seed=2, vocab_size=512, dtype="bfloat16", quantize_model=False, hidden_size=128,
         layers=3, heads=8, kv_heads=2, intermediate_size=256, head_dim=16, tie_word_embeddings=False,
         norm_jitter=0.1)
    make("llama_q4_bf16", seed=3, vocab_size=512, dtype="bfloat16", quantize_model=True, hidden_size=128,
         layers=2, heads=8, kv_heads=2, intermediate_size=256, head_dim=16, tie_word_embeddings=False,
         norm_jitter=0.1)
    make("qwen3_bf16", seed=4, model_type="qwen3", vocab_size=512, dtype="bfloat16", quantize_model=False,
         hidden_size=128, layers=2, heads=5, kv_heads=1, intermediate_size=256, head_dim=64,
         tie_word_embeddings=False, norm_jitter=0.1)
    make("llama_q8_f16", seed=5, vocab_size=512, dtype="float16", quantize_model=True, q_bits=8, hidden_size=128,
         layers=2, heads=4, kv_heads=4, intermediate_size=256, tie_word_embeddings=True)
    # a dense f16 model wide enough (K % 256 == 0) for the [hi | lo] matrix-core path of the float32-activation mode
    make("llama_f16", seed=6, vocab_size=512, dtype="float16", quantize_model=False, hidden_size=256, layers=2, heads=4,
         kv_heads=2, intermediate_size=512, head_dim=64, tie_word_embeddings=False, norm_jitter=0.1)
    return out
