"""The C restatement (oracle/c/ref_decode.c, used as bench.py's timed CPU baseline) must agree
with the NumPy oracle: same block, same bf16 rounding points."""
import ctypes as C

import numpy as np
import pytest

from oracle import c_ref, ref_model
from oracle.numerics import bf16_bits_to_f32, round_to


@pytest.mark.parametrize("qk_norm", [False, True])
def test_c_layer_step_matches_numpy_oracle(qk_norm):
    rng = np.random.default_rng(3)
    H, Hq, Hkv, D, I, V, B, steps = 64, 4, 2, 16, 96, 120, 3, 4
    cap = 8
    dt = "bfloat16"

    def w(n, k, s=0.2):
        return round_to(rng.standard_normal((n, k)).astype(np.float32) * s, dt)

    def nrm(n):
        return round_to(1 + 0.1 * rng.standard_normal(n).astype(np.float32), dt)

    p = "model.layers.0."
    W = {
        "model.embed_tokens": ref_model.Linear(dt, weight=w(V, H, 1.0)),
        p + "self_attn.q_proj": ref_model.Linear(dt, weight=w(Hq * D, H)),
        p + "self_attn.k_proj": ref_model.Linear(dt, weight=w(Hkv * D, H)),
        p + "self_attn.v_proj": ref_model.Linear(dt, weight=w(Hkv * D, H)),
        p + "self_attn.o_proj": ref_model.Linear(dt, weight=w(H, Hq * D)),
        p + "mlp.gate_proj": ref_model.Linear(dt, weight=w(I, H)),
        p + "mlp.up_proj": ref_model.Linear(dt, weight=w(I, H)),
        p + "mlp.down_proj": ref_model.Linear(dt, weight=w(H, I)),
        p + "input_layernorm": (nrm(H), dt), p + "post_attention_layernorm": (nrm(H), dt),
        p + "self_attn.q_norm": (nrm(D), dt), p + "self_attn.k_norm": (nrm(D), dt),
        "model.norm": (nrm(H), dt), "lm_head": ref_model.Linear(dt, weight=w(V, H)),
    }
    cfg = ref_model.RefConfig(model_type="qwen3" if qk_norm else "llama", hidden_size=H, num_hidden_layers=1,
                              intermediate_size=I, num_attention_heads=Hq, num_key_value_heads=Hkv, head_dim=D,
                              vocab_size=V, rms_norm_eps=1e-5, rope_theta=10000.0, tie_word_embeddings=False)
    ref = ref_model.RefModel(cfg, W, max_pos=64)
    cache = ref.make_cache(B, paged=False)

    keep = {}

    def bits(name, arr):
        keep[name] = c_ref.bf16_bits(arr)
        return keep[name].ctypes.data

    L = c_ref.RefLayer()
    L.H, L.Hq, L.Hkv, L.D, L.I, L.eps, L.rope_theta, L.qk_norm = H, Hq, Hkv, D, I, 1e-5, 10000.0, int(qk_norm)
    for f, n in (("wq", "self_attn.q_proj"), ("wk", "self_attn.k_proj"), ("wv", "self_attn.v_proj"),
                 ("wo", "self_attn.o_proj"), ("wg", "mlp.gate_proj"), ("wu", "mlp.up_proj"), ("wd", "mlp.down_proj")):
        setattr(L, f, bits(f, W[p + n].weight))
    L.in_norm = bits("in", W[p + "input_layernorm"][0])
    L.post_norm = bits("post", W[p + "post_attention_layernorm"][0])
    L.q_norm = bits("qn", W[p + "self_attn.q_norm"][0])
    L.k_norm = bits("kn", W[p + "self_attn.k_norm"][0])
    fn, hd = c_ref.bf16_bits(W["model.norm"][0]), c_ref.bf16_bits(W["lm_head"].weight)
    lib = c_ref.lib()
    kc = np.zeros((B, Hkv, cap, D), np.uint16)
    vc = np.zeros_like(kc)
    scratch = np.zeros(int(lib.ref_layer_scratch_floats(C.byref(L), B)) + B * H, np.float32)

    toks = rng.integers(0, V, size=(B, steps))
    for s in range(steps):
        want = ref(toks[:, s:s + 1], cache=cache)[:, 0]
        h = np.ascontiguousarray(W["model.embed_tokens"].weight[toks[:, s]], dtype=np.float32)
        pos = np.full(B, s, np.int32)
        lib.ref_layer_step(C.byref(L), h.ctypes.data, kc.ctypes.data, vc.ctypes.data, cap, pos.ctypes.data, B,
                           scratch.ctypes.data)
        logits = np.zeros((B, V), np.float32)
        lib.ref_head(fn.ctypes.data, hd.ctypes.data, V, H, 1e-5, h.ctypes.data, B, logits.ctypes.data, scratch.ctypes.data)
        # identical rounding points; differences are single bf16 flips from fp32-vs-fp64 accumulation
        err = np.abs(logits - want)
        assert np.mean(err > 0.05) < 0.02 and err.max() < 0.2, (s, err.max())
    assert np.mean(np.abs(bf16_bits_to_f32(kc) - cache[0].keys[:, :, :cap]) > 0.02) < 0.01
