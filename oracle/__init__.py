"""CPU oracle for the mlx_parallm batched-decode hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain NumPy (and, under ``oracle/c``, plain C) restatement of the
reference algorithm on the path BASELINE.json's north_star names:
``mlx_parallm.utils.generate_step`` / ``batch_generate`` / ``sample_utils.top_p_sampling``
and the llama / qwen3 model forward + KV caches underneath them.  Every function cites
the reference ``file:line`` it follows (paths are relative to the reference checkout).

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the *checker* only.  The product package
(``mlx_parallm_amd``) never imports it and fails loudly when the HIP library is missing.

PARITY UNPINNED.  The arithmetic of the reference lives in third-party wheels that are not
in the reference tree and not installable here (``mlx==0.25.2``, ``mlx-lm==0.24.1``;
``import mlx`` raises ModuleNotFoundError), and the reference's own tests hold no golden
vectors / known-answer values for this path (they assert HTTP status codes and key
presence only).  So this oracle is pinned by (1) the reference *source text* it restates,
(2) an independent second implementation of the same decoder maths (HuggingFace
``transformers`` Llama/Mistral/Qwen3 in fp32, see tests/test_oracle_vs_hf.py) and (3) the
C restatement under ``oracle/c`` -- NOT by outputs of MLX itself.  MLX op semantics used
here (RMSNorm / RoPE / SDPA / affine quantisation / LoRALinear rounding points) are the
publicly documented ones restated in SURVEY.md App. A and are the build's definition.
"""
