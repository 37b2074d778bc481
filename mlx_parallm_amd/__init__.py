"""mlx_parallm_amd -- MI355X-native batched decode engine behind the mlx_parallm generation API.

Drop-in for ONE path of misanthropic-ai/mlx_parallm: ``utils.load`` / ``generate_step`` /
``batch_generate`` / ``generate`` / ``stream_generate`` / ``batch_stream_generate_text`` /
``batch_generate_text`` and ``sample_utils.top_p_sampling``.  The arithmetic runs in
``csrc/libmi355_decode.so`` (hand-written HIP for gfx950); see DESIGN.md.
"""
__version__ = "0.1.0"
