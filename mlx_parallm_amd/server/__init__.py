"""OpenAI-compatible server over the MI355X decode engine (mirror of ``mlx_parallm/server``)."""
