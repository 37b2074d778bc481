"""Model families on the hot path: llama (incl. mistral) and qwen3 (``utils._get_classes``)."""
