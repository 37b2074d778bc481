"""``python -m mlx_parallm_amd.cli --model-path <dir> [...]``: start the server.

Same flags and defaults as the reference's ``ServerCLIArgs`` (``mlx_parallm/cli.py:15-32``); parsed with
argparse (pydantic_cli is not part of this environment).  ``--diverse-mode`` accepts both the bare flag
and ``--diverse-mode true|false`` (the form the reference's test helper passes, ``tests/helpers.py:124``).
Extra flag of this build: ``--device`` (HIP device index of the engine).
"""
from __future__ import annotations

import argparse
import logging
from typing import Optional, Sequence

from .server.main import ServerConfig, _truthy

logging.basicConfig(level=logging.INFO)
logger = logging.getLogger(__name__)

# read by mlx_parallm_amd.server.main.create_app() when the app is built without an explicit config
current_server_args: Optional[ServerConfig] = None


def _bool_arg(v: Optional[str]) -> bool:
    if v is None:
        return True
    b = _truthy(v)
    if b is None:
        raise argparse.ArgumentTypeError(f"expected true/false, got {v!r}")
    return b


def build_parser() -> argparse.ArgumentParser:
    d = ServerConfig()
    p = argparse.ArgumentParser(prog="mlx_parallm_amd_serve", description="mlx_parallm-compatible server on MI355X")
    p.add_argument("--model-path", required=True, help="Path of the base model directory to load.")
    p.add_argument("--host", default=d.host)
    p.add_argument("--port", type=int, default=d.port)
    p.add_argument("--lora-path", default=None, help="Optional LoRA adapter directory to apply at start-up.")
    p.add_argument("--max-batch-size", type=int, default=d.max_batch_size)
    p.add_argument("--batch-timeout", type=float, default=d.batch_timeout, help="Batching window in seconds.")
    p.add_argument("--request-timeout-seconds", type=float, default=d.request_timeout_seconds)
    p.add_argument("--max-concurrent-streams", type=int, default=d.max_concurrent_streams)
    p.add_argument("--scheduler", default=d.scheduler, help="'default' or 'continuous'.")
    p.add_argument("--diverse-mode", nargs="?", const=True, default=False, type=_bool_arg,
                   help="Disable prompt de-duplication and prefix sharing.")
    p.add_argument("--max-context-length", type=int, default=d.max_context_length)
    p.add_argument("--device", type=int, default=d.device, help="HIP device index.")
    p.add_argument("--devices", default=None, help="Comma-separated HIP device indices: one model replica per GPU, "
                   "requests spread over them (implies --scheduler continuous).")
    p.add_argument("--chunk-tokens", type=int, default=d.chunk_tokens,
                   help="continuous scheduler: prompt tokens per step that enter the cache inside the live rows' decode steps (0: off).")
    p.add_argument("--kv-blocks", type=int, default=None, help="continuous scheduler: 64-token blocks of the paged KV arena "
                   "(default: every slot at full length, capped at 80 %% of the free device memory).")
    p.add_argument("--no-prefix-cache", action="store_true", help="continuous scheduler: do not reuse the KV blocks of a common prompt prefix.")
    p.add_argument("--version", action="version", version="0.2.0")
    return p


def parse_args(argv: Optional[Sequence[str]] = None) -> ServerConfig:
    ns = build_parser().parse_args(argv)
    return ServerConfig(model_path=ns.model_path, host=ns.host, port=ns.port, lora_path=ns.lora_path,
                        max_batch_size=ns.max_batch_size, batch_timeout=ns.batch_timeout,
                        request_timeout_seconds=ns.request_timeout_seconds,
                        max_concurrent_streams=ns.max_concurrent_streams, scheduler=ns.scheduler,
                        diverse_mode=bool(ns.diverse_mode), max_context_length=ns.max_context_length, device=ns.device,
                        devices=[int(x) for x in ns.devices.split(",")] if ns.devices else None,
                        chunk_tokens=ns.chunk_tokens, kv_blocks=ns.kv_blocks, prefix_cache=not ns.no_prefix_cache)


def cli_runner(argv: Optional[Sequence[str]] = None) -> None:
    import uvicorn

    from .server.main import create_app

    global current_server_args
    current_server_args = parse_args(argv)
    logger.info("Starting server with initial model: %s", current_server_args.model_path)
    logger.info("Server will listen on %s:%d", current_server_args.host, current_server_args.port)
    uvicorn.run(create_app(current_server_args), host=current_server_args.host, port=current_server_args.port,
                log_level="info")


if __name__ == "__main__":
    cli_runner()
