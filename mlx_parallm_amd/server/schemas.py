"""Wire schemas of the OpenAI-compatible server (mirror of ``mlx_parallm/server/schemas.py``).

Same model names, field names, defaults and bounds as the reference, because clients and the
reference's own tests (``tests/test_server_basic.py``) bind to them; the docstrings and grouping are
this repo's.  Reference lines are cited per class.
"""
from __future__ import annotations

import time
import uuid
from enum import Enum
from typing import Any, Dict, List, Literal, Optional, Union

from pydantic import BaseModel, ConfigDict, Field


def _now() -> int:
    return int(time.time())


# ---- model registry (schemas.py:8-74) ----
class ModelStatus(str, Enum):
    LOADED = "loaded"
    AVAILABLE_NOT_LOADED = "available_not_loaded"
    ERROR_LOADING = "error_loading"
    LOADING = "loading"


ModelKind = Literal["causal_lm", "embedding", "classifier", "reward", "general_nn"]


class ModelCard(BaseModel):
    id: str
    object: Literal["model"] = "model"
    created: int = Field(default_factory=_now)
    owned_by: str = "mlx_parallm"
    root: Optional[str] = None
    parent: Optional[str] = None
    status: ModelStatus = ModelStatus.AVAILABLE_NOT_LOADED
    type: Optional[ModelKind] = None
    path_or_hf_id: Optional[str] = None


class ModelList(BaseModel):
    object: Literal["list"] = "list"
    data: List[ModelCard] = Field(default_factory=list)


class InternalModelRecord(BaseModel):
    """Registry entry; the two ``*_instance`` fields never leave the process."""
    model_config = ConfigDict(arbitrary_types_allowed=True, protected_namespaces=())

    id: str
    path_or_hf_id: str
    model_type: Optional[ModelKind] = None
    status: ModelStatus = ModelStatus.AVAILABLE_NOT_LOADED
    created_timestamp: int = Field(default_factory=_now)
    owned_by: str = "mlx_parallm"
    adapter_path: Optional[str] = None
    model_instance: Optional[Any] = None
    tokenizer_instance: Optional[Any] = None

    def to_model_card(self) -> ModelCard:
        return ModelCard(id=self.id, created=self.created_timestamp, owned_by=self.owned_by, status=self.status,
                         type=self.model_type, path_or_hf_id=self.path_or_hf_id)


# ---- /v1/completions (schemas.py:80-118) ----
class CompletionUsage(BaseModel):
    prompt_tokens: int
    completion_tokens: int
    total_tokens: int


class CompletionChoice(BaseModel):
    text: str
    index: int = 0
    logprobs: Optional[Any] = None          # {tokens, token_logprobs, top_logprobs, text_offset}
    finish_reason: Optional[Literal["stop", "length"]] = "stop"


class CompletionResponse(BaseModel):
    id: str = Field(default_factory=lambda: f"cmpl-{uuid.uuid4().hex[:29]}")
    object: str = "text_completion"
    created: int = Field(default_factory=_now)
    model: str
    choices: List[CompletionChoice]
    usage: Optional[CompletionUsage] = None


class CompletionRequest(BaseModel):
    model: str
    prompt: str
    max_tokens: int = Field(100, ge=0)
    temperature: float = Field(0.0, ge=0.0, le=2.0)
    top_p: float = Field(1.0, ge=0.0, le=1.0)
    stream: Optional[bool] = False
    n: Optional[int] = 1
    logprobs: Optional[int] = None
    echo: Optional[bool] = False
    logit_bias: Optional[Dict[str, float]] = None


# ---- /v1/chat/completions (schemas.py:121-172) ----
class ChatMessage(BaseModel):
    role: str
    content: str
    name: Optional[str] = None


class ChatCompletionRequest(BaseModel):
    model: str
    messages: List[ChatMessage]
    temperature: Optional[float] = 0.7
    top_p: Optional[float] = 1.0
    n: Optional[int] = 1
    stream: Optional[bool] = False
    stop: Optional[Union[str, List[str]]] = None
    max_tokens: Optional[int] = None
    presence_penalty: Optional[float] = 0.0
    frequency_penalty: Optional[float] = 0.0
    logit_bias: Optional[Dict[str, float]] = None
    user: Optional[str] = None


class ChatCompletionChoice(BaseModel):
    index: int
    message: ChatMessage
    finish_reason: Optional[str] = "stop"


class ChatCompletionResponse(BaseModel):
    id: str = Field(default_factory=lambda: f"chatcmpl-{uuid.uuid4().hex[:28]}")
    object: str = "chat.completion"
    created: int = Field(default_factory=_now)
    model: str
    choices: List[ChatCompletionChoice]
    usage: CompletionUsage


class DeltaMessage(BaseModel):
    role: Optional[str] = None
    content: Optional[str] = None


class ChatCompletionStreamChoice(BaseModel):
    index: int
    delta: DeltaMessage
    finish_reason: Optional[str] = None


class ChatCompletionChunk(BaseModel):
    id: str
    object: str = "chat.completion.chunk"
    created: int = Field(default_factory=_now)
    model: str
    choices: List[ChatCompletionStreamChoice]


# ---- /v1/perplexity (schemas.py:180-190) ----
class PerplexityRequest(BaseModel):
    model: str
    text: str


class PerplexityResponse(BaseModel):
    model: str
    token_count: int
    avg_nll: float
    ppl: float
