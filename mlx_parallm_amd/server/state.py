"""Process-wide model registry and the lock weight updates take (``mlx_parallm/server/state.py:1-16``)."""
from __future__ import annotations

from threading import RLock
from typing import Dict, Optional

from .schemas import InternalModelRecord, ModelStatus

model_registry: Dict[str, InternalModelRecord] = {}
weight_update_lock = RLock()


def get_active_record() -> Optional[InternalModelRecord]:
    """First LOADED record that actually holds a model."""
    for rec in model_registry.values():
        if rec.status == ModelStatus.LOADED and rec.model_instance is not None:
            return rec
    return None
