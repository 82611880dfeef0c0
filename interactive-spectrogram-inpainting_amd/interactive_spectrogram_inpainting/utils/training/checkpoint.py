"""Training checkpoint record (reference utils/training/checkpoint.py:6-31): an ordered mapping with the
keys `model, epoch, validation_loss, validation_metrics, optimizer, scheduler, scaler, use_amp`, ready for
torch.save; `epoch` counts completed epochs."""
from __future__ import annotations

from collections import OrderedDict
from typing import Any, Dict, Mapping, Optional

StateDict = Mapping[str, Any]


class Checkpoint(OrderedDict):
    def __init__(self, model=None, epoch: int = 0, validation_loss: float = float("nan"),
                 validation_metrics: Optional[Dict[str, float]] = None, optimizer=None, scheduler=None, scaler=None):
        if model is None:      # pickle rebuilds an OrderedDict subclass as cls() and then re-inserts the items
            super().__init__()
            return

        def state(obj) -> Optional[StateDict]:
            return obj.state_dict() if obj is not None else None

        super().__init__(model=model.state_dict(), epoch=epoch, validation_loss=validation_loss,
                         validation_metrics=validation_metrics, optimizer=optimizer.state_dict(),
                         scheduler=state(scheduler), scaler=state(scaler), use_amp=scaler is not None)
