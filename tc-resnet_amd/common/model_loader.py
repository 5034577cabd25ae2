"""Checkpoint -> model variables, with the reference loader's knobs (common/model_loader.py:10-165): scope include /
exclude prefixes, `--ignore_missing_vars`, and `--use_ema` (trainables are read from their
`<name>/ExponentialMovingAverage` shadow, what ema.variables_to_restore() maps)."""
from __future__ import annotations

import logging
from typing import Callable, Dict, List, Optional

import numpy as np

from . import tf_bundle

EMA_SUFFIX = "/ExponentialMovingAverage"


def _split_strip(scopes: str) -> List[str]:
    return [s.strip() for s in (scopes or "").split(",") if s.strip()]


class Ckpt:
    def __init__(self, engine, include_scopes: str = "", exclude_scopes: str = "", ignore_missing_vars: bool = False,
                 use_ema: bool = False, ema_decay: Optional[float] = None, logger=None):
        if use_ema and ema_decay is None:
            raise ValueError("ema_decay undefined")
        self.engine = engine
        self.inclusions, self.exclusions = _split_strip(include_scopes), _split_strip(exclude_scopes)
        self.ignore_missing_vars, self.use_ema = bool(ignore_missing_vars), bool(use_ema)
        self.logger = logger or logging.getLogger("Ckpt Loader")

    def _selected(self, name: str) -> bool:
        if self.inclusions and not any(name.startswith(p) for p in self.inclusions):
            return False
        return not any(name.startswith(p) for p in self.exclusions)

    def variables_to_restore(self, extra: Optional[Dict[str, Callable]] = None) -> Dict[str, str]:
        """checkpoint name -> variable name, after the scope filters."""
        out = {}
        for name, ti in self.engine.tensors.items():
            if self._selected(name):
                out[name + EMA_SUFFIX if (self.use_ema and ti.arena == 0) else name] = name
        for name in (extra or {}):
            if self._selected(name):
                out[name] = name
        return out

    def load(self, checkpoint_stempath: str, extra: Optional[Dict[str, Callable[[np.ndarray], None]]] = None) -> List[str]:
        """Assigns every selected variable from the checkpoint.  `extra`: further variables of the caller (global_step,
        optimiser slots) as name -> setter.  Returns the names that were restored."""
        reader = tf_bundle.CheckpointReader(str(checkpoint_stempath))
        sd, restored = {}, []
        for ckpt_name, var in self.variables_to_restore(extra).items():
            if not reader.has_tensor(ckpt_name):
                msg = f"Checkpoint is missing variable [{ckpt_name}]"
                if self.ignore_missing_vars:
                    self.logger.warning(msg)
                    continue
                raise ValueError(msg)
            value = reader.get_tensor(ckpt_name)
            if extra and var in extra:
                extra[var](value)
            else:
                want = self.engine.tf_shape(var)        # the shape the reference's graph declares (a KWS matmul weight is [K, N])
                if tuple(value.shape) != want:
                    raise ValueError(f"Total size of new array must be unchanged for {ckpt_name} "
                                     f"lh_shape: [{value.shape}], rh_shape: [{want}]")
                sd[var] = value
            restored.append(var)
        self.engine.load_state_dict(sd, strict=False)
        self.logger.info("Restore from %s (%d variables)", checkpoint_stempath, len(restored))
        return restored
