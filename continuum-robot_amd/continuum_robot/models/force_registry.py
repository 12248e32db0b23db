"""ForceRegistry / InputRegistry: the functional composition layer of the drop-in API.

Reference: src/continuum_robot/models/force_registry.py (ForceRegistry :6-89, InputRegistry :92-173).
Semantics kept: registration silently drops components that are disabled at that moment; the
aggregate re-checks is_enabled() on every call, so toggling a registered component takes effect
immediately; getters hand out copies of the internal lists.
"""
from typing import Callable, List

import numpy as np

from .abstractions import AbstractForce, AbstractInputHandler


class _Registry:
    def __init__(self):
        self._items = []

    def register(self, item) -> None:
        if item.is_enabled():
            self._items.append(item)

    def unregister(self, item) -> bool:
        if item in self._items:
            self._items.remove(item)
            return True
        return False

    def clear(self) -> None:
        self._items.clear()

    def __len__(self) -> int:
        return len(self._items)

    def __contains__(self, item) -> bool:
        return item in self._items


class ForceRegistry(_Registry):
    @property
    def _forces(self):
        return self._items

    def get_registered_forces(self) -> List[AbstractForce]:
        return list(self._items)

    def create_aggregated_function(self) -> Callable:
        def aggregate_forces(x: np.ndarray, t: float = 0.0) -> np.ndarray:
            total = None
            for force in self._items:
                if not force.is_enabled():
                    continue
                part = force.compute_forces(x, t)
                if total is None:
                    total = part.copy()
                else:
                    total += part
            if total is None:
                total = np.zeros(len(x) // 2)
            return total

        return aggregate_forces


class InputRegistry(_Registry):
    @property
    def _input_handlers(self):
        return self._items

    def get_registered_handlers(self) -> List[AbstractInputHandler]:
        return list(self._items)

    def create_aggregated_function(self) -> Callable:
        def aggregate_input_processing(x: np.ndarray, u: np.ndarray, t: float = 0.0) -> np.ndarray:
            total = u.copy()
            for handler in self._items:
                if handler.is_enabled():
                    total += handler.compute_input(x, u, t)  # handlers see the ORIGINAL input
            return total

        return aggregate_input_processing
