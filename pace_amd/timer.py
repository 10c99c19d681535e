"""Named wall-clock timers with the interface the reference driver hands down to the dycore
(``ndsl.performance.timer.Timer``: ``clock(name)`` context manager, ``times`` / ``hits`` per name, ``reset``)
[REF driver/pace/driver/driver.py:630-643 (``with timer.clock("mainloop")``), tests/main/driver/test_driver.py:77-121
(the timer names ``mainloop`` / ``DynCore`` / ``TracerAdvection`` / ``Remapping``)].  ``sync`` is called before a clock starts and
before it stops (the device synchronisation of a GPU backend: the reference's ``device_sync``)."""
from __future__ import annotations

import contextlib
import time
from typing import Callable, Dict, Optional


class Timer:
    def __init__(self, enabled: bool = True, sync: Optional[Callable[[], None]] = None, sync_names=None):
        self._enabled = enabled
        self._sync = sync
        self._sync_names = None if sync_names is None else set(sync_names)  # None: every clock synchronises
        self._clock_starts: Dict[str, float] = {}
        self._accumulated: Dict[str, float] = {}
        self._hits: Dict[str, int] = {}

    def start(self, name: str):
        if self._enabled:
            if name in self._clock_starts:
                raise ValueError(f"clock already started for '{name}'")
            if self._sync and (self._sync_names is None or name in self._sync_names):
                self._sync()
            self._clock_starts[name] = time.perf_counter()

    def stop(self, name: str):
        if self._enabled:
            if self._sync and (self._sync_names is None or name in self._sync_names):
                self._sync()
            dt = time.perf_counter() - self._clock_starts.pop(name)
            self._accumulated[name] = self._accumulated.get(name, 0.0) + dt
            self._hits[name] = self._hits.get(name, 0) + 1

    @contextlib.contextmanager
    def clock(self, name: str):
        self.start(name)
        try:
            yield
        finally:
            self.stop(name)

    @property
    def times(self) -> Dict[str, float]:
        if self._clock_starts:
            raise RuntimeError("Cannot retrieve times while clocks are still going: " + ", ".join(self._clock_starts))
        return dict(self._accumulated)

    @property
    def hits(self) -> Dict[str, int]:
        return dict(self._hits)

    def reset(self):
        self._accumulated.clear()
        self._hits.clear()

    @property
    def enabled(self) -> bool:
        return self._enabled


class NullTimer(Timer):
    def __init__(self):
        super().__init__(enabled=False)
