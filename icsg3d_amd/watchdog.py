"""Hang protection for multi-process runs (new -- the reference is single-process).

A collective that never completes (a rank that died, a communicator that was never matched) blocks the surviving ranks
inside a HIP / RCCL wait for ever, and on a leased 8-GPU node that holds the lease until the driver's own time-out.
`StepWatchdog` is a daemon thread per rank: the rank calls `beat("where it is")` whenever it makes progress (every step,
every phase); when no beat arrives for `timeout` seconds the thread writes which phase the rank was in to stderr and ends
the process with `os._exit(3)` -- never a re-exec, and no clean-up that could block on the same hang.
torch.distributed.run then tears the other ranks down and exits non-zero, and icsg3d_amd.launcher relays that status.
The C-ABI calls run with the GIL released (ctypes), so the thread keeps running while the main thread is stuck in one.
"""
from __future__ import annotations

import os
import sys
import threading
import time


class StepWatchdog:
    def __init__(self, timeout=None, rank=None, exit_fn=os._exit, poll=None):
        """timeout: seconds without progress before the rank is ended (ICSG3D_WATCHDOG_S, default 120; <= 0 disables)."""
        if timeout is None:
            timeout = float(os.environ.get("ICSG3D_WATCHDOG_S", "120"))
        self.timeout = float(timeout)
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self._exit = exit_fn
        self._where = "start"
        self._last = time.monotonic()
        self._paused = False
        self._stop = threading.Event()
        self._poll = poll if poll is not None else max(0.05, min(1.0, self.timeout / 10.0))
        self._thread = None
        if self.timeout > 0:
            self._thread = threading.Thread(target=self._run, name="icsg3d-watchdog", daemon=True)
            self._thread.start()

    def beat(self, where=None):
        """Progress: restart the clock (and remember where the rank is for the message)."""
        if where is not None:
            self._where = where
        self._last = time.monotonic()
        self._paused = False

    def pause(self, where=None):
        """A phase that legitimately takes long on ONE rank with no collective in it (the CPU baseline subprocess)."""
        if where is not None:
            self._where = where
        self._paused = True

    def stop(self):
        self._stop.set()

    def _run(self):
        while not self._stop.wait(self._poll):
            if self._paused:
                continue
            idle = time.monotonic() - self._last
            if idle > self.timeout:
                try:
                    sys.stderr.write("[watchdog] rank %d: no progress for %.0f s in '%s' -- a collective or a step is hung; "
                                     "ending this rank (exit 3)\n" % (self.rank, idle, self._where))
                    sys.stderr.flush()
                finally:
                    self._exit(3)
                return


class NoWatchdog:
    """Same surface, does nothing (single-process runs without a communicator)."""
    def beat(self, where=None): pass
    def pause(self, where=None): pass
    def stop(self): pass
