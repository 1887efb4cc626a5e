"""Loader threads in front of the GPU step -- the counterpart of Keras' `fit_generator(workers=4,
use_multiprocessing=False)` (/root/reference/unet/unet.py:370-377): `workers` Python threads evaluate
`sequence[i]` (np.load of every grid of the batch, /root/reference/unet/data.py:64-100) plus a `prepare`
hook (float32 cast, one-hot -> uint8 class ids) and hand the ready batches over IN ORDER through a bounded
window, so disk I/O and host-side label work overlap with the previous step on the device (ctypes releases
the GIL for the duration of a C-ABI call)."""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor


def prefetched(sequence, workers=4, max_queue_size=10, prepare=None):
    """Yield prepare(sequence[i]) for i in range(len(sequence)), in order, computed up to `max_queue_size`
    items ahead by `workers` threads.  workers <= 0: plain synchronous iteration."""
    n = len(sequence)

    def job(i):
        item = sequence[i]
        return prepare(item) if prepare is not None else item

    if workers <= 0 or n <= 1:
        for i in range(n):
            yield job(i)
        return
    window = max(1, int(max_queue_size))
    with ThreadPoolExecutor(max_workers=int(workers), thread_name_prefix="icsg3d-loader") as pool:
        pending = []
        nxt = 0
        while nxt < n and len(pending) < window:
            pending.append(pool.submit(job, nxt)); nxt += 1
        while pending:
            item = pending.pop(0).result()       # re-raises a loader exception in the training thread
            if nxt < n:
                pending.append(pool.submit(job, nxt)); nxt += 1
            yield item
