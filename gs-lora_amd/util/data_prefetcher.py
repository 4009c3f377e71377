"""`data_prefetcher(loader, device, prefetch=True).next()` — the forget-batch feeder of the engines (API of the reference's
util/data_prefetcher.py:10-58: `.next()` yields `(samples, targets)` and `(None, None)` once the loader is exhausted).

MI355X-first implementation: a small ring of batches is kept in flight on a dedicated copy stream (a `hipStream_t` on ROCm). Host
batches are staged through reusable PINNED buffers so that the H2D copy is a true asynchronous DMA over PCIe; every ring slot carries
an event that (a) the consumer's stream waits on before using the batch and (b) guards the slot's pinned buffers against being
overwritten while their copy is still running. Batches that already live on the device pass through untouched."""
import collections

import torch


class data_prefetcher:
    DEPTH = 2      # batches in flight

    def __init__(self, loader, device, prefetch=True):
        self._source = iter(loader)
        self._device = torch.device(device)
        self._overlap = bool(prefetch) and self._device.type == "cuda" and torch.cuda.is_available()
        self._ring = collections.deque()
        self._slots = [dict(pinned={}, done=None) for _ in range(self.DEPTH + 1)]
        self._slot_no = 0
        self._copy_stream = torch.cuda.Stream(self._device) if self._overlap else None
        if self._overlap:
            for _ in range(self.DEPTH):
                self._launch_one()

    # ---- producer side -------------------------------------------------------------------------------------------------
    def _staged(self, slot, key, t):
        """Device copy of one tensor of the batch; host tensors go through the slot's pinned buffer."""
        if t.is_cuda:
            return t.to(self._device, non_blocking=True)
        buf = slot["pinned"].get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = slot["pinned"][key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        buf.copy_(t)
        return buf.to(self._device, non_blocking=True)

    def _launch_one(self):
        batch = next(self._source, None)
        if batch is None:
            self._ring.append(None)
            return
        slot = self._slots[self._slot_no % len(self._slots)]
        self._slot_no += 1
        if slot["done"] is not None:
            slot["done"].synchronize()          # the previous DMA out of this slot's pinned buffers has finished
        with torch.cuda.stream(self._copy_stream):
            moved = tuple(self._staged(slot, k, t) for k, t in enumerate(batch))
            slot["done"] = torch.cuda.Event()
            slot["done"].record(self._copy_stream)
        self._ring.append((moved, slot["done"]))

    # ---- consumer side -------------------------------------------------------------------------------------------------
    def next(self):
        if not self._overlap:
            batch = next(self._source, None)
            if batch is None:
                return None, None
            samples, targets = batch
            return samples.to(self._device, non_blocking=True), targets.to(self._device, non_blocking=True)
        entry = self._ring.popleft() if self._ring else None
        self._launch_one()
        if entry is None:
            return None, None
        (samples, targets), ready = entry
        consumer = torch.cuda.current_stream(self._device)
        consumer.wait_event(ready)
        for t in (samples, targets):
            t.record_stream(consumer)           # allocated under the copy stream, consumed on this one
        return samples, targets
