"""Forget-batch prefetcher on a side HIP stream (reference util/data_prefetcher.py:10-58).
torch.cuda.Stream *is* a hipStream_t on ROCm; record_stream keeps the caching allocator honest."""
import torch


def to_cuda(samples, targets, device):
    return samples.to(device, non_blocking=True), targets.to(device, non_blocking=True)


class data_prefetcher:
    def __init__(self, loader, device, prefetch=True):
        self.loader = iter(loader)
        self.prefetch = prefetch and torch.cuda.is_available()
        self.device = device
        if self.prefetch:
            self.stream = torch.cuda.Stream()
            self.preload()

    def _fetch(self):
        try:
            return next(self.loader)
        except StopIteration:
            return None, None

    def preload(self):
        self.next_samples, self.next_targets = self._fetch()
        if self.next_samples is None:
            return
        with torch.cuda.stream(self.stream):
            self.next_samples, self.next_targets = to_cuda(self.next_samples, self.next_targets, self.device)

    def next(self):
        if not self.prefetch:
            samples, targets = self._fetch()
            if samples is not None:
                samples, targets = to_cuda(samples, targets, self.device)
            return samples, targets
        torch.cuda.current_stream().wait_stream(self.stream)
        samples, targets = self.next_samples, self.next_targets
        for t in (samples, targets):
            if t is not None and t.is_cuda:
                t.record_stream(torch.cuda.current_stream())
        self.preload()
        return samples, targets
