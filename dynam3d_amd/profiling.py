"""HIP-event timing of individual launches on the stream they are issued on (torch.cuda.Event records on the
current stream, which is the stream every kernel of this package is launched on)."""
from __future__ import annotations

from collections import defaultdict
from contextlib import contextmanager

import torch


class EventTimer:
    def __init__(self):
        self.pairs = defaultdict(list)
        self.enabled = False

    @contextmanager
    def range(self, name: str):
        if not self.enabled:
            yield
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        yield
        b.record()
        self.pairs[name].append((a, b))

    def summary(self):
        """name -> (count, mean ms); call after torch.cuda.synchronize()."""
        out = {}
        for k, v in self.pairs.items():
            ts = [a.elapsed_time(b) for a, b in v]
            out[k] = (len(ts), sum(ts) / max(len(ts), 1))
        return out

    def reset(self):
        self.pairs.clear()


TIMER = EventTimer()
