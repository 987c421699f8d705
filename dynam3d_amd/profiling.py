"""HIP-event timing of individual launches on the stream they are issued on (torch.cuda.Event records on the
current stream, which is the stream every kernel of this package is launched on)."""
from __future__ import annotations

from collections import defaultdict
from contextlib import contextmanager

import torch


class EventTimer:
    def __init__(self):
        self.pairs = defaultdict(list)
        self.meta = defaultdict(list)
        self.enabled = False

    @contextmanager
    def range(self, name: str, **meta):
        """`meta`: numbers that belong to THIS launch (e.g. rows=real tokens), kept beside its event pair (`records`)."""
        if not self.enabled:
            yield
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        yield
        b.record()
        self.pairs[name].append((a, b))
        if meta:
            self.meta[name].append(meta)

    def summary(self):
        """name -> (count, mean ms); call after torch.cuda.synchronize()."""
        out = {}
        for k, v in self.pairs.items():
            ts = [a.elapsed_time(b) for a, b in v]
            out[k] = (len(ts), sum(ts) / max(len(ts), 1))
        return out

    def records(self, name: str):
        """[(ms, meta dict)] per timed launch of `name`, in issue order; call after torch.cuda.synchronize()."""
        ts = [a.elapsed_time(b) for a, b in self.pairs.get(name, [])]
        ms = self.meta.get(name, [])
        return [(t, ms[i] if i < len(ms) else {}) for i, t in enumerate(ts)]

    def reset(self):
        self.pairs.clear()
        self.meta.clear()


TIMER = EventTimer()
