"""Optional event markers on the launch stream ("hipEvent/torch.cuda.Event around the region").

The product path calls :func:`mark` at its stage boundaries and :func:`span` around the hand-written kernels it launches;
both are no-ops (one ``is None`` test) unless a recorder is installed.  ``bench.py`` installs one inside its timed region so
that the numbers it reports are taken around the real ``nerfdet.forward_test``, not around a copy of it."""
from __future__ import annotations

from typing import Callable, Optional

import torch

recorder: Optional["Recorder"] = None


class Recorder:
    """Collects (name, start_event, end_event, info) for spans and (name, event) for stage marks; events are recorded on
    torch's current stream, which is the stream the C ABI launches on."""

    def __init__(self, sample: Callable[[str], bool] = lambda name: True):
        self.spans, self.marks, self.sample = [], [], sample

    @staticmethod
    def _event():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def mark(self, name: str):
        self.marks.append((name, self._event()))

    def span(self, name: str, thunk, info):
        if not self.sample(name):
            return thunk()
        e0 = self._event()
        r = thunk()
        self.spans.append((name, e0, self._event(), info))
        return r

    def stage_ms(self):
        """Average elapsed ms between consecutive marks, keyed by the later mark's name ('begin' opens a step)."""
        acc = {}
        for (n0, e0), (n1, e1) in zip(self.marks, self.marks[1:]):
            if n1 == "begin":
                continue
            acc.setdefault(n1, []).append(e0.elapsed_time(e1))
        return {k: sum(v) / len(v) for k, v in acc.items()}

    def span_ms(self):
        """name -> list of (ms, info)."""
        out = {}
        for name, e0, e1, info in self.spans:
            out.setdefault(name, []).append((e0.elapsed_time(e1), info))
        return out


def mark(name: str):
    if recorder is not None:
        recorder.mark(name)


def span(name: str, thunk, **info):
    return thunk() if recorder is None else recorder.span(name, thunk, info)
