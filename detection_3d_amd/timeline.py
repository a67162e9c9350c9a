"""Marks for a per-stage timeline of one pass (scripts/lane_timeline.py).  `MARKS` is None unless a probe sets it to a
list; then `mark` appends (label, level, timed event recorded on `stream` or None, host clock).  Off, a mark is one
global lookup."""
import time

import torch

MARKS = None


def mark(label, level=-1, stream=None, host=False):
    if MARKS is None:
        return
    if host:
        MARKS.append((label, level, None, time.perf_counter()))
        return
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(stream if stream is not None else torch.cuda.current_stream())
    MARKS.append((label, level, ev, time.perf_counter()))
