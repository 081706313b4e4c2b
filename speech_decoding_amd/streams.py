"""HIP stream policy of the training loop.

The engine runs the backward's weight-gradient GEMMs (and the per-step operand packing) on side streams of NORMAL priority,
beside the chain the caller drives: forward, loss, the data-gradient chain, optimiser.  That chain is the critical path
(its stream is busy 6.8 of a 7.0 ms step at config 2; the weight-gradient stream 3.5 ms inside the backward window), so the
dispatcher should hand a freed CU slot to ITS next workgroup first: the training loop makes a HIGH-priority stream current
for its whole life.  Measured on one box, three alternations (tools/step_series.py, SDA_MAIN_PRIO): 7.13-7.18 ms on the
default stream, 6.96-7.03 on a priority -1 stream, 7.07-7.16 on an explicit priority 0 stream."""
from __future__ import annotations

import torch


def use_training_stream(device=None, priority: int = -1) -> torch.cuda.Stream:
    """Make a stream of `priority` (-1 = high on ROCm: torch.cuda.Stream.priority_range() == (0, -1)) the current stream of
    `device`; everything queued on the previously current stream is waited for first.  Returns the stream (the caller
    keeps it current: torch.cuda.set_stream is per thread and lasts until changed)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    prev = torch.cuda.current_stream(dev)
    s = torch.cuda.Stream(device=dev, priority=priority)
    s.wait_stream(prev)
    torch.cuda.set_stream(s)
    return s
