#!/usr/bin/env python3
"""Achievable HBM write bandwidth on this GPU for the observation tier's traffic shape: a plain fill of ~1 GB (torch
zero_ = a vectorised store kernel), timed with events.  The observation kernels are pure streaming stores; this is their
practical ceiling, next to the 8 TB/s datasheet figure."""
import json
import torch
n = 65536 * 15096
x = torch.empty(n, dtype=torch.int8, device="cuda")
for _ in range(3):
    x.zero_()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    x.zero_()
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) * 1e3 / 20
y = torch.empty(n // 8, dtype=torch.int64, device="cuda")
a.record()
for _ in range(20):
    y.fill_(0x0101010101010101)
b.record()
torch.cuda.synchronize()
us2 = a.elapsed_time(b) * 1e3 / 20
print(json.dumps(dict(bytes=n, zero_us=us, zero_TBps=n / us / 1e6, fill64_us=us2, fill64_TBps=n / us2 / 1e6)))
