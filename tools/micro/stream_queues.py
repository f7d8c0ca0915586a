#!/usr/bin/env python3
"""Which HSA queue does each torch stream land on?  Run under `rocprofv3 --kernel-trace`: each stream launches a fill
of a distinct size; the trace's Queue_Id per Grid_Size tells the mapping (ROCclr hands out at most GPU_MAX_HW_QUEUES
queues per priority and shares them between streams)."""
import sys
import torch
torch.cuda.set_device(0)
x = torch.zeros(1 << 20, device="cuda")
torch.cuda.synchronize()
streams = [("null", torch.cuda.current_stream())]
for i in range(6):
    streams.append((f"norm{i}", torch.cuda.Stream()))
for i in range(6):
    streams.append((f"high{i}", torch.cuda.Stream(priority=-1)))
for k, (name, s) in enumerate(streams):
    with torch.cuda.stream(s):
        n = 256 * 64 * (k + 1)
        x[:n].fill_(1.0)
    print(name, "fill elements", n, "stream id", s.stream_id, flush=True)
torch.cuda.synchronize()
