#!/usr/bin/env python3
"""Which hardware queue does each HIP stream of a data-parallel process get?  Run under AMD_LOG_LEVEL=4 and grep `acquireQueue`:
    AMD_LOG_LEVEL=4 python tools/queue_probe.py [--streams-first] 2>&1 | grep -i "acquireQueue\|MARK"
MARK lines (stderr) bracket the phases: streams created, process group created, first collective."""
import os
import sys
import torch
import torch.distributed as dist

first = "--streams-first" in sys.argv
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.zeros(1, device=dev)


def mark(s):
    sys.stderr.write(f"MARK {s}\n")
    sys.stderr.flush()


def mk():
    out = []
    for i in range(4):
        mark(f"creating stream {i}")
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            torch.zeros(1, device=dev)
        torch.cuda.synchronize()
        out.append(st)
    return out


if first:
    ss = mk()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
mark("init_process_group")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
mark("first collective")
t = torch.ones(1 << 20, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
if not first:
    ss = mk()
mark("done")
dist.destroy_process_group()
