"""H2D / D2H rate of pinned buffers allocated while the process is bound to the CPUs of each NUMA node (first touch
decides where the pages live).  On the two-socket MI355X hosts one node copies to the card at 57 GB/s and the other at
22-28 GB/s (D2H 57 either way) -- and not always the node sysfs names for the card.  Binding the library's pinned
buffers (by sysfs, then by a measured probe) was tried and dropped: the streamed rate of bench.py moved between 530 and
1020 Mreads/s from box to box with or without it.  Run on the GPU box."""
import glob
import os
import time

import torch


def cpus_of(node):
    txt = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
    out = []
    for part in txt.split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
print("numa nodes:", nodes, "| gpu numa:", [open(p).read().strip() for p in glob.glob("/sys/class/drm/card*/device/numa_node")][:8], flush=True)
allowed = sorted(os.sched_getaffinity(0))
print("allowed cpus:", len(allowed), allowed[:4], "...", allowed[-4:], flush=True)
dev = torch.device("cuda:0")
n = 1 << 30
d = torch.empty(n, dtype=torch.uint8, device=dev)
for node in nodes:
    cp = [c for c in cpus_of(node) if c in allowed]
    if not cp:
        print("node", node, "no allowed cpus"); continue
    os.sched_setaffinity(0, cp)
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    h.fill_(1)
    for direction in ("h2d", "d2h"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            if direction == "h2d":
                d.copy_(h, non_blocking=True)
            else:
                h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("node %d (%d cpus) %s: %.1f GB/s" % (node, len(cp), direction, 5 * n / dt / 1e9), flush=True)
    del h
os.sched_setaffinity(0, allowed)

