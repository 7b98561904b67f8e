"""Which torch streams of one process run concurrently on MI355X?  (round 6: the throughput mode puts one batch per stream; two streams that
HIP maps onto the same hardware queue serialise.)  Creates N streams, runs a spin kernel of ~T ms on each pair together and prints the pair's
wall time relative to one spin alone: ~1.0 = concurrent, ~2.0 = same queue."""
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
streams = [torch.cuda.Stream(dev) for _ in range(N)]
ticks = int(2e8)


def spin(sts):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in sts:
        with torch.cuda.stream(s):
            torch.cuda._sleep(ticks)
    for s in sts:
        s.synchronize()
    return time.perf_counter() - t0


spin(streams[:1])
one = min(spin([s]) for s in streams[:2] for _ in range(2))
print(f"one spin: {one * 1e3:.2f} ms")
for i in range(N):
    row = []
    for j in range(N):
        row.append("  . " if i == j else f"{spin([streams[i], streams[j]]) / one:4.1f}")
    print(f"stream {i}: " + " ".join(row))
print("all together:", round(spin(streams) / one, 2))
for k in (2, 3, 4, 5, 6):
    print(f"first {k} together: {spin(streams[:k]) / one:.2f}")
