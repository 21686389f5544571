"""PCIe floor of the host path: time of pinned-host -> HBM copies of one tick's frame (65 536 x 160 int16 = 21 MB), alone,
split over two streams, and with a device kernel running beside them.  Prints one JSON line."""
import json, time, torch
n = 65536*160
host = torch.empty(n, dtype=torch.int16).pin_memory()
dev = torch.empty(n, dtype=torch.int16, device="cuda")
s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
out = {}
def timed(fn, reps=40):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0)/reps*1e3
def one():
    with torch.cuda.stream(s1):
        dev.copy_(host, non_blocking=True)
def two():
    h = n//2
    with torch.cuda.stream(s1):
        dev[:h].copy_(host[:h], non_blocking=True)
    with torch.cuda.stream(s2):
        dev[h:].copy_(host[h:], non_blocking=True)
out["h2d_one_stream_ms"] = timed(one)
out["h2d_two_streams_ms"] = timed(two)
out["gbps_one"] = n*2/out["h2d_one_stream_ms"]/1e6
out["gbps_two"] = n*2/out["h2d_two_streams_ms"]/1e6
print(json.dumps(out))
