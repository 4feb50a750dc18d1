"""Host turnaround of reading a 4-byte device flag behind a ~1 ms kernel chain: Tensor.item() against a pinned copy + event wait.
python tools/probes/flag_read_latency.py"""
import time
import torch

dev = torch.device("cuda", 0)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
pinned = torch.zeros(1, dtype=torch.int32).pin_memory()
x = torch.rand(1 << 22, device=dev)


def chain():
    y = x
    for _ in range(20):
        y = y * 1.0001
    return y


def run(reader, n=200):
    for _ in range(10):
        chain(); reader()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    gpu = 0.0
    for _ in range(n):
        e0.record(); chain(); e1.record()
        reader()
        gpu += e0.elapsed_time(e1)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    return wall, gpu / n


def r_item():
    return bool(flag.item())


def r_pinned_event():
    pinned.copy_(flag, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    ev.synchronize()
    return bool(pinned[0])


def r_pinned_stream():
    pinned.copy_(flag, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return bool(pinned[0])


def r_pinned_spin():
    pinned.copy_(flag, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    while not ev.query():
        pass
    return bool(pinned[0])


for name, r in (("Tensor.item()", r_item), ("pinned copy + event.synchronize()", r_pinned_event), ("pinned copy + stream.synchronize()", r_pinned_stream),
                ("pinned copy + event.query() spin", r_pinned_spin)):
    wall, gpu = run(r)
    print(f"{name:40s} wall {wall:7.3f} ms per iteration, kernels {gpu:7.3f} ms, host turnaround {1e3 * (wall - gpu):7.1f} us", flush=True)
