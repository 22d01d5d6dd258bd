"""Achievable HBM bandwidth of the box, to put beside the 8 TB/s peak: device-to-device copy and fill of framebuffer-sized
buffers, timed with events on the current stream (torch is used only as the allocator / launcher here)."""
import torch
dev = torch.device("cuda:0")
for mib in (768, 3072):
    n = mib * 1024 * 1024
    a = torch.empty(n, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
    for name, fn, traffic in (("copy (read + write)", lambda: b.copy_(a), 2 * n), ("fill (write only)", lambda: b.fill_(7), n)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{mib:5d} MiB {name:20s}: {ms:.3f} ms  = {traffic / ms / 1e6:7.0f} GB/s of HBM traffic")
