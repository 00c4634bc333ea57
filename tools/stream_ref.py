"""What a plain streaming kernel achieves on this GPU for the byte counts of the model's products: torch's own
elementwise add (2 reads + 1 write) and copy, timed with events on cache-cold operands (development reference)."""
import json
import torch

dev = torch.device("cuda:0")
N = 1203042
res = {}
for name, w in (("N x 64", 64), ("N x 128", 128)):
    bufs = [(torch.randn(N, w, device=dev), torch.randn(N, w, device=dev), torch.empty(N, w, device=dev)) for _ in range(6)]
    for a, b, o in bufs:
        torch.add(a, b, out=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        for a, b, o in bufs:
            torch.add(a, b, out=o)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 18 * 1e3
    res["add " + name] = {"us": round(us, 1), "GBps": round(3 * N * w * 4 / us / 1e3, 1)}
    e0.record()
    for _ in range(3):
        for a, b, o in bufs:
            o.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 18 * 1e3
    res["copy " + name] = {"us": round(us, 1), "GBps": round(2 * N * w * 4 / us / 1e3, 1)}
print(json.dumps(res))
