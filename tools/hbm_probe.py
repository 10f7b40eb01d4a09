"""Practical HBM roof of this box for streaming kernels: torch's own elementwise kernels (copy 1R:1W, add 2R:1W, sum 1R:0W)
at the working-set sizes of the kernels priced against the 8 TB/s figure (warp: ~0.3 GB; resblock tail: 2.8-3.7 GB)."""
import json, torch

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for mb in (96, 320, 1024, 3072):
    n = mb * (1 << 20) // 2 // 3                     # three bf16 arrays of mb/3 MB each
    a = torch.randn(n, device="cuda").to(torch.bfloat16); b = a.clone(); c = torch.empty_like(a)
    ms_copy = t(lambda: c.copy_(a)); ms_add = t(lambda: torch.add(a, b, out=c)); ms_sum = t(lambda: a.float().sum() if False else torch.sum(a, dtype=torch.float32))
    by = n * 2
    print(json.dumps({"arrays_MB": round(by / 1e6, 1), "copy_TBps": round(2 * by / ms_copy / 1e9, 3), "add_TBps": round(3 * by / ms_add / 1e9, 3),
                      "sum_TBps": round(by / ms_sum / 1e9, 3), "copy_ms": round(ms_copy, 4), "add_ms": round(ms_add, 4)}))
