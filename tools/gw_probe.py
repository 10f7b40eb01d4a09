"""grid_warp variants timing (direct C-ABI calls, HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflowdiffusion_amd import _lib as L
lib = L.lib()
def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, H = 16, 440
for W in (1024, 1022):
    torch.manual_seed(0)
    img = torch.rand(B, 3, H, W, device="cuda")
    flow = torch.nn.functional.avg_pool2d(torch.randn(B, 2, H, W, device="cuda") * 72, 9, 1, 4).clamp(-20, 20)
    out, mask = torch.empty_like(img), torch.empty_like(img)
    for label, m in (("mask", mask), ("nomask", None)):
        us = timed(lambda: L.check(lib.ofd_grid_warp_fwd(L.ptr(img), L.ptr(flow), L.ptr(out), L.ptr(m), B, 3, H, W, L.stream())))
        by = (44.0 if m is not None else 32.0) * B * H * W
        print(f"W={W} {label}: {us:.1f} us  {by / us / 1e6:.2f} TB/s")
    zf = torch.zeros_like(flow)
    us = timed(lambda: L.check(lib.ofd_grid_warp_fwd(L.ptr(img), L.ptr(zf), L.ptr(out), L.ptr(mask), B, 3, H, W, L.stream())))
    print(f"W={W} zero flow: {us:.1f} us")
    us = timed(lambda: out.copy_(img))
    print(f"W={W} torch copy (24 B/px): {us:.1f} us  {24.0 * B * H * W / us / 1e6:.2f} TB/s")
