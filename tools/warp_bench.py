"""Flow-warp kernels at BASELINE size (B=16, 440x1024): achieved HBM GB/s against the algorithmic
bytes of SURVEY 8d (forward splat 40 B/px with 3 image channels + weight; grid_sample 32 B/px,
44 with the mask), plus the CPU oracle on a bounded sample.  Prints one JSON line per kernel."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PEAK = 8000.0


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    import opticalflowdiffusion_amd as m
    from opticalflowdiffusion_amd.softsplat import splat_forward
    B, H, W = 16, 440, 1024
    torch.manual_seed(0)
    px = B * H * W
    img4 = torch.rand(B, 4, H, W, device="cuda")
    img3 = img4[:, :3].contiguous()
    smooth = torch.nn.functional.avg_pool2d(torch.randn(B, 2, H, W, device="cuda") * 8 * 9, 9, 1, 4)   # box-smoothed N(0,8^2)-like
    cases = {"smooth": smooth.clamp(-20, 20), "integer": torch.round(smooth.clamp(-20, 20)), "uniform20": (torch.rand(B, 2, H, W, device="cuda") * 2 - 1) * 20}
    for name, flow in cases.items():
        ms = timed(lambda: splat_forward(img4, flow))
        by = 40.0 * px
        print(json.dumps({"kernel": "splat_fwd (softsplat_out, C=3+1)", "flow": name, "ms": ms, "algorithmic_bytes": by,
                          "GBps": by / ms / 1e6, "frac_of_8TBps": by / ms / 1e6 / PEAK}))
        ms = timed(lambda: m.warp(None, img3, flow, mode="backward"))
        by = 44.0 * px
        print(json.dumps({"kernel": "grid_warp_fwd (grid_sample x2 + mask, C=3)", "flow": name, "ms": ms, "algorithmic_bytes": by,
                          "GBps": by / ms / 1e6, "frac_of_8TBps": by / ms / 1e6 / PEAK}))
        ms = timed(lambda: m.warp(img3, None, flow, mode="forward"))
        print(json.dumps({"kernel": "warp(mode=forward) wrapper: prep + splat + holes", "flow": name, "ms": ms}))
    # backward kernels
    flow = cases["smooth"]
    from opticalflowdiffusion_amd._lib import lib, check, ptr, stream
    g = torch.rand(B, 4, H, W, device="cuda")
    gi, gf = torch.empty_like(img4), torch.empty_like(flow)
    ms = timed(lambda: check(lib().ofd_splat_bwd_in(ptr(flow), ptr(g), ptr(gi), B, 4, H, W, 1, 0, 0, stream())))
    print(json.dumps({"kernel": "splat_bwd_in", "ms": ms, "GBps": 40.0 * px / ms / 1e6}))
    ms = timed(lambda: check(lib().ofd_splat_bwd_flow(ptr(img4), ptr(flow), ptr(g), ptr(gf), B, 4, H, W, 1, 0, 0, stream())))
    print(json.dumps({"kernel": "splat_bwd_flow", "ms": ms, "GBps": 48.0 * px / ms / 1e6}))
    # grid_sample warp backward: d/d second = scatter (splat kernel on grid coordinates), d/d flow = gather
    g3 = torch.rand(B, 3, H, W, device="cuda")
    ws = torch.empty(lib().ofd_splat_workspace_bytes(B, H, W), dtype=torch.uint8, device="cuda")
    gs, gfl = torch.empty_like(img3), torch.empty_like(flow)
    ms = timed(lambda: check(lib().ofd_grid_warp_bwd(ptr(img3), ptr(flow), ptr(g3), ptr(gs), None, B, 3, H, W, 24, ptr(ws), ws.numel(), stream())))
    print(json.dumps({"kernel": "grid_warp_bwd d/d second (scatter)", "ms": ms, "GBps": 32.0 * px / ms / 1e6}))
    ms = timed(lambda: check(lib().ofd_grid_warp_bwd(ptr(img3), ptr(flow), ptr(g3), None, ptr(gfl), B, 3, H, W, 24, None, 0, stream())))
    print(json.dumps({"kernel": "grid_warp_bwd d/d flow (gather)", "ms": ms, "GBps": 40.0 * px / ms / 1e6}))
    # CPU oracle on a bounded sample (1 of 16 samples)
    from oracle import warp_ref as WR
    t0 = time.time()
    WR.splat_out(img4[:1].cpu(), cases["smooth"][:1].cpu())
    dt = time.time() - t0
    print(json.dumps({"cpu_baseline": "oracle/splat_ref.c scalar, 1 core", "sample": "1 of 16 samples", "ms_per_sample": dt * 1e3,
                      "GBps": 40.0 * H * W / dt / 1e9}))


if __name__ == "__main__":
    main()
