"""Does the GroupNorm+SiLU backward (reduce pass then apply pass over the same two tensors) run faster when it is launched per
sample chunk, so that the apply pass finds the chunk's dy / h in the 256 MB Infinity Cache?  One call over B=16 against
16 / 8 / 4 calls over 1 / 2 / 4 samples, at the full-resolution (64 ch) and half-resolution (128 ch) sizes."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from opticalflowdiffusion_amd import _lib as L
    lib = L.lib()
    for (B, H, W, C) in ((16, 440, 1024, 64), (16, 220, 512, 128), (16, 110, 256, 256)):
        n = B * H * W * C
        g = torch.randn(n, device="cuda").to(torch.bfloat16)
        h = torch.randn(n, device="cuda").to(torch.bfloat16)
        dh = torch.empty_like(h)
        a = torch.rand(B, C, device="cuda") + 0.5
        s = torch.randn(B, C, device="cuda") * 0.1
        stats = torch.stack([torch.zeros(B, 8, device="cuda"), torch.ones(B, 8, device="cuda")], -1).contiguous()
        gam, bet = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        dgam, dbet, dcb = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        wsp = torch.empty(lib.ofd_gn_bwd_workspace_floats(B, H, W, C), device="cuda")
        per = H * W * C * 2

        def run(chunk):
            for b0 in range(0, B, chunk):
                L.check(lib.ofd_gn_silu_backward(g.data_ptr() + b0 * per, h.data_ptr() + b0 * per, a.data_ptr() + b0 * C * 4, s.data_ptr() + b0 * C * 4,
                                                 stats.data_ptr() + b0 * 64, L.ptr(gam), L.ptr(bet), None, 0, 0, dh.data_ptr() + b0 * per, L.ptr(dgam),
                                                 L.ptr(dbet), None, L.ptr(dcb), L.ptr(wsp), chunk, H, W, C, L.stream()))
        ref = None
        for chunk in (16, 4, 2, 1, 16):
            run(chunk); torch.cuda.synchronize()
            if ref is None: ref = dh.clone()
            same = bool(torch.equal(ref, dh))
            t0 = time.perf_counter()
            for _ in range(10): run(chunk)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            print(json.dumps({"shape": [B, H, W, C], "samples_per_call": chunk, "ms": round(ms, 4), "GBps_algorithmic(10B/el)": round(n * 10 / ms / 1e6, 1), "dh_equal": same}), flush=True)


if __name__ == "__main__":
    main()
