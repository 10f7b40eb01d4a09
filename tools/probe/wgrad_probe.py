"""Times ofd_conv_wgrad (3x3) on the layer shapes of the training step, with the ablation bits of conv_wgrad3_db_kernel (OFD_WGRAD_DBG, read per call)."""
import ctypes, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from opticalflowdiffusion_amd import _lib as L
    lib = L.lib()
    shapes = [(16, 440, 1024, 64, 64), (16, 220, 512, 128, 128), (16, 55, 128, 512, 512)]
    if os.environ.get("WGRAD_PROBE_SHAPES"): shapes = shapes[1:2]
    dbgs = [int(v) for v in os.environ.get("WGRAD_PROBE_DBGS", "0,1,2,4,3,5,6").split(",")]
    for (B, H, W, Cin, Cout) in shapes:
        x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
        dy = torch.randn(B, H, W, Cout, device="cuda").to(torch.bfloat16)
        acc = torch.zeros(9 * Cin * Cout, device="cuda")
        a = L.ConvArgs()
        a.B, a.H, a.W, a.ksize, a.n_src, a.Cout = B, H, W, 3, 1, Cout
        a.src[0].src = x.data_ptr(); a.src[0].channels = Cin; a.src[0].src_channels = Cin
        for dbg in dbgs:
            os.environ["OFD_WGRAD_DBG"] = str(dbg)
            for _ in range(3):
                L.check(lib.ofd_conv_wgrad(ctypes.byref(a), L.ptr(dy), L.ptr(acc), L.stream()))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                L.check(lib.ofd_conv_wgrad(ctypes.byref(a), L.ptr(dy), L.ptr(acc), L.stream()))
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            fl = 2.0 * 9 * Cin * Cout * B * H * W
            print(json.dumps({"shape": [B, H, W, Cin, Cout], "dbg": dbg, "ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
