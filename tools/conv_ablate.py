"""Diagnostic: time one conv layer under the kernel's ablation bits (OFD_CONV_DBG).
bit0 (1): synthetic X instead of global loads; bit2 (4): no MFMA; bit4 (16): no epilogue stores."""
import ctypes, math, os, subprocess, sys, torch

def run_one(cin, cout, H, W, B, ks, prologue):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from opticalflowdiffusion_amd import _lib as L
    L.lib()
    x = torch.randn(B, H, W, cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(cout, cin, ks, ks, device="cuda") / math.sqrt(cin * ks * ks)
    wp = torch.empty(L.lib().ofd_conv_weight_elems(cout, cin, ks), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().ofd_conv_weight_prep(L.ptr(w), L.ptr(wp), cout, cin, cin, ks, -1.0, 0, L.stream()))
    out = torch.empty(B, H, W, cout, dtype=torch.bfloat16, device="cuda")
    bias = torch.zeros(cout, device="cuda")
    gn = torch.empty(L.lib().ofd_conv_gn_partial_count(B, H, W, cout), device="cuda")
    sc = torch.ones(B, cin, device="cuda"); sh = torch.zeros(B, cin, device="cuda")
    a = L.ConvArgs()
    a.B, a.H, a.W, a.ksize, a.n_src, a.Cout = B, H, W, ks, 1, cout
    a.src[0].src = x.data_ptr(); a.src[0].channels = cin; a.src[0].src_channels = cin
    a.weight = wp.data_ptr(); a.bias = bias.data_ptr(); a.out = out.data_ptr(); a.gn_partial = gn.data_ptr()
    if prologue:
        a.in_scale = sc.data_ptr(); a.in_shift = sh.data_ptr()
    for _ in range(int(os.environ.get('OFD_ABL_WARM', 40))):
        L.check(L.lib().ofd_conv_forward(ctypes.byref(a), L.stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = int(os.environ.get('OFD_ABL_N', 40))
    for _ in range(n):
        L.check(L.lib().ofd_conv_forward(ctypes.byref(a), L.stream()))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * H * W * cout * cin * ks * ks
    print(f"dbg={os.environ.get('OFD_CONV_DBG','0'):>3s} {cin}->{cout} k{ks} {H}x{W} prologue={prologue}: {ms:.3f} ms  {fl/ms/1e9:.0f} TF/s")

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        cin, cout, H, W, B, ks, pro = map(int, sys.argv[2:9])
        run_one(cin, cout, H, W, B, ks, pro)
    else:
        c = (64, 64, 440, 1024, 16, 3, 0)
        for grid in (256,):
            env = dict(os.environ, OFD_CONV_DBG="0", OFD_PP_GRID=str(grid))
            print("grid", grid, end=" ", flush=True)
            subprocess.run([sys.executable, __file__, "one"] + [str(v) for v in c], env=env)
        for dbg in (1, 4, 16, 48, 52):
            env = dict(os.environ, OFD_CONV_DBG=str(dbg))
            subprocess.run([sys.executable, __file__, "one"] + [str(v) for v in c], env=env)
