import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from opticalflowdiffusion_amd import _lib as L
from test_backward_gpu import conv_args
L.lib()
B, H, W, Cin, Cout = 1, 8, 32, 64, 64
for (py, px, ci, qy, qx, co) in ((3, 5, 7, 3, 5, 9), (3, 5, 40, 2, 4, 33), (0, 0, 0, 1, 1, 63)):
    x = torch.zeros(B, H, W, Cin, dtype=torch.bfloat16, device="cuda"); x[0, py, px, ci] = 1
    dy = torch.zeros(B, H, W, Cout, dtype=torch.bfloat16, device="cuda"); dy[0, qy, qx, co] = 2
    a = conv_args(L, B, H, W, 3, [dict(t=x)], Cout)
    acc = torch.zeros(9 * Cin * Cout, device="cuda")
    L.check(L.lib().ofd_conv_wgrad(ctypes.byref(a), L.ptr(dy), L.ptr(acc), L.stream()))
    torch.cuda.synchronize()
    acc = acc.cpu().reshape(9, Cin, Cout)
    nz = acc.nonzero()
    # expected: x at p + (ky-1, kx-1) = (py, px) with p = (qy, qx) -> ky = py - qy + 1, kx = px - qx + 1
    print("x@", (py, px, ci), "dy@", (qy, qx, co), "expect tap", ((py - qy + 1) * 3 + (px - qx + 1), ci, co), "got", [(tuple(i.tolist()), float(acc[tuple(i)])) for i in nz][:8])
