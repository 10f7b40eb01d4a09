"""Diagnostic: per-phase timeline of conv3x3_wp_kernel from in-kernel s_memtime stamps (OFD_WP_STAMPS=1 build of conv_wp.hip).

    bash tools/build_wp_variants.sh stamps "-DOFD_WP_STAMPS=1"
    OFD_LIB=$PWD/opticalflowdiffusion_amd/lib/libofd_hip_stamps.so python tools/probe/wp_stamps.py 64 64 440 1024 16 1

Stamp slots: 0 entry, 1 preamble done, 2 first tile landed (vmcnt(0)), 3 first tile staged (prologue + ds_write), 4 barrier of chunk 0 passed,
5 / 6 before / after the staging of the next chunk, 7 chunk 0 issued, 8 barrier of chunk 1 passed, 9 chunk 1 issued, 10 epilogue issued,
11 stores drained; 14 / 15 s_memrealtime at entry / exit (100 MHz), 13 HW_ID.
"""
import ctypes, json, math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from opticalflowdiffusion_amd import _lib as L

cin, cout, H, W, B, pro = map(int, sys.argv[1:7])
lib = L.lib()
x = torch.randn(B, H, W, cin, device="cuda").to(torch.bfloat16)
w = torch.randn(cout, cin, 3, 3, device="cuda") / math.sqrt(cin * 9)
wp = torch.empty(lib.ofd_conv_weight_elems(cout, cin, 3), dtype=torch.bfloat16, device="cuda")
L.check(lib.ofd_conv_weight_prep(L.ptr(w), L.ptr(wp), cout, cin, cin, 3, -1.0, 0, L.stream()))
out = torch.empty(B, H, W, cout, dtype=torch.bfloat16, device="cuda")
bias = torch.zeros(cout, device="cuda")
gn = torch.empty(lib.ofd_conv_gn_partial_count(B, H, W, cout), device="cuda")
sc = torch.ones(B, cin, device="cuda"); sh = torch.zeros(B, cin, device="cuda")
a = L.ConvArgs()
a.B, a.H, a.W, a.ksize, a.n_src, a.Cout = B, H, W, 3, 1, cout
a.src[0].src = x.data_ptr(); a.src[0].channels = cin; a.src[0].src_channels = cin
a.weight = wp.data_ptr(); a.bias = bias.data_ptr(); a.out = out.data_ptr()
if int(os.environ.get("STATS", "1")):
    a.gn_partial = gn.data_ptr()
if pro:
    a.in_scale = sc.data_ptr(); a.in_shift = sh.data_ptr()
for _ in range(30):
    L.check(lib.ofd_conv_forward(ctypes.byref(a), L.stream()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    L.check(lib.ofd_conv_forward(ctypes.byref(a), L.stream()))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20

if os.environ.get("OFD_CONV_PC", "1") != "0" and cout == 64:
    # producer / consumer kernel (conv3x3_pc_kernel): producers stamp step 24 of their walk, consumers item 12 (see PC_STAMP_* in conv_wp.hip)
    raw = ctypes.CDLL(L.LIB_PATH)
    SL, NW = 16, 1 << 16
    buf = np.zeros(SL * NW, dtype=np.uint64)
    assert raw.ofd_dbg_wp_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
    s = buf.reshape(NW, SL).astype(np.int64)
    s = s[:256 * 8].reshape(256, 8, SL)
    cons, prod = s[:, :4].reshape(-1, SL), s[:, 4:].reshape(-1, SL)
    med = lambda a: float(np.median(a))
    res = {"kernel": "conv3x3_pc_kernel", "shape": [cin, cout, H, W, B], "prologue": pro, "ms": ms,
           "producer_step_cycles": {"fetch (issue global loads of chunk i + 3)": med(prod[:, 2] - prod[:, 1]), "stage chunk i + 1 (prologue + ds_write)": med(prod[:, 3] - prod[:, 2]),
                                    "wait at the barrier": med(prod[:, 4] - prod[:, 3])},
           "consumer_item_cycles": {"chunk 0 issued": med(cons[:, 6] - cons[:, 5]), "barrier": med(cons[:, 7] - cons[:, 6]), "chunk 1 issued": med(cons[:, 8] - cons[:, 7]),
                                    "barrier ": med(cons[:, 9] - cons[:, 8]), "epilogue": med(cons[:, 10] - cons[:, 9]), "item": med(cons[:, 10] - cons[:, 5])},
           "wave_life_cycles": med(s[:, :, 15].reshape(-1) - s[:, :, 14].reshape(-1)) * 20.0}
    it = buf[65536:65536 + 256 * 64].reshape(256, 64).astype(np.int64)
    t0 = s[:, 0, 14] * 0 + it[:, :1]                      # first item end as the origin of a workgroup's walk
    n_it = int((it[0] > 0).sum())
    if n_it > 2:
        d_it = np.diff(it[:, :n_it], axis=1)              # cycles per item, per workgroup
        res["item_cycles_by_index_median"] = [float(np.median(d_it[:, k])) for k in range(0, n_it - 1, max(1, (n_it - 1) // 14))]
        res["items"] = n_it
        c0 = s[:, 0, :]                                   # consumer wave 0 of every workgroup
        res["startup_cycles_median"] = {"kernel entry -> barrier 0 passed": float(np.median(c0[:, 11] - c0[:, 12])), "kernel entry -> first item done": float(np.median(it[:, 0] - c0[:, 12])),
                                         "last item done -> wave exit (x clock ratio)": None}
        res["walk_cycles_median"] = float(np.median(it[:, n_it - 1] - c0[:, 12]))
    print(json.dumps(res))
    sys.exit(0)
raw = ctypes.CDLL(L.LIB_PATH)
SL, NW = 16, 1 << 16
buf = np.zeros(SL * NW, dtype=np.uint64)
rc = raw.ofd_dbg_wp_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
s = buf.reshape(NW, SL).astype(np.int64)
s = s[s[:, 0] != 0]
d = s[:, :12] - s[:, :1]
life = d[:, 11]
clk = life / np.maximum(s[:, 15] - s[:, 14], 1) * 100.0      # MHz
names = ["preamble", "first tile lands", "stage tile 0 (prologue + ds_write)", "barrier 0", "chunk 0: g = 0, 1", "stage tile 1",
         "chunk 0: g = 2..5", "barrier 1", "chunk 1", "epilogue issue", "store drain"]
seg = np.diff(d, axis=1)
res = {"shape": [cin, cout, H, W, B], "prologue": pro, "ms": ms, "waves": int(len(s)), "wave_life_cycles_median": float(np.median(life)),
       "clock_MHz_median": float(np.median(clk)), "segments_cycles_median": {n: float(np.median(seg[:, i])) for i, n in enumerate(names)},
       "segments_cycles_p90": {n: float(np.percentile(seg[:, i], 90)) for i, n in enumerate(names)}}
print(json.dumps(res))
