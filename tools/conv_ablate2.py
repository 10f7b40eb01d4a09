"""Ablation sweep of the generic 3x3 kernel on the shapes it runs in the UNet (bits of OFD_CONV_DBG: 1 no X loads, 4 no MFMA, 16 no stores)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
cases = [(512, 512, 55, 128, 16, 3, 1), (512, 512, 55, 128, 16, 3, 0), (256, 256, 110, 256, 16, 3, 1), (128, 128, 220, 512, 16, 3, 1), (128, 64, 440, 1024, 16, 3, 0)]
for c in cases:
    for dbg in (0, 1, 4, 16, 5, 17, 20, 21):
        env = dict(os.environ, OFD_CONV_DBG=str(dbg))
        subprocess.run([sys.executable, os.path.join(HERE, "conv_ablate.py"), "one"] + [str(v) for v in c], env=env)
