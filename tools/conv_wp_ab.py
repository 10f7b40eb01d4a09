"""A/B of the wave-private-weights 3x3 kernel (OFD_CONV_WP) against the shared-slab kernel on the UNet's shapes."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
cases = [(512, 512, 55, 128, 16, 3, 1), (512, 512, 55, 128, 16, 3, 0), (768, 512, 55, 128, 16, 3, 0), (256, 256, 110, 256, 16, 3, 1), (128, 128, 220, 512, 16, 3, 1),
         (128, 64, 440, 1024, 16, 3, 0), (64, 64, 440, 1024, 16, 3, 1), (64, 64, 440, 1024, 16, 3, 0)]
which = sys.argv[1:] or ["0", "7"]
for c in cases:
    for wp in which:
        env = dict(os.environ, OFD_CONV_WP=wp)
        print(f"WP={wp} ", end="", flush=True)
        subprocess.run([sys.executable, os.path.join(HERE, "conv_ablate.py"), "one"] + [str(v) for v in c], env=env)
