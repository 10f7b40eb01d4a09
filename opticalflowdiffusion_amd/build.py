"""Builds libofd_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m opticalflowdiffusion_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, "libofd_hip.so")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics",
          "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]
# per-file extra flags: the warp kernels need the reference's exact fp32 operation order
EXTRA = {"warp.hip": ["-ffp-contract=off"], "diffusion.hip": ["-ffp-contract=off"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "ofd.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force):
    obj = os.path.join(OUT_DIR, "obj", src + ".o")
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), _deps_mtime())):
        return obj, False
    cmd = ["hipcc", "-x", "hip", "-c", path, "-o", obj] + COMMON + EXTRA.get(src, [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(os.path.join(OUT_DIR, "obj"), exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    if any(c for _, c in results) or not os.path.exists(LIB):
        cmd = ["hipcc", "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"built {LIB} from {len(objs)} objects")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
