"""Builds libssr_hip.so (the C-ABI library, include/ssr_hip.h) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["conv.hip", "conv_res.hip", "conv_ws.hip", "conv_big.hip", "conv_big_x3.hip", "conv_x3q.hip", "conv_x3r.hip", "conv_x3c.hip", "conv_thin.hip", "rdb_fwd.hip", "rdb_tile.hip", "wgrad.hip", "wgrad_bf16.hip", "wgrad_x3.hip", "misc.hip", "metrics.hip", "vgg.hip"]
OUT = os.path.join(HERE, "libssr_hip.so")


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"]
STAMP = OUT + ".srchash"     # sha256 of every source, header and flag the library was built from (travels with the .so)


def source_hash() -> str:
    """Content hash of the inputs of the build: a stale prebuilt library (older sources, other flags) is rebuilt whatever
    the file times say, and `build()` re-using a library means it was built from exactly these sources."""
    import glob
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    deps = [os.path.join(CSRC, s) for s in SOURCES] + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + \
           [os.path.join(ROOT, "include", "ssr_hip.h")]
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def build(force: bool = False, verbose: bool = True) -> str:
    """One hipcc -c per source, in parallel (the big-tile conv alone is ~50 s of the ~110 s a single command takes), then a
    shared-library link.  Objects go to csrc/build/ (git-ignored)."""
    if not force and not needs_build():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = FLAGS + ["-I" + os.path.join(ROOT, "include")]
    stamp = source_hash()

    import glob
    import hashlib
    hdr = hashlib.sha256(" ".join(FLAGS).encode())
    for d in sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(ROOT, "include", "ssr_hip.h")]:
        with open(d, "rb") as f:
            hdr.update(f.read())

    def compile_one(src: str) -> str:
        """One object per source; an object is reused only if its source, every header and the flags are unchanged."""
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        h = hdr.copy()
        with open(os.path.join(CSRC, src), "rb") as f:
            h.update(f.read())
        tag = obj + ".srchash"
        if not force and os.path.exists(obj) and os.path.exists(tag) and open(tag).read().strip() == h.hexdigest():
            return obj
        cmd = [hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        with open(tag, "w") as f:
            f.write(h.hexdigest() + "\n")
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
    if verbose:
        print("[build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as f:
        f.write(stamp + "\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
