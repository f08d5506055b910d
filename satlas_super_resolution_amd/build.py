"""Builds libssr_hip.so (the C-ABI library, include/ssr_hip.h) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["conv.hip", "conv_res.hip", "conv_ws.hip", "conv_big.hip", "conv_thin.hip", "rdb_fwd.hip", "wgrad.hip", "wgrad_bf16.hip", "misc.hip"]
OUT = os.path.join(HERE, "libssr_hip.so")


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "wgrad_common.h"), os.path.join(CSRC, "conv_epilogue.h"),
                                                       os.path.join(ROOT, "include", "ssr_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """One hipcc -c per source, in parallel (the big-tile conv alone is ~50 s of the ~110 s a single command takes), then a
    shared-library link.  Objects go to csrc/build/ (git-ignored)."""
    if not force and not needs_build():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-I" + os.path.join(ROOT, "include")]

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
    if verbose:
        print("[build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
