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
    if not force and not needs_build():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
           "-I" + os.path.join(ROOT, "include")] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print("[build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
