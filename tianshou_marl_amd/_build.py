"""Build the HIP extension: every tianshou_marl_amd/csrc/*.hip -> lib/libtsmarl_hip.so (gfx950).

hipcc cross-compiles without a GPU.  Objects are cached by mtime under csrc/_obj/; the shared
library is kept in-tree so that it travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "lib", "libtsmarl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-ffp-contract=off"]


def _newest_header() -> float:
    hs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "tsmarl.h")]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, force: bool, hdr_time: float) -> str:
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= hdr_time):
        return obj
    cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if not srcs:
        raise RuntimeError("no HIP sources found")
    hdr_time = _newest_header()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr_time), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} from {len(srcs)} sources")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
