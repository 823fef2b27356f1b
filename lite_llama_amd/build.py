"""Build the gfx950 C-ABI shared library in-tree with hipcc (no torch extension machinery).

    python -m lite_llama_amd.build          # -> lite_llama_amd/lib/liblite_llama_amd.so

hipcc cross-compiles for gfx950 without a GPU; the built .so is git-ignored but travels
to the GPU box with the repo snapshot.
"""

from __future__ import annotations

import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "liblite_llama_amd.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off"]
FLAGS += os.environ.get("LL_EXTRA_HIPCC_FLAGS", "").split()  # debug builds (e.g. -DV2_DEBUG_ABLATE)
# per-source flags.  gemm_short.hip: its kernel's first 16 argument dwords arrive in SGPRs with the wave (gfx950 kernarg preload)
# instead of through dependent scalar loads -- a launch of ~5 us spends ~0.3 us there otherwise.
FILE_FLAGS = {"gemm_short.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=16"]}


def flags_for(src: str) -> list[str]:
    return FLAGS + FILE_FLAGS.get(os.path.basename(src), [])


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def sources() -> list[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(paths: list[str]) -> str:
    h = hashlib.sha256()
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    for p in sorted(paths) + headers + [os.path.join(HERE, "..", "include", "lite_llama_amd.h")]:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sources()
    stamp = os.path.join(LIBDIR, "build.stamp")
    digest = _digest(srcs)
    if not force and os.path.exists(lib_path()) and os.path.exists(stamp):
        if open(stamp).read().strip() == digest:
            return lib_path()
    hipcc = _hipcc()
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc, *flags_for(src), "-c", src, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", lib_path()]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(digest)
    return lib_path()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
