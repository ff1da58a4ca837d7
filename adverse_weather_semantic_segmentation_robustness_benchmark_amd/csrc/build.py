"""Build libawseg_hip.so for gfx950 with hipcc (in-tree; the .so travels to the GPU box).

    python -m adverse_weather_semantic_segmentation_robustness_benchmark_amd.csrc.build

-ffp-contract=off is load-bearing: the parity kernels must round once per operation exactly
as numpy / torch evaluate the reference expressions.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent
PKG = CSRC.parent
LIB = PKG / "libawseg_hip.so"
SOURCES = ["core.hip", "metrics.hip", "weather.hip", "loss.hip", "heads.hip", "backbone.hip", "depth.hip", "wino.hip", "gemm.hip", "attn.hip", "gemm_split.hip", "gemm_split3.hip", "wino_split.hip", "depthfuse.hip", "mixffn.hip", "dwtrain.hip", "bntrain.hip", "smallops.hip"]
ARCH = "gfx950"
# per-file extras.  wino.hip: the SLP vectoriser packs the scalar inverse transform of the fused-head epilogue into
# v_pk_add_f32 fed by ~220 v_mov (and spills); the kernel packs by hand where adjacent registers make it free.
HEADER = PKG.parent / "include" / "awseg.h"


def header_hash() -> int:
    """60 bits of sha256(include/awseg.h): compiled into the library (awseg_header_hash) and compared by _native.lib() when it loads
    it, so that a library built from another state of the ABI is refused instead of being called with the wrong arguments."""
    import hashlib
    return int(hashlib.sha256(HEADER.read_bytes()).hexdigest()[:15], 16)


EXTRA_FLAGS = {"attn.hip": ["-DAWSEG_ATTN_SPLIT_WAVES=" + os.environ.get("AWSEG_ATTN_SPLIT_WAVES", "2")], "wino.hip": ["-fno-slp-vectorize"], "wino_split.hip": ["-fno-slp-vectorize", "-DAWSEG_WS_ASM_PK"],
               "core.hip": ["-DAWSEG_HEADER_HASH=0x%xULL" % header_hash()]}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libawseg_hip.so)")


def _flags():
    return [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
            "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-inline-asm"]   # inline asm: wino.hip clobbers m0 on purpose


STAMP = PKG / "libawseg_hip.flags"       # the flags the installed .so was built with (mtimes alone miss an environment-selected -D)


def _flag_stamp() -> str:
    return repr((_flags(), sorted(EXTRA_FLAGS.items()), SOURCES))


def needs_build() -> bool:
    if not LIB.exists():
        return True
    if not STAMP.exists() or STAMP.read_text() != _flag_stamp():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [CSRC / "awseg_common.h", HEADER]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    cc = hipcc()
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)

    def compile_one(src):
        obj = objdir / (Path(src).stem + ".o")
        cmd = [cc, *_flags(), *EXTRA_FLAGS.get(src, []), "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = LIB.with_suffix(".so.tmp")
    rocm_lib = str(Path(cc).resolve().parent.parent / "lib")
    subprocess.check_call([cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(tmp), *map(str, objs),
                           f"-L{rocm_lib}", "-lhipblaslt", f"-Wl,-rpath,{rocm_lib}"])     # gemm.hip: plain library GEMMs
    os.replace(tmp, LIB)
    STAMP.write_text(_flag_stamp())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
