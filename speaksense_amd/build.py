"""Builds libspeaksense_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libspeaksense_hip.so")
LIB_W155 = os.path.join(HERE, "libspeaksense_whisper_post154.so")
SOURCES = ["model.cpp", "kernels_mel.hip", "kernels_gemm.hip", "kernels_gemm_fp8.hip", "kernels_attn.hip", "kernels_misc.hip", "kernels_decode.hip", "kernels_denoise.hip", "kernels_resample.hip", "engine.cpp", "capi.cpp"]
HEADERS = ["common.h", "kernels.h", "gemm_common.h", "wave_ops.h", "engine.h", os.path.join("..", "..", "include", "speaksense.h"), os.path.join("..", "..", "include", "whisper_compat.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-x", "hip"]


def _stale(obj: str, src: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src] + [os.path.join(CSRC, h) for h in HEADERS if os.path.exists(os.path.join(CSRC, h))]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    srcs = [s for s in SOURCES + ["whisper_compat.cpp"] if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s + ".o")
        if force or _stale(obj, src):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])
    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r.stderr
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn:
                print(warn, file=sys.stderr)
    objs = [os.path.join(objdir, s + ".o") for s in srcs]
    if jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"])
    # the whisper.h shim once more, laid out as whisper.h v1.5.5 (include/whisper_compat.h SS_WHISPER_H_POST_1_5_4): a small library of its own that
    # a whisper-rs-sys build links BEFORE libspeaksense_hip.so.  -Bsymbolic: its whisper_* calls among themselves stay inside it.
    src = os.path.join(CSRC, "whisper_compat.cpp")
    if os.path.exists(src) and (jobs or not os.path.exists(LIB_W155) or _stale(LIB_W155, src)):
        obj = os.path.join(objdir, "whisper_compat_post154.o")
        run([hipcc] + FLAGS + ["-DSS_WHISPER_H_POST_1_5_4", "-c", src, "-o", obj])
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", LIB_W155, obj, "-L" + HERE, "-lspeaksense_hip", "-Wl,-rpath,$ORIGIN"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
