"""Build libvitk.so (the C-ABI kernel library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU.  The library lands next to this file
(``vit_pytorch_amd/libvitk.so``): git-ignored, but shipped to the GPU box by
gpurun.  Objects are cached under ``vit_pytorch_amd/csrc/build/``.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvitk.so")
LIB_F16 = os.path.join(HERE, "libvitk_f16.so")
SOURCES = ["elementwise.hip", "layernorm.hip", "gemm_bf16.hip", "gemm_nt_persist.hip", "gemm_nt_w128.hip", "gemm_tn_w128.hip", "gemm_tn_fp8.hip", "gemm_generic.hip", "attention.hip", "attention_pipe.hip", "attention_varlen.hip", "comm.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_nt_plan.h"), os.path.join(CSRC, "gemm_nt_epi.h"), os.path.join(CSRC, "attention_pipe.h"), os.path.join(CSRC, "attention_frag.h"), os.path.join(os.path.dirname(HERE), "include", "vitk.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _build_one(lib: str, bdir: str, extra, force: bool, verbose: bool) -> str:
    hipcc = _hipcc()
    os.makedirs(bdir, exist_ok=True)
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(bdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            jobs.append([hipcc, *FLAGS, *extra, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[vitk build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1) or 1) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-ldl"])
    return lib


def build_lib(force: bool = False, verbose: bool = True) -> str:
    """libvitk.so (16-bit type = bfloat16) and libvitk_f16.so (same sources, 16-bit type = IEEE half).

    VITK_BUILD_EXPERIMENTS=1 builds the EXPERIMENTS flavour instead -- the knobs of vitk_exp() in csrc/common.h compiled in (tile orders,
    cost-model constants, debug stamps: what the tools/ scripts switch) -- under its own names, libvitk_exp.so / libvitk_f16_exp.so, from its
    own object directory: the product libraries are never overwritten by it (a later product build used to see nothing stale and keep the
    experiments flavour under the product name).  Load it with VITK_LIB=.../libvitk_exp.so (the float16 sibling libvitk_f16_exp.so is found beside it)."""
    exp = os.environ.get("VITK_BUILD_EXPERIMENTS", "0") not in ("0", "")
    if exp:
        flags = ["-DVITK_EXPERIMENTS=1"]
        _build_one(LIB_F16.replace(".so", "_exp.so"), os.path.join(CSRC, "build", "exp", "f16"), ["-DVITK_HALF_IS_F16=1", *flags], force, verbose)
        return _build_one(LIB.replace(".so", "_exp.so"), os.path.join(CSRC, "build", "exp"), flags, force, verbose)
    _build_one(LIB_F16, os.path.join(CSRC, "build", "f16"), ["-DVITK_HALF_IS_F16=1"], force, verbose)
    return _build_one(LIB, os.path.join(CSRC, "build"), [], force, verbose)


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
