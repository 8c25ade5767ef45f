"""CPU: the C-ABI library loads without a GPU and exports exactly what include/vitk.h declares;
host-side argument validation works without launching anything."""
import ctypes
import os
import re

import pytest

from vit_pytorch_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vitk.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+(vitk_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[name] = n
    return out


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return L.load()


def test_header_binding_and_exports_agree(lib):
    hdr = header_functions()
    assert len(hdr) >= 28
    assert set(hdr) == set(L.SIGNATURES), set(hdr) ^ set(L.SIGNATURES)
    for name, nargs in hdr.items():
        assert hasattr(lib, name), f"{name} not exported by libvitk.so"
        assert len(L.SIGNATURES[name][1]) == nargs, (name, nargs, len(L.SIGNATURES[name][1]))


def test_f16_flavour_exports_the_same_abi():
    """libvitk_f16.so: the same sources with the 16-bit type switched to IEEE half; same symbols, told apart by vitk_half_type()."""
    if not os.path.exists(L.LIB_PATH_F16):
        import __graft_entry__ as g
        g.build()
    lib16 = L.load_f16()
    assert lib16.vitk_half_type() == L.HALF_TYPE_F16 and L.load().vitk_half_type() == L.BF16
    for name in header_functions():
        assert hasattr(lib16, name), f"{name} not exported by libvitk_f16.so"


def test_version_and_no_torch_symbols(lib):
    assert lib.vitk_version() == L.VITK_VERSION
    # the boundary is plain C: the library must not depend on libtorch / libc10
    import subprocess
    deps = subprocess.run(["ldd", L.LIB_PATH], capture_output=True, text=True).stdout
    assert "libtorch" not in deps and "libc10" not in deps
    assert "libamdhip64" in deps


def test_host_side_validation_without_gpu(lib):
    # bad shapes / null pointers are rejected on the host with a negative code and a message; nothing is launched
    rc = lib.vitk_gemm_nt_bf16(None, 64, None, 64, None, 64, 64, 64, 64, 0, None, None, None, None)
    assert rc == -1 and b"null" in lib.vitk_last_error()
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p = (p + 255) // 256 * 256
    rc = lib.vitk_gemm_nt_bf16(p, 40, p, 40, p, 64, 64, 64, 40, 0, None, None, None, None)   # K % 32 != 0
    assert rc == -2 and b"K % 32" in lib.vitk_last_error()
    rc = lib.vitk_layernorm_fwd(p, 0, p, p, 0, p, 0, p, p, 4, 5000, 1e-5, L.IDENT, L.IDENT, None, 0, 0, None)  # D > 4096
    assert rc == -2
    q = L.BHND(p, 64, 64, 64)
    rc = lib.vitk_attn_fwd_bf16(q, q, q, q, p, 1, 1, 16, 80, 0.1, None)  # dim_head != 64 on the fused path
    assert rc == -2 and b"dim_head" in lib.vitk_last_error()
    assert lib.vitk_gemm_tn_splits(50432, 2304, 768) >= 1
    assert lib.vitk_layernorm_bwd_blocks(50432, 768) == 512 and lib.vitk_layernorm_bwd_blocks(50432, 1024) == 768
    # round 4's entry points: the multi-tensor cast, the image gradient of the Rearrange, the per-head RMSNorm at a free dim_head
    tab = (ctypes.c_void_p * 2)(p, p + 2)                       # second float32 tensor not 16-byte aligned
    n2 = (ctypes.c_int64 * 2)(8, 8)
    cv = lambda a: ctypes.cast(a, ctypes.c_void_p)
    assert lib.vitk_cast_many(cv(tab), cv(tab), cv(n2), 0, 0, 1, None) == 0                       # empty table: nothing to do
    assert lib.vitk_cast_many(None, cv(tab), cv(n2), 2, 0, 1, None) == -1
    assert lib.vitk_cast_many(cv(tab), cv(tab), cv(n2), 2, 0, 1, None) == -1 and b"misaligned" in lib.vitk_last_error()
    assert lib.vitk_cast_many(cv(tab), cv(tab), cv(n2), 2, 0, 7, None) == -4                      # unknown dtype tag (VITK_E_DTYPE)
    assert lib.vitk_unpatchify(p, p, 0, 2, 3, 30, 32, 4, 4, None) == -2 and b"divisible" in lib.vitk_last_error()
    assert lib.vitk_unpatchify(None, p, 0, 2, 3, 32, 32, 4, 4, None) == -1
    assert lib.vitk_rmsnorm_heads_fwd(p, 96, p, p, 96, p, 0, 4, 2, 50, None) == -2 and b"dim_head" in lib.vitk_last_error()    # 50 % 4 != 0
    assert lib.vitk_rmsnorm_heads_fwd(p, 640, p, p, 640, p, 0, 4, 2, 320, None) == -2                                           # > 256


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.VitkError, match="not built"):
        L.load()


C_HOST = r"""
/* a host in plain C: what a maintainer binding libvitk from C / cgo / JNI compiles (INTEGRATION.md) */
#include <stdio.h>
#include <string.h>
#include "vitk.h"

int main(void) {
    int32_t plan[5] = {0, 0, 0, 0, 0};
    printf("version %d\n", vitk_version());
    if (vitk_gemm_nt_plan(50432, 2304, 768, 2304, plan) != 0) return 2;
    printf("plan %d %d %d %d %d\n", (int)plan[0], (int)plan[1], (int)plan[2], (int)plan[3], (int)plan[4]);
    printf("splits %lld\n", (long long)vitk_gemm_tn_splits(50432, 2304, 768));
    printf("rows %lld\n", (long long)vitk_rmsnorm_heads_rows(4096, 16));
    /* argument validation happens on the host, before any launch: a null pointer is VITK_E_ARG with a message */
    int rc = vitk_cast(NULL, 0, NULL, 1, 16, NULL);
    printf("rc %d msg %s\n", rc, vitk_last_error());
    return rc == VITK_E_ARG && strlen(vitk_last_error()) > 0 ? 0 : 3;
}
"""


def test_header_is_c99_and_a_plain_c_host_links_and_runs(lib, tmp_path):
    """`extern "C"`, plain pointers and sizes: the header compiles as strict C99 (and as C++), and a C program linked against
    libvitk.so calls the host-side entry points without a GPU."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", HEADER], check=True)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", HEADER], check=True)
    src = tmp_path / "host.c"
    src.write_text(C_HOST)
    exe = tmp_path / "host"
    libdir = os.path.dirname(L.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-l:" + os.path.basename(L.LIB_PATH),
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert f"version {lib.vitk_version()}" in r.stdout and "plan" in r.stdout and "rc -" in r.stdout
