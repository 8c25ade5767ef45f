"""CPU: ISA audit of the variable-length attention kernels (csrc/attention_varlen.hip).

The kernels fill LDS by LDS-DMA while they multiply the previous chunk; that only overlaps if NO vector-memory wait sits inside
the chunk loop besides the one in front of the barrier.  hipcc inserts such waits on its own in front of a `ds_read_tr` builtin
while a DMA is in flight, and in front of every scratch reload -- so the property depends on how the source is written (asm
transposing reads, register budgets without spills) and is checked on the generated assembly of every instance
(tools/vl_isa_audit.py).  hipcc cross-compiles gfx950 without a GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_varlen_attention_loops_hold_no_vector_memory_wait(tmp_path):
    asm = tmp_path / "attention_varlen.s"
    src = os.path.join(ROOT, "vit_pytorch_amd", "csrc", "attention_varlen.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", str(asm)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "vl_isa_audit.py"), str(asm)], capture_output=True, text=True)
    assert a.returncode == 0, a.stderr[-2000:]
    lines = a.stdout.strip().splitlines()
    assert lines[-1] == "AUDIT ok", "\n".join(l for l in lines if "CHECK" in l or "AUDIT" in l)
    # every geometry / head width / dropout flavour of the three kernels was looked at: 5 widths x 2 geometries x 2 x 3 kernels
    assert sum("attn_varlen_" in l for l in lines) == 60
