"""CPU: ISA audits of hand-ordered kernels (hipcc cross-compiles gfx950 without a GPU).

Variable-length attention kernels (csrc/attention_varlen.hip):

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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_four_wave_nt_gemm_keeps_its_registers_and_its_in_flight_loads(tmp_path):
    """csrc/gemm_nt_w128.hip: every instance of the kernel (one per epilogue) must
      * hold the 64 accumulator tiles in 256 AGPRs and use NO scratch -- a spill drains the LDS-DMA ring through vmcnt(0) waits, and
        [measured in round 5] a variant that spilled computed wrong column blocks;
      * not let the compiler touch a register between an uncounted asm load of it and the counted wait that names it
        (tools/asm_inflight_audit.py: residual / gelu'-factor rows prefetched in the epilogue);
      * issue its K-step exactly as written: 64 MFMAs, 16 fragment reads, 8 LDS-DMA pieces between two barriers -- in both feeds of the
        activation operand (round 6: 128-byte rows, a K-step pair per instruction; round 5: 64-byte pieces, VITK_NTW_A128=0)."""
    import re
    asm = tmp_path / "gemm_nt_w128.s"
    src = os.path.join(ROOT, "vit_pytorch_amd", "csrc", "gemm_nt_w128.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        src, "-o", str(asm)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    # per function: name ... AGPRs ... ScratchSize (the file also holds the one-lane XCC_ID probe of the stream-K tail)
    recs = re.findall(r"Function Name: (\S+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)", r.stderr, flags=re.S)
    recs = [(n, int(a), int(s)) for n, a, s in recs if "gemm_ntw_kernel" in n]
    names = [n for n, _, _ in recs]
    assert len(recs) == 20, recs       # 10 epilogues x 2 activation feeds
    assert all(s == 0 for _, _, s in recs), recs
    assert all(a == 256 for _, a, _ in recs), recs
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_inflight_audit.py"), str(asm)], capture_output=True, text=True)
    assert a.returncode == 0, a.stderr[-2000:]
    lines = [l for l in a.stdout.strip().splitlines() if "gemm_ntw_kernel" in l]
    assert len(lines) == 20 and all(": 0 compiler instruction(s)" in l for l in lines), a.stdout[-3000:]
    # the steady-state K-step of the plain epilogue's instance: the instructions between two consecutive barriers of the inner loop
    text = open(asm).read()
    for feed in ("Li1E", "Li0E"):
        body = text[text.index("gemm_ntw_kernelILi0ELi0E" + feed):]
        body = body[:body.index(".Lfunc_end")]
        steps = body.split("s_barrier")
        counts = [(seg.count("v_mfma_f32_16x16x32"), seg.count("ds_read_b128"), seg.count("buffer_load_dwordx4")) for seg in steps]
        assert counts.count((64, 16, 8)) >= 8, (feed, counts)
