"""ISA audit of csrc/attention_varlen.hip: inside the chunk loop of every kernel the ONLY vector-memory wait may be VL_SYNC's
(`s_waitcnt vmcnt(0) lgkmcnt(0)` in front of the barrier) -- any other vmcnt wait would make the steps of chunk c wait for the
LDS-DMA of chunk c + 1 -- and no scratch access may sit in the loop (a scratch reload is a vector-memory load: hipcc waits vmcnt for it).
Usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only csrc/attention_varlen.hip -o /tmp/vl.s; python tools/vl_isa_audit.py /tmp/vl.s"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
cur, ker = None, {}
for l in lines:
    m = re.match(r"^(_ZN\S*attn_varlen\S*):", l)
    if m:
        cur = m.group(1); ker[cur] = []
    elif cur is not None:
        ker[cur].append(l)
        if "s_endpgm" in l:
            cur = None
bad = 0
for k, ls in ker.items():
    name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", k)
    name = re.sub(r"EEEv.*", "", name)
    waits, inloop, scratch_loop = [], False, 0
    for l in ls:
        if re.match(r"^\.LBB\d+_\d+:", l):
            inloop = "Loop" in l          # "=>This Loop Header", "in Loop: Header=", "Parent Loop"
        elif l.startswith("; %bb."):
            inloop = "Loop" in l
        if inloop and "s_waitcnt" in l and "vmcnt" in l:
            waits.append(l.strip())
        if inloop and "scratch_" in l:
            scratch_loop += 1
    scratch = sum(1 for l in ls if "scratch_" in l)
    tr = sum(1 for l in ls if "ds_read_b64_tr_b16" in l)
    dma = sum(1 for l in ls if "global_load_lds" in l)
    extra = [w for w in waits if w != "s_waitcnt vmcnt(0) lgkmcnt(0)"]
    flag = "" if not extra and not scratch_loop else "   <-- CHECK"
    bad += bool(flag)
    print(f"{name:48s} lines {len(ls):5d} tr {tr:3d} dma {dma:3d} scratch {scratch} (in the loop: {scratch_loop}) loop vmcnt waits {waits}{flag}")
print("AUDIT", "FAILED" if bad else "ok")
