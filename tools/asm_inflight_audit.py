"""Audit a hipcc .s for inline-asm register loads that the compiler touches before the counted wait.

hipcc does not model `asm volatile("global_load_dwordx4 %0, ...")`: the destination counts as written at ;;#ASMEND, so a copy,
spill or reuse of that register before the kernel's own `s_waitcnt vmcnt(N)` statement (which names the destinations "+v")
moves or clobbers data that has not landed.  This walks the control-flow graph of every kernel in the file: the destinations of
asm loads are IN FLIGHT from their asm block until an asm block containing s_waitcnt vmcnt; any compiler instruction that reads
or writes an in-flight register is reported.     python tools/asm_inflight_audit.py file.s [kernel-substring]"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def used(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", line):
        out |= regs(tok)
    return out


def audit(name, lines):
    # instructions with (kind, text); blocks by label
    ins = []
    label_at = {}
    inasm = False
    for l in lines:
        s = l.strip()
        if not s or s.startswith(";") and not s.startswith(";;#ASM"):
            continue
        if s.startswith(";;#ASMSTART"):
            inasm = True; continue
        if s.startswith(";;#ASMEND"):
            inasm = False; continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            label_at[m.group(1)] = len(ins); continue
        if s.startswith("."):
            continue
        ins.append(("asm" if inasm else "c", s))
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, (k, s) in enumerate(ins):
        op = s.split()[0]
        if op == "s_branch":
            succ[i] = [label_at[s.split()[1]]]
        elif op.startswith("s_cbranch"):
            succ[i] = [label_at[s.split()[1]]] + ([i + 1] if i + 1 < n else [])
        elif op == "s_endpgm":
            succ[i] = []
        else:
            succ[i] = [i + 1] if i + 1 < n else []
    state_in = [set() for _ in range(n)]
    work = [0]
    seen = [False] * n
    bad = {}
    while work:
        i = work.pop()
        k, s = ins[i]
        cur = set(state_in[i])
        if k == "asm":
            if s.startswith("global_load") or s.startswith("buffer_load") and " lds" not in s:
                cur |= regs(s.split()[1].rstrip(","))
            elif s.startswith("s_waitcnt") and "vmcnt" in s:
                cur = set()
        else:
            if cur and (used(s) & cur):
                bad[i] = s
        for j in succ[i]:
            if not seen[j] or not cur <= state_in[j]:
                state_in[j] |= cur
                seen[j] = True
                work.append(j)
    print(f"{name[:90]}: {len(bad)} compiler instruction(s) touch in-flight asm destinations")
    for i in sorted(bad)[:12]:
        print("     ", bad[i])
    return len(bad)


def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    total = 0
    for m in re.finditer(r"\n(_Z\w+):[^\n]*\n", txt):
        name = m.group(1)
        if flt not in name:
            continue
        end = txt.index(".Lfunc_end", m.end())          # the whole function (a kernel may hold more than one s_endpgm: early exits)
        body = txt[m.end():end].split("\n")
        if not any("ASMSTART" in l for l in body):
            continue
        total += audit(name, body)
    sys.exit(1 if total else 0)


main()
