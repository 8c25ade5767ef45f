"""profiles/<tag>_pmc_traffic.json from the two PMC passes of tools/profile_round.sh:
    python tools/pmc_traffic_json.py gpurun_out/<tag>_pmc_fetch gpurun_out/<tag>_pmc_write profiles/<tag>_pmc_traffic.json \
           [gpurun_out/<tag>_pmc_mfma profiles/<tag>_pmc_mfma_util.json]
Launch groups (consecutive dispatches of one GEMM kernel) are labelled by the fixed order of tools/kprof.py.  FETCH_SIZE /
WRITE_SIZE are KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-B requests at 64 B)."""
import csv, glob, json, os, re, sys

M, D, F = 50432, 768, 3072


def groups(root, counter):
    rows = []
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            n = r["Kernel_Name"]
            if "gemm_" not in n or "reduce" in n or "pack_w" in n:
                continue
            rows.append((int(r["Dispatch_Id"]), n, float(r["Counter_Value"])))
    rows.sort()
    # a CALL of the NT entry point may be two dispatches since round 5: gemm_ntw_kernel<EPI, ..> on the whole rounds, then gemm_ntp_kernel<EPI, ..>
    # on the remaining rows -- their counters are summed
    calls = []
    for _, n, v in rows:
        m = re.search(r"(gemm_\w+)(<[^>]*>)?", n)
        name, targs = m.group(1), (m.group(2) or "")
        epi = targs.strip("<>").split(",")[0].strip() if targs else ""
        if (name == "gemm_ntp_kernel" and calls and calls[-1][2] == ("gemm_ntw_kernel", epi) and not calls[-1][3]):
            calls[-1][1] += v
            calls[-1][0] += " + gemm_ntp_kernel" + targs
            calls[-1][3] = True
        else:
            calls.append([name + targs, v, (name, epi), False])
    out = []
    for key, v, _, _ in calls:
        if out and out[-1][0] == key and len(out[-1][1]) < 3:
            out[-1][1].append(v)
        else:
            out.append([key, [v]])
    return out


def main():
    fetch_dir, write_dir, dst = sys.argv[1:4]
    mfma_dir, mfma_dst = (sys.argv[4:6] if len(sys.argv) >= 6 else (None, None))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    labels = []
    f32s = bool(os.environ.get("KPROF_F32_STREAM"))        # rounds 1-3: float32 forward stream (EPI_RESID); round 4 default: 16-bit (EPI_RESID16)
    rn, re_ = ("f32 residual", 3) if f32s else ("16-bit residual", 5)
    for lab, n, k, epi in (("QKV", 3 * D, D, 0), ("FF1 bias+GELU", F, D, 2), (f"out-proj + {rn}", D, D, re_), (f"FF2 + {rn}", D, F, re_),
                           ("dFF1 GELU' + column sums", F, D, 4), ("dX of FF1 (K=3072)", D, F, 0)):
        # FF1 / dFF1: the second (M, N) tensor is the gelu' factor -- 8-bit codes since round 5 (KPROF_DG16 / KPROF_NO_DG: 16-bit values)
        aux_b = (2 if (os.environ.get("KPROF_DG16") or os.environ.get("KPROF_NO_DG")) else 1) if epi in (2, 4) else 0
        alg = 2 * (M * k + n * k) + (8 * M * n if epi == 3 else 4 * M * n if epi == 5 else (2 + aux_b) * M * n)
        labels.append((lab, n, k, alg))
    for lab, n, k in (("dW qkv", 3 * D, D), ("dW ff1", F, D)):
        labels.append((lab, n, k, 2 * (M * n + M * k + n * k)))
    gf, gw = groups(fetch_dir, "FETCH_SIZE"), groups(write_dir, "WRITE_SIZE")
    assert len(gf) == len(gw) == len(labels), (len(gf), len(gw), len(labels), [g[0] for g in gf])
    kernels = {}
    for (lab, n, k, alg), (kf, vf), (kw, vw) in zip(labels, gf, gw):
        assert kf == kw, (kf, kw)
        fk, wk = vf[-1], vw[-1]                     # last launch of the group: steady state
        kernels[f"{kf} {lab} {M}x{n}x{k}"] = {"FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "traffic_bytes": int((2 * fk + wk) * 1024),
                                              "algorithmic_bytes": alg}
    doc = {"note": "HBM-side bytes per launch from rocprofv3 PMC, separate passes (--pmc FETCH_SIZE ; --pmc WRITE_SIZE) on tools/kprof.py at "
                   "BASELINE config-2 shapes. FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md "
                   "(HBM section) prescribes for gfx950 wide coalesced reads; WRITE_SIZE is taken as is. The TN kernels write f32 split-M "
                   "partials (reduced by gemm_tn_reduce, not counted here).",
           "kernels": kernels,
           "collected": f"tools/profile_round.sh -> tools/pmc_traffic_json.py {fetch_dir} {write_dir}"}
    with open(dst, "w") as f:
        json.dump(doc, f, indent=1)
    if mfma_dir:
        cs = {c: groups(mfma_dir, c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES")}
        util = {}
        for i, (lab, n, k, alg) in enumerate(labels):
            ent = {c: round(cs[c][i][1][-1]) for c in cs if len(cs[c]) == len(labels)}
            if ent.get("GRBM_GUI_ACTIVE"):
                ent["mfma_util"] = round(ent["SQ_VALU_MFMA_BUSY_CYCLES"] / (ent["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
                ent["mfma_busy_cycles_expected"] = 2 * M * n * k // 1024       # one 16x16x32 bf16 MFMA = 32768 flop = 32 SIMD cycles
            util[f"{cs['GRBM_GUI_ACTIVE'][i][0]} {lab} {M}x{n}x{k}"] = ent
        with open(mfma_dst, "w") as f:
            json.dump({"note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES (one pass, "
                               "tools/kprof.py, last launch of each group). Whole-GPU sums per launch; GRBM_GUI_ACTIVE is summed over the 8 XCDs. "
                               "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs), under the profiler (launches run a few "
                               "percent longer than unprofiled).", "kernels": util}, f, indent=1)
        for k_, v in util.items():
            print(f"{k_:80s} mfma_util {v.get('mfma_util')}  lds bank conflicts {v.get('SQ_LDS_BANK_CONFLICT')}")
    for k, v in kernels.items():
        print(f"{k:80s} traffic {v['traffic_bytes'] / 1e6:8.1f} MB  algorithmic {v['algorithmic_bytes'] / 1e6:8.1f} MB")


if __name__ == "__main__":
    main()
