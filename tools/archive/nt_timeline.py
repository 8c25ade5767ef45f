"""Timeline of ONE wave of one workgroup of the persistent NT GEMM across tile boundaries (s_memtime stamps kept in LDS; the kernel is
built with -DQ_TIMELINE=1 into vit_pytorch_amd/libvitk_tl.so):

    python tools/nt_timeline.py build          # here (hipcc cross-compiles): compiles gemm_nt_persist.hip with the stamps, links libvitk_tl.so
    VITK_LIB=vit_pytorch_amd/libvitk_tl.so python tools/nt_timeline.py [shape ...]      # on the GPU box

Prints, per shape and for waves 0 and 4: the mean K-step in the middle of a tile, and for every tile boundary the epilogue's issue time
(tag 2 -> 3), the time from the end of the epilogue to the start of the next tile's K-steps 0..5, i.e. where the store drain is paid."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build():
    from vit_pytorch_amd import _build as B
    B.build_lib(verbose=False)
    bdir = os.path.join(B.CSRC, "build", "tl")
    os.makedirs(bdir, exist_ok=True)
    o = os.path.join(bdir, "gemm_nt_persist.o")
    subprocess.run([B._hipcc(), *B.FLAGS, "-DQ_TIMELINE=1", "-c", os.path.join(B.CSRC, "gemm_nt_persist.hip"), "-o", o], check=True)
    objs = [o if s == "gemm_nt_persist.hip" else os.path.join(B.CSRC, "build", s.replace(".hip", ".o")) for s in B.SOURCES]
    lib = os.path.join(B.HERE, "libvitk_tl.so")
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-ldl"], check=True)
    print(lib)


def main():
    import torch
    from vit_pytorch_amd import kernels as K, _lib as L
    dev = "cuda"; BF = torch.bfloat16
    M = 50432
    shapes = {"qkv": (2304, 768, L.EPI_NONE), "ff1": (3072, 768, L.EPI_BIAS_GELU), "dx_ff1": (768, 3072, L.EPI_NONE), "out16": (768, 768, L.EPI_RESID16)}
    want = sys.argv[1:] or list(shapes)
    st = torch.zeros(4096, dtype=torch.int64, device=dev)
    for name in want:
        n, k, epi = shapes[name]
        A = torch.randn(M, k, device=dev).to(BF); W = (torch.randn(n, k, device=dev) * k ** -0.5).to(BF)
        Wp = torch.empty(K.pack_w_nt_bytes(n, k) // 2, dtype=BF, device=dev); K.pack_w_nt(W, k, n, k, Wp, None)
        C = torch.zeros(M, n, dtype=BF, device=dev); aux = torch.empty(M, n, dtype=BF, device=dev); bias = torch.randn(n, device=dev).to(BF)
        kw = dict(bias=bias if epi != L.EPI_NONE else None, resid=C if epi == L.EPI_RESID16 else None, aux=aux if epi == L.EPI_BIAS_GELU else None)
        for _ in range(2):
            K.gemm_nt_bf16(A, k, Wp, 0, C, n, M, n, k, epi, **kw)
        for wave in (0, 4):
            st.zero_()
            os.environ["VITK_NTP_STAMPS"] = str(st.data_ptr()); os.environ["VITK_NTP_DBG"] = str(256 * wave)
            K.gemm_nt_bf16(A, k, Wp, 0, C, n, M, n, k, epi, **kw)
            torch.cuda.synchronize()
            os.environ.pop("VITK_NTP_STAMPS"); os.environ.pop("VITK_NTP_DBG")
            t = st.cpu().tolist()
            cnt = t[0]
            ev = [(v >> 4, v & 15) for v in t[1:1 + cnt]]
            # split into tiles at tag 2 (epilogue start)
            nt = k // 32
            ks = [c for c, tag in ev if tag == 1]
            steps = [b - a for a, b in zip(ks, ks[1:])]
            mid = sorted(steps)[len(steps) // 2] if steps else 0
            print(f"== {name} N={n} K={k} wave {wave}: {cnt} stamps, median K-step {mid} cycles ({nt} K-steps per tile)")
            i = 0
            tile = 0
            while i < len(ev):
                if ev[i][1] == 2:
                    t2 = ev[i][0]
                    t3 = next((c for c, tag in ev[i:] if tag == 3), None)
                    nxt = [c for c, tag in ev[i:] if tag == 1][:7]
                    prev = [c for c, tag in ev[:i] if tag == 1][-3:]
                    if t3 is not None and len(nxt) >= 6 and len(prev) == 3:
                        print(f"  tile {tile}: last K-steps {prev[1] - prev[0]} {prev[2] - prev[1]} | last K-step start -> epilogue start {t2 - prev[2]} | epilogue issue {t3 - t2} | "
                              f"epilogue end -> K-step 0 start {nxt[0] - t3} | next K-steps " + " ".join(str(b - a) for a, b in zip(nxt, nxt[1:])))
                    tile += 1
                i += 1


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        main()
