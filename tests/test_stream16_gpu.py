"""GPU: the 16-bit FORWARD residual stream (opt-in, VITK_FWD_STREAM=16; DESIGN section 7 "next", item 5).

Kernel: the RESID16 epilogue of the persistent NT GEMM -- C16 = T(resid16 + A W^T + bias), the sum formed in f32 -- against float64.
Model: ViT at BASELINE config 2's width against the goldens the reference produced, held to the SAME gate as the default (f32-stream)
bf16 run: error vs the reference's f32 outputs <= 1.5 x the error of the reference's own bf16 run + 1e-3."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vit_oracle as O  # noqa: E402
from oracle.params import WIDE_CASES, make_images, make_params, sample_index  # noqa: E402
from vit_pytorch_amd import ViT, _lib as L, kernels as K, ops  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("M,N,Kd,with_bias", [(6304, 768, 768, True), (6304, 768, 3072, True), (2308, 1280, 5120, False), (1182, 1024, 1024, True)])
def test_resid16_epilogue_against_float64(M, N, Kd, with_bias):
    assert K.gemm_nt_plan(M, N, Kd, N)["persistent"]
    g = torch.Generator(device=DEV).manual_seed(8)
    A = torch.randn(M, Kd, device=DEV, generator=g).to(BF)
    W = (torch.randn(N, Kd, device=DEV, generator=g) * Kd ** -0.5).to(BF)
    bias = torch.randn(N, device=DEV, generator=g).to(BF) if with_bias else None
    resid = (torch.randn(M, N, device=DEV, generator=g) * torch.linspace(0.5, 3.0, N, device=DEV)).to(BF)     # asymmetric in the column index
    out = torch.full((M, N), float("nan"), dtype=BF, device=DEV)
    K.gemm_nt_bf16(A, Kd, W, Kd, out, N, M, N, Kd, L.EPI_RESID16, bias=bias, resid=resid)
    ref = resid.double() + A.double() @ W.double().t() + (bias.double() if with_bias else 0)
    assert torch.isfinite(out.float()).all()
    assert rel(out, ref) < 3e-3, rel(out, ref)                              # one bf16 rounding of the sum
    # K-blocked weights (what the engine passes) give the same bits
    Wp, ldw = ops.nt_weight(torch.nn.Parameter(W), M, False)
    if ldw == 0:
        out2 = torch.empty_like(out)
        K.gemm_nt_bf16(A, Kd, Wp, 0, out2, N, M, N, Kd, L.EPI_RESID16, bias=bias, resid=resid)
        assert torch.equal(out2, out)
    # against the f32-stream epilogue rounded once: identical sums
    out32 = torch.empty(M, N, device=DEV)
    K.gemm_nt_bf16(A, Kd, W, Kd, out32, N, M, N, Kd, L.EPI_RESID, bias=bias, resid=resid.float())
    assert torch.equal(out, out32.to(BF))
    with pytest.raises(L.VitkError):
        K.gemm_nt_bf16(A[:512], Kd, W, Kd, out[:512], N, 512, N, Kd, L.EPI_RESID16, bias=bias, resid=resid[:512])     # not a persistent-kernel shape


def test_16bit_forward_stream_model_vs_reference_golden(monkeypatch):
    name = "vit_b16_width"
    case = WIDE_CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    params = make_params(case["kind"], case["cfg"], case["seed"])
    img = make_images(case["cfg"], case["batch"], case["seed"] + 1000).to(DEV, dtype=BF)
    ref_logits = torch.from_numpy(gold["logits"])
    keys = [k for k in params if params[k].numel()]

    def run():
        m = ViT(**case["cfg"]); m.load_state_dict(params, strict=True)
        m = m.to(DEV, dtype=BF)
        out = m(img)
        O.loss_fn(out).backward()
        named = dict(m.named_parameters())
        mine, ref, ref16 = [], [], []
        for k in keys:
            gk = named[k].grad.detach().float().flatten().cpu()
            idx = torch.from_numpy(sample_index(gk.numel(), case.get("sample", 4096)))
            mine.append(gk[idx]); ref.append(torch.from_numpy(gold["gsample::" + k]).float())
            ref16.append(torch.from_numpy(gold["bf16::gsample::" + k]).float())
        return out.detach().clone(), rel(out, ref_logits), rel(torch.cat(mine), torch.cat(ref)), rel(torch.cat(ref16), torch.cat(ref))

    monkeypatch.setenv("VITK_FWD_STREAM", "f32")
    assert not ops.fwd_stream_16(BF)
    o32, e32, g32, g_ref16 = run()
    monkeypatch.setenv("VITK_FWD_STREAM", "16")
    assert ops.fwd_stream_16(BF)
    o16, e16, g16, _ = run()
    e_ref16 = rel(torch.from_numpy(gold["bf16::logits"]), ref_logits)
    print(f"16-bit forward stream: logits {e16:.2e} grads {g16:.2e}; f32 stream {e32:.2e} / {g32:.2e}; reference's own bf16 {e_ref16:.2e} / {g_ref16:.2e}")
    assert not torch.equal(o16, o32)
    assert e16 <= 1.5 * e_ref16 + 1e-3 and g16 <= 1.5 * g_ref16 + 1e-3, (e16, e_ref16, g16, g_ref16)
