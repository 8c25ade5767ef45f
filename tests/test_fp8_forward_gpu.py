"""GPU: opt-in e4m3 forward GEMM operands in the fused engine (vit_pytorch_amd/fp8.py): delayed scaling bookkeeping, and the
model against its own 16-bit run and the f32 oracle.  Tolerances are those of a 3-bit-mantissa operand format: a few
percent on logits and gradients (stated per assert)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vit_oracle as O  # noqa: E402
from oracle.params import make_images, make_params  # noqa: E402
from vit_pytorch_amd import SimpleViT, ViT  # noqa: E402
from vit_pytorch_amd.fp8 import enable_fp8_forward  # noqa: E402

DEV = "cuda"
CFG = dict(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=2, heads=12, mlp_dim=3072)


def rel(a, b):
    a = a.detach().double().flatten().cpu(); b = b.detach().double().flatten().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("kind", ["vit", "simple_vit"])
def test_fp8_forward_against_16bit_and_oracle(kind, monkeypatch):
    monkeypatch.setenv("VITK_FWD_STREAM", "f32")      # the fp8 path keeps the float32 stream: its recording step equals the 16-bit run under that stream
    monkeypatch.setenv("VITK_GELU_DG", "0")           # ... and saves the pre-activation (the 16-bit default stores the gelu' factor instead)
    params = make_params(kind, CFG, 7)
    img = make_images(CFG, 8, 1007)                      # M = 8 * 197 = 1576 rows: the 256-row kernel's range
    cls = ViT if kind == "vit" else SimpleViT

    def build():
        m = cls(**CFG); m.load_state_dict(params, strict=True)
        return m.to(DEV, dtype=torch.bfloat16)

    def run(m):
        m.zero_grad(set_to_none=True)
        out = m(img.to(DEV, dtype=torch.bfloat16))
        O.loss_fn(out).backward()
        return out.detach().clone(), torch.cat([p.grad.float().flatten() for p in m.parameters() if p.numel()])

    o16, g16 = run(build())
    m8 = enable_fp8_forward(build())
    st = m8.transformer._fp8
    assert not st.ready and float(st.scales.abs().sum()) == 0.0
    o_rec, g_rec = run(m8)                               # first call: 16-bit GEMMs, amax recorded
    assert torch.equal(o_rec, o16) and st.ready
    from vit_pytorch_amd.fp8 import SLOTS_PER_LAYER
    act = st.scales.view(CFG["depth"], SLOTS_PER_LAYER, 2)[:, :4]        # the four activation slots of a layer (gradient slots: enable_fp8)
    assert (act[..., 0] > 0).all() and torch.allclose(act[..., 0] * act[..., 1], torch.ones_like(act[..., 0]), rtol=1e-5)
    assert float(st.scales.view(CFG["depth"], SLOTS_PER_LAYER, 2)[:, 4:].abs().sum()) == 0.0 and not st.backward
    assert int(st.amax.abs().sum()) == 0                 # records folded and reset
    o8, g8 = run(m8)                                     # second call: e4m3 operands for QKV / out-projection / FF1 / FF2
    assert not torch.equal(o8, o16)
    e, g = rel(o8, o16), rel(g8, g16)
    ref_out, _ = O.run_fwd_bwd(kind, CFG, params, img, torch.float32)
    print(f"{kind}: fp8-forward vs 16-bit: logits {e:.2e} grads {g:.2e}; vs f32 oracle: {rel(o8, ref_out):.2e} (16-bit run: {rel(o16, ref_out):.2e})")
    assert e < 6e-2 and g < 1.2e-1 and rel(o8, ref_out) < 6e-2
    n_w = len(st._w)
    run(m8)
    assert len(st._w) == n_w                             # unchanged weights are not re-quantised
    with torch.no_grad():
        m8.transformer.layers[0][1].net[1].weight.mul_(1.0)          # in-place update -> version bump -> re-quantised
    v0 = st._w[id(m8.transformer.layers[0][1].net[1].weight)][0]
    run(m8)
    assert st._w[id(m8.transformer.layers[0][1].net[1].weight)][0] != v0
