"""GPU: HIP-graph capture of the forward and of forward+backward (vit_pytorch_amd/graphs.py): replays are bit-identical to
eager calls on new inputs, and the launch path is gone at small batch."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

from vit_pytorch_amd import ViT  # noqa: E402
from vit_pytorch_amd.graphs import GraphedForward, GraphedForwardBackward  # noqa: E402

DEV = "cuda"
CFG = dict(image_size=64, patch_size=8, num_classes=10, dim=128, depth=3, heads=2, mlp_dim=256)


def test_graphed_forward_matches_eager_and_is_faster_at_small_batch():
    torch.manual_seed(0)
    m = ViT(**CFG).to(DEV, dtype=torch.bfloat16).eval()
    g = torch.Generator(device=DEV).manual_seed(1)
    xs = [torch.randn(2, 3, 64, 64, device=DEV, generator=g).to(torch.bfloat16) for _ in range(3)]
    fwd = GraphedForward(m, xs[0])
    for x in xs:
        with torch.no_grad():
            ref = m(x)
        assert torch.equal(fwd(x), ref)
    def timed(fn, n=50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
    with torch.no_grad():
        t_eager = timed(lambda: m(xs[0]))
    t_graph = timed(lambda: fwd(xs[0]))
    print(f"depth-3 ViT, batch 2: eager {t_eager * 1e6:.0f} us, graph {t_graph * 1e6:.0f} us")
    assert t_graph < t_eager
    with pytest.raises(Exception):
        fwd(torch.zeros(3, 3, 64, 64, device=DEV, dtype=torch.bfloat16))      # captured for batch 2


def test_graphed_forward_backward_matches_eager():
    import copy
    torch.manual_seed(0)
    m = ViT(**CFG).to(DEV, dtype=torch.bfloat16)
    m_ref = copy.deepcopy(m)
    loss_fn = lambda out, y: torch.nn.functional.cross_entropy(out.float(), y)
    g = torch.Generator(device=DEV).manual_seed(2)
    data = [(torch.randn(4, 3, 64, 64, device=DEV, generator=g).to(torch.bfloat16), torch.randint(0, 10, (4,), device=DEV, generator=g)) for _ in range(3)]
    step = GraphedForwardBackward(m, loss_fn, *data[0])
    for x, y in data:
        loss_g = step(x, y)
        m_ref.zero_grad(set_to_none=True)
        loss_e = loss_fn(m_ref(x), y)
        loss_e.backward()
        assert torch.equal(loss_g, loss_e.detach())
        for (k, p), (_, q) in zip(m.named_parameters(), m_ref.named_parameters()):
            assert torch.equal(p.grad, q.grad), k
        m.zero_grad(set_to_none=True)                # the next replay re-attaches the captured gradient tensors
