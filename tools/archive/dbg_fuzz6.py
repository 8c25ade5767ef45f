"""debug: fuzz draw 6 in fp16 gives NaN gradients -- which tensors, which path"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_fuzz_gpu as F
from oracle import vit_oracle as O
from oracle.params import make_images, make_params
from vit_pytorch_amd import SimpleViT, ViT

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 6
kind, cfg, batch = F.draw(seed)
print(kind, cfg, batch)
params = make_params(kind, cfg, 50 + seed); img = make_images(cfg, batch, 1050 + seed)
ref_out, ref_g = O.run_fwd_bwd(kind, cfg, params, img, torch.float32)
for dtype, scale, hook in ((torch.float16, 256.0, False), (torch.float16, 1.0, False), (torch.float16, 256.0, True), (torch.bfloat16, 1.0, False)):
    for env in ({}, {"VITK_GRAD_STREAM": "f32"}, {"VITK_GELU_DG": "0"}, {"VITK_DW_STREAM": "1"}):
        os.environ.update(env)
        m = (ViT if kind == "vit" else SimpleViT)(**cfg); m.load_state_dict(params); m = m.to("cuda", dtype=dtype)
        hs = [l[0].attend.register_forward_hook(lambda *a: None) for l in m.transformer.layers] if hook else []
        out = m(img.to("cuda", dtype=dtype)); (O.loss_fn(out) * scale).backward()
        bad = {k: (int(torch.isnan(p.grad).sum()), int(torch.isinf(p.grad).sum()), float(p.grad.float().abs().nan_to_num(0, 0, 0).max())) for k, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()}
        worst = max((float((p.grad.float() / scale - ref_g[k].cuda()).norm() / (ref_g[k].norm() + 1e-30)), k) for k, p in m.named_parameters() if p.numel() and torch.isfinite(p.grad).all())
        print(dtype, "scale", scale, "hook", hook, env, "non-finite:", bad, "worst finite:", worst)
        for k in env: os.environ.pop(k)
