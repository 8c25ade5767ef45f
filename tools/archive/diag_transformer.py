import torch
from vit_pytorch_amd.vit import Transformer
torch.manual_seed(0)
def rel(a,b):
    a=a.detach().double().flatten().cpu(); b=b.detach().double().flatten().cpu(); return ((a-b).norm()/b.norm()).item()
for dim, heads, dim_head, N in ((144,1,144,49),(64,4,16,49),(64,1,64,49),(128,1,128,49),(144,4,36,49),(144,1,144,16),(256,1,256,49), (144,2,72,49)):
    t = Transformer(dim=dim, depth=1, heads=heads, dim_head=dim_head, mlp_dim=dim).to("cuda")
    x = torch.randn(2, N, dim, device="cuda", requires_grad=True)
    r = torch.randn(2, N, dim, device='cuda'); y = t(x); (y * r).sum().backward()
    attn, ff = t.layers[0]
    P = {k: v.detach().double().requires_grad_(True) for k, v in t.named_parameters()}
    xd = x.detach().double().requires_grad_(True)
    ln = lambda z, w, b: torch.nn.functional.layer_norm(z, (dim,), w, b)
    h = ln(xd, P["layers.0.0.norm.weight"], P["layers.0.0.norm.bias"])
    q, k, vv = (h @ P["layers.0.0.to_qkv.weight"].t()).chunk(3, dim=-1)
    sp = lambda z: z.view(2, N, heads, dim_head).transpose(1, 2)
    a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * dim_head ** -0.5, dim=-1) @ sp(vv)
    a = a.transpose(1, 2).reshape(2, N, heads * dim_head)
    if "layers.0.0.to_out.0.weight" in P:
        a = a @ P["layers.0.0.to_out.0.weight"].t() + P["layers.0.0.to_out.0.bias"]
    x1 = a + xd
    h2 = torch.nn.functional.gelu(ln(x1, P["layers.0.1.net.0.weight"], P["layers.0.1.net.0.bias"]) @ P["layers.0.1.net.1.weight"].t() + P["layers.0.1.net.1.bias"])
    x2 = h2 @ P["layers.0.1.net.4.weight"].t() + P["layers.0.1.net.4.bias"] + x1
    ref = ln(x2, P["norm.weight"], P["norm.bias"])
    (ref * r.double()).sum().backward()
    errs = {k: rel(v.grad, P[k].grad) for k, v in t.named_parameters()}
    worst = max(errs, key=errs.get)
    print(f"dim={dim} heads={heads} dh={dim_head} N={N}: out {rel(y,ref):.1e} dx {rel(x.grad, xd.grad):.1e} worst param {worst} {errs[worst]:.1e}", {k: f"{v:.0e}" for k, v in errs.items() if v > 1e-4})
