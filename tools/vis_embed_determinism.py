import sys; sys.path.insert(0, "/root/repo")
import torch
from vln_hamt_amd import ops
nn = torch.nn
torch.manual_seed(0)
dev = "cuda"
for (M, K, H) in ((16, 64, 128), (148, 64, 128), (576, 64, 128), (2368, 768, 768)):
    mods = nn.ModuleList([nn.Linear(K, H), nn.Linear(4, H), nn.LayerNorm(H, eps=1e-12), nn.LayerNorm(H, eps=1e-12)]).to(dev)
    img, ang, gy = torch.randn(M, K, device=dev), torch.randn(M, 4, device=dev), torch.randn(M, H, device=dev)
    ref = None
    bad = 0
    for it in range(40):
        for p in mods.parameters(): p.grad = None
        x = img.clone().requires_grad_(True)
        y = ops.vis_embed(x, ang, mods[0], mods[2], mods[1], mods[3], "bf16", want16=True)
        y.backward(gy)
        torch.cuda.synchronize()
        cur = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in mods.parameters()]
        if ref is None: ref = cur
        else:
            for i, (a, b) in enumerate(zip(cur, ref)):
                if not torch.equal(a, b):
                    bad += 1; print("  mismatch", (M, K, H), "iter", it, "tensor", i, float((a - b).abs().max()))
        # garbage in the allocator between iterations
        junk = torch.full((1 << 20,), float("nan"), device=dev); del junk
    print((M, K, H), "mismatches:", bad)
