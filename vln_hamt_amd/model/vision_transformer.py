"""ViT-B/16 panorama backbone (reference: pretrain_src/model/vision_transformer.py, a timm copy).

Same class surface and parameter names as the reference's ``VisionTransformer`` without classifier head
(``patch_embed.proj``, ``cls_token``, ``pos_embed``, ``blocks.N.{norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2}``,
``norm``), so timm / reference checkpoints load with ``load_state_dict``.  ``forward_features`` (vision_transformer.py:
335-348) runs on the HAMT HIP kernels:

* PatchEmbed's conv (kernel = stride = patch, :216-221) = ``hamt_patchify`` (image -> patch rows, bf16 or fp32) + one GEMM
  against ``proj.weight.view(D, C*P*P)``;
* cls token + position embedding (:337-342) = row gather-add;
* pre-LN blocks (:181-198): LayerNorm (eps 1e-6) -> fused qkv Linear -> attention over the packed [q|k|v] buffer
  (scale head_dim^-0.5, :160, dropout on the probabilities) -> proj -> residual add; LayerNorm -> fc1 + erf-GELU ->
  fc2 -> residual add;
* final LayerNorm, cls row.

bf16 mode with the branch dropouts off: one autograd Function per half block (``blocks_preln.py``: bf16 images between
the kernels, residual adds in GEMM epilogues / in the LayerNorm-backward kernel, queued weight gradients).  Otherwise
(fp32 mode, branch dropout on) the fine-grained Functions of ``ops.py``.  Every op has its backward, so images that need
gradients -- observation / history images in image_vilmodel.py:40-59 -- train end to end; the 36-view panorama pass runs
under ``torch.no_grad()`` as in the reference.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from .. import _lib as L
from .. import blocks_preln, ops, streams
from ..ops import _p, _stream


class Mlp(nn.Module):
    """vision_transformer.py:135-151"""

    def __init__(self, in_features, hidden_features, drop=0.0, prec="bf16"):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)
        self.drop = nn.Dropout(drop)
        self.prec = prec

    def forward(self, x, residual=None):
        """mlp(x) (+ residual).  With dropout off the residual add rides in fc2's GEMM epilogue."""
        p = float(self.drop.p) if self.training else 0.0
        h = ops.linear(x, self.fc1.weight, self.fc1.bias, ops.ACT_GELU, self.prec)
        h = ops.dropout(h, p, self.training)
        if residual is not None and p == 0.0:
            return ops.linear(h, self.fc2.weight, self.fc2.bias, ops.ACT_NONE, self.prec, residual=residual)
        y = ops.dropout(ops.linear(h, self.fc2.weight, self.fc2.bias, ops.ACT_NONE, self.prec), p, self.training)
        return y if residual is None else ops.add3(residual, y)


class Attention(nn.Module):
    """vision_transformer.py:154-178 (fused qkv Linear; heads are column blocks of the packed buffer)"""

    def __init__(self, dim, num_heads, attn_drop=0.0, proj_drop=0.0, prec="bf16"):
        super().__init__()
        if (dim // num_heads) != 64:
            raise ValueError("the attention kernels are built for head_dim 64 (ViT-B/16: 768 / 12)")
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.prec = prec

    def forward(self, x, residual=None):
        """attn(x) (+ residual).  With dropout off the residual add rides in the projection's GEMM epilogue."""
        B, N, D = x.shape
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias, ops.ACT_NONE, self.prec).view(B * N, 3 * D)
        pa = float(self.attn_drop.p) if self.training else 0.0
        pp = float(self.proj_drop.p) if self.training else 0.0
        ctx = ops.attention(qkv, None, None, B, self.num_heads, pa, self.prec)
        if residual is not None and pp == 0.0:
            return ops.linear(ctx, self.proj.weight, self.proj.bias, ops.ACT_NONE, self.prec, residual=residual.reshape(B * N, D)).view(B, N, D)
        y = ops.dropout(ops.linear(ctx, self.proj.weight, self.proj.bias, ops.ACT_NONE, self.prec), pp, self.training).view(B, N, D)
        return y if residual is None else ops.add3(residual, y)


class Block(nn.Module):
    """vision_transformer.py:181-198 (drop_path = 0 as in image_vilmodel.py:30-33)"""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, drop=0.0, attn_drop=0.0, prec="bf16"):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads, attn_drop, drop, prec)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), drop, prec)

    def forward(self, x):
        a, m = self.attn, self.mlp
        pd = float(a.proj_drop.p) if self.training else 0.0
        md = float(m.drop.p) if self.training else 0.0
        if x.dim() == 3 and blocks_preln.usable(a.prec, x):       # one autograd Function per half block
            pa = float(a.attn_drop.p) if self.training else 0.0
            x = blocks_preln.PreLnAttnFn.apply(x, a.num_heads, pa, pd, self.norm1.eps, self.norm1.weight, self.norm1.bias,
                                               a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias)
            return blocks_preln.PreLnMlpFn.apply(x, md, self.norm2.eps, self.norm2.weight, self.norm2.bias,
                                                 m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, torch.is_grad_enabled())
        x = self.attn(ops.layer_norm(x, None, self.norm1, want16=True), residual=x)
        return self.mlp(ops.layer_norm(x, None, self.norm2, want16=True), residual=x)


class PatchEmbed(nn.Module):
    """vision_transformer.py:201-223; the parameter keeps the conv layout [D][C][P][P]"""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, prec="bf16"):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.patch_grid = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.patch_grid[0] * self.patch_grid[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)   # parameter container only
        self.prec = prec

    def forward(self, x):
        if not x.is_cuda:
            raise L.HamtError("PatchEmbed: input must live on the GPU (no CPU fallback)")
        B, Cc, H, W = x.shape
        assert (H, W) == self.img_size, f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        P = self.patch_size[0]
        rows, K = B * self.num_patches, Cc * P * P
        x = x.contiguous().float()
        patches = torch.empty(rows, K, dtype=torch.float32, device=x.device)
        L.check(L.load().hamt_patchify(B, Cc, H, W, P, _p(x), _p(patches), K, L.HAMT_F32, rows, _stream()), "hamt_patchify")
        y = ops.linear(patches, self.proj.weight.view(self.proj.weight.shape[0], K), self.proj.bias, ops.ACT_NONE, self.prec)
        return y.view(B, self.num_patches, -1)


class VisionTransformer(nn.Module):
    """vision_transformer.py:226-362 without distillation token / classifier head (`num_classes=0`)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 drop_rate=0.0, attn_drop_rate=0.0, hamt_precision="bf16"):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, hamt_precision)
        n_tok = self.patch_embed.num_patches + 1
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n_tok, embed_dim))
        self.pos_drop = nn.Dropout(drop_rate)
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_ratio, drop_rate, attn_drop_rate, hamt_precision)
                                      for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)

    def forward_features(self, x):
        streams.gate(self.cls_token, self.pos_embed, self.norm)                      # (called as a method: no module hook in front)
        x = self.patch_embed(x)                                                      # (B, N, D)
        B, N, D = x.shape
        x = torch.cat([self.cls_token.expand(B, -1, -1), x], 1).view(B * (N + 1), D)  # :337-339
        idx = torch.arange(N + 1, device=x.device).repeat(B)
        x = ops.gather_rows(self.pos_embed.view(N + 1, D), idx, base=x)             # x + pos_embed (:342)
        x = ops.dropout(x, float(self.pos_drop.p), self.training).view(B, N + 1, D)
        x = self.blocks(x)
        x = ops.layer_norm(x, None, self.norm)
        return x[:, 0]

    def forward(self, x):
        return self.forward_features(x)


def vit_base_patch16_224(**kw):
    """ViT-B/16 (vision_transformer.py:486-493) without classifier head."""
    return VisionTransformer(img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, **kw)
