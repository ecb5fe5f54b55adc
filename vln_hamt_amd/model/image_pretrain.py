"""Image-input proxy-task pretraining (BASELINE config 4; reference: pretrain_src/model/image_vilmodel.py:23-126 and
image_pretrain.py:18-208): the ViT-B/16 backbone turns raw views into the 768-d features the feature-input model
consumes, everything after that IS the feature-input model.

`MultiStepNavImagePreTraining(config)` keeps the reference's parameter names (`bert.vision_backbone.*` next to the
`bert.*` / head names of MultiStepNavCMTPreTraining) and its batch keys (`hist_images` (N,T,3,H,W), `hist_pano_images`
(N,T,P,3,H,W), `ob_images` (N,V,3,H,W), `ob_v_exists`, `hist_mrc_masks` ... image_pretrain.py:47-100):

* history-step and observation images go through the backbone WITH gradient, the T x 36 panorama views without
  (image_vilmodel.py:44-59: "due to memory issue, we cannot propagate to pano images in the history");
* MRC zero-fills the features of masked steps (:84-86), missing observation views are zero-filled (:101-102), the STOP
  token is a zero feature row appended after the views (:104-106).
"""
from __future__ import annotations

import torch

from .pretrain_cmt import MultiStepNavCMTPreTraining
from .vision_transformer import VisionTransformer


class MultiStepNavImagePreTraining(MultiStepNavCMTPreTraining):
    def __init__(self, config, vit_kwargs=None):
        super().__init__(config)
        kw = dict(img_size=224, patch_size=16, embed_dim=config.image_feat_size, depth=12, num_heads=config.image_feat_size // 64,
                  drop_rate=config.hidden_dropout_prob, attn_drop_rate=config.attention_probs_dropout_prob,     # image_vilmodel.py:27-30
                  hamt_precision=getattr(config, "hamt_precision", "bf16"))
        kw.update(vit_kwargs or {})
        self.bert.vision_backbone = VisionTransformer(**kw)

    def forward_vision_backbone(self, images, detach=False):
        """image_vilmodel.py:40-59"""
        is_pano = images.dim() == 6
        lead = images.shape[:3] if is_pano else images.shape[:2]
        flat = images.reshape(-1, *images.shape[-3:])
        if is_pano:
            with torch.no_grad():
                feats = self.bert.vision_backbone.forward_features(flat)
        else:
            feats = self.bert.vision_backbone.forward_features(flat)
        feats = feats.reshape(*lead, -1)
        return feats.detach() if detach else feats

    def forward(self, batch, task, compute_loss=True):
        fb = dict(batch)
        if fb.get("hist_images") is not None:
            hf = self.forward_vision_backbone(fb["hist_images"])
            pf = self.forward_vision_backbone(fb["hist_pano_images"], detach=True)
            m = fb.get("hist_mrc_masks")
            if m is not None and task.startswith("mrc"):                                   # image_vilmodel.py:84-86
                hf = hf.masked_fill(m.unsqueeze(-1), 0)
                pf = pf.masked_fill(m.unsqueeze(-1).unsqueeze(-1), 0)
            fb["hist_img_fts"], fb["hist_pano_img_fts"] = hf, pf
        else:
            fb["hist_img_fts"] = fb["hist_pano_img_fts"] = None
        if fb.get("ob_images") is not None and not (task.startswith("mlm") or task.startswith("mrc") or task.startswith("itm")):
            of = self.forward_vision_backbone(fb["ob_images"])
            ex = fb.get("ob_v_exists")
            if ex is not None:                                                              # :101-102
                of = of.masked_fill(ex.logical_not().unsqueeze(-1), 0)
            fb["ob_img_fts"] = torch.cat([of, of.new_zeros(of.shape[0], 1, of.shape[2])], 1)   # STOP token (:104-106)
        return super().forward(fb, task, compute_loss)
