"""MI355X-native mirror of ``pretrain_src/model/pretrain_cmt.py``: the six proxy-task heads and their
losses on top of the HIP trunk.  Same class names / parameter names / ``forward(batch, task, compute_loss)``
contract (losses with reduction='none'; MRC and ITM return 2-tuples when compute_loss is False).
"""
from __future__ import annotations

import os
from collections import defaultdict

import torch
from torch import nn

from .. import ops
from ..modeling import precision_of
from .vilmodel import BertLayerNorm, BertOnlyMLMHead, BertPreTrainedModel, NavPreTrainedModel


# The five small prediction heads compute in exact fp32 (v_mfma_f32_16x16x4_f32) in bf16 mode too: they sit BEHIND the trunk, where
# an error is no longer averaged by anything -- their LayerNorm re-scales whatever rounding the first dense layer added -- and they
# are a few hundred to 2 400 rows of 768: ~0.02 ms per step averaged over the task mix.  Measured on the R2R-canon model at per-GPU
# batch 16 (tests/golden/canon_multi.npz, three weight seeds): ITM logits 1.06e-2 / 1.35e-2 / 1.00e-2 off the reference with bf16
# heads -- over north_star's 1e-2 -- against 8.6e-3 / 9.5e-3 / 5.2e-3 with fp32 heads.  HAMT_HEADS_BF16=1 restores bf16 heads.
HEADS_FP32 = os.environ.get("HAMT_HEADS_BF16") is None
TXT_PACK = os.environ.get("HAMT_NO_TXT_PACK") is None      # use a batch's text packing plan when it carries one


class _MlpHead(nn.Module):
    """Linear -> ReLU -> LayerNorm(1e-12) -> [Dropout] -> Linear, kept as nn.Sequential `net` so the
    parameter names are net.0 / net.2 / net.4 (or net.3) as in pretrain_cmt.py:13-71."""

    def __init__(self, in_size, hidden_size, out_size, dropout_rate=None, prec="bf16"):
        super().__init__()
        mods = [nn.Linear(in_size, hidden_size), nn.ReLU(), BertLayerNorm(hidden_size, eps=1e-12)]
        if dropout_rate is not None:
            mods.append(nn.Dropout(dropout_rate))
        mods.append(nn.Linear(hidden_size, out_size))
        self.net = nn.Sequential(*mods)
        self.prec = prec
        if HEADS_FP32:      # read as fp32 masters, never through the bf16 shadow: the fp32-read region of the optimizer's arena
            mods[0].weight._hamt_fp32_read = True      # (what a sharded data-parallel update all-gathers in fp32, optim.adamw.shadow_only)
            mods[-1].weight._hamt_fp32_read = True

    def forward(self, x):
        lin0, ln, last = self.net[0], self.net[2], self.net[-1]
        p = float(self.net[3].p) if (len(self.net) == 5 and self.training) else 0.0
        prec = "fp32" if HEADS_FP32 else self.prec
        if prec == "fp32" and x.dtype != torch.float32:
            x = x.float()
        h = ops.linear(x, lin0.weight, lin0.bias, ops.ACT_RELU, prec)
        h = ops.layer_norm(h, None, ln, p_post=p)
        return ops.linear(h, last.weight, last.bias, ops.ACT_NONE, prec)


class NextActionPrediction(_MlpHead):          # pretrain_cmt.py:13-23
    def __init__(self, hidden_size, dropout_rate, prec="bf16"):
        super().__init__(hidden_size, hidden_size, 1, dropout_rate, prec)


class NextActionRegression(_MlpHead):          # pretrain_cmt.py:25-35
    def __init__(self, hidden_size, dropout_rate, prec="bf16"):
        super().__init__(hidden_size, hidden_size, 3, dropout_rate, prec)


class SpatialRelRegression(_MlpHead):          # pretrain_cmt.py:37-47
    def __init__(self, hidden_size, dropout_rate, prec="bf16"):
        super().__init__(hidden_size * 2, hidden_size, 2, dropout_rate, prec)


class RegionClassification(_MlpHead):          # pretrain_cmt.py:49-60  (MRC-kl)
    def __init__(self, hidden_size, label_dim, prec="bf16"):
        super().__init__(hidden_size, hidden_size, label_dim, None, prec)


class ItmPrediction(_MlpHead):                 # pretrain_cmt.py:62-71
    def __init__(self, hidden_size, prec="bf16"):
        super().__init__(hidden_size, hidden_size, 1, None, prec)


class MultiStepNavCMTPreTraining(BertPreTrainedModel):
    """pretrain_cmt.py:73-262."""
    _hamt_container = True      # (optim.AdamW.attach) forward reads parameters only through self.bert / the heads' __call__

    def __init__(self, config):
        super().__init__(config)
        self.config = config
        self.bert = NavPreTrainedModel(config)
        prec = precision_of(config)
        H = config.hidden_size
        if 'mlm' in config.pretrain_tasks:
            self.mlm_head = BertOnlyMLMHead(config)
        if 'sap' in config.pretrain_tasks:
            self.next_action = NextActionPrediction(H, config.pred_head_dropout_prob, prec)
        if 'sar' in config.pretrain_tasks:
            self.regress_action = NextActionRegression(H, config.pred_head_dropout_prob, prec)
        if 'sprel' in config.pretrain_tasks:
            self.sprel_head = SpatialRelRegression(H, config.pred_head_dropout_prob, prec)
        if 'mrc' in config.pretrain_tasks:
            self.image_classifier = RegionClassification(H, config.image_prob_size, prec)
        if 'itm' in config.pretrain_tasks:
            self.itm_head = ItmPrediction(H, prec)
        self.init_weights()
        self.tie_weights()

    def tie_weights(self):
        if 'mlm' in self.config.pretrain_tasks:
            self._tie_or_clone_weights(self.mlm_head.predictions.decoder, self.bert.embeddings.word_embeddings)

    def forward(self, batch, task, compute_loss=True):
        batch = defaultdict(lambda: None, batch)
        if batch['txt_ids'] is not None:
            # optional packing plan of a ragged batch (synth.make_batch / data.collate): the text-only layers then skip the padding
            # (vilmodel.NavPreTrainedModel._text); it rides on the id tensor so that the trunk keeps the reference's signature
            pk = batch['txt_pack_idx']
            batch['txt_ids']._hamt_pack = (pk, batch['txt_cu'], batch['txt_unpack_idx']) if (pk is not None and TXT_PACK) else None
        hist = (batch['txt_ids'], batch['txt_masks'], batch['hist_img_fts'], batch['hist_ang_fts'],
                batch['hist_pano_img_fts'], batch['hist_pano_ang_fts'], batch['hist_masks'])
        ob = (batch['ob_img_fts'], batch['ob_ang_fts'], batch['ob_nav_types'], batch['ob_masks'])
        if task.startswith('mlm'):
            return self.forward_mlm(*hist, batch['txt_labels'], compute_loss, label_idx=batch['txt_label_idx'])
        elif task.startswith('sap'):
            return self.forward_sap(*hist, *ob, batch['ob_action_viewindex'], compute_loss)
        elif task.startswith('sar'):
            return self.forward_sar(*hist, *ob, batch['ob_action_angles'], batch['ob_progress'], compute_loss)
        elif task.startswith('sprel'):
            return self.forward_sprel(*hist, *ob, batch['sp_anchor_idxs'], batch['sp_targets'], compute_loss)
        elif task.startswith('mrc'):
            return self.forward_mrc(*hist, batch['hist_mrc_masks'], batch['hist_img_probs'], compute_loss,
                                    mrc_idx=batch['hist_mrc_idx'])
        elif task.startswith('itm'):
            return self.forward_itm(*hist, 4, compute_loss, neg_idxs=batch['itm_neg_idxs'],
                                    shuffled_pos_ids=batch['itm_shuffled_pos_ids'])
        else:
            raise ValueError('invalid task')

    # ---- A15
    def forward_mlm(self, txt_ids, txt_masks, hist_img_fts, hist_ang_fts, hist_pano_img_fts, hist_pano_ang_fts,
                    hist_masks, txt_labels, compute_loss, label_idx=None):
        """`label_idx` (optional, int64 flat positions of the masked tokens in row-major order, e.g. built by the
        collate on the host) avoids the device->host sync of boolean indexing, so the step is graph-capturable."""
        if label_idx is None:
            label_idx = (txt_labels != -1).reshape(-1).nonzero(as_tuple=False).squeeze(1)
        # (only the masked rows of the text output are read: the vision side of the last cross-modal layer and the other rows of its text-side
        # feed-forward block are dead code, vilmodel.LXRTXLayer.forward)
        txt_embeds, _, _ = self.bert(txt_ids, txt_masks, hist_img_fts, hist_ang_fts, hist_pano_img_fts,
                                     hist_pano_ang_fts, hist_masks, None, None, None, None, need="lang", lang_rows=label_idx)
        masked_output = txt_embeds if txt_embeds.dim() == 2 else ops.gather_rows(txt_embeds.reshape(-1, txt_embeds.size(-1)), label_idx, unique=True)
        prediction_scores = self.mlm_head(masked_output)
        if compute_loss:
            return ops.cross_entropy(prediction_scores, txt_labels.reshape(-1).index_select(0, label_idx))
        return prediction_scores

    def _compute_masked_hidden(self, hidden, mask):
        """rows of `hidden` where `mask` is set, in row-major order (pretrain_cmt.py:161-165): the index list
        is integer work done once, the row gather (and its scatter in backward) is a HIP kernel."""
        idx = mask.reshape(-1).nonzero(as_tuple=False).squeeze(1)
        return ops.gather_rows(hidden.reshape(-1, hidden.size(-1)), idx, unique=True)

    # ---- A16
    def forward_sap(self, txt_ids, txt_masks, hist_img_fts, hist_ang_fts, hist_pano_img_fts, hist_pano_ang_fts,
                    hist_masks, ob_img_fts, ob_ang_fts, ob_nav_types, ob_masks, act_labels, compute_loss):
        cls_rows = ops.const_index("arange_mul", int(txt_ids.size(0)), int(txt_ids.size(1)), device=txt_ids.device)
        txt_embeds, hist_embeds, ob_embeds = self.bert(txt_ids, txt_masks, hist_img_fts, hist_ang_fts,
                                                       hist_pano_img_fts, hist_pano_ang_fts, hist_masks,
                                                       ob_img_fts, ob_ang_fts, ob_nav_types, ob_masks, lang_rows=cls_rows)
        fused = ops.mul_bcast(ob_embeds, txt_embeds if txt_embeds.dim() == 2 else txt_embeds[:, 0])     # (2-D: the [CLS] rows already)
        prediction_scores = self.next_action(fused).squeeze(-1)
        prediction_scores = ops.fill_where_zero(prediction_scores, ob_nav_types, -float('inf'))
        if compute_loss:
            return ops.cross_entropy(prediction_scores, act_labels)
        return prediction_scores

    # ---- A17
    def forward_sar(self, txt_ids, txt_masks, hist_img_fts, hist_ang_fts, hist_pano_img_fts, hist_pano_ang_fts,
                    hist_masks, ob_img_fts, ob_ang_fts, ob_nav_types, ob_masks, ob_act_angles, ob_progress, compute_loss):
        B, L = txt_ids.shape
        cls_rows = ops.const_index("arange_mul", int(B), int(L), device=txt_ids.device)
        txt_embeds, hist_embeds, ob_embeds = self.bert(txt_ids, txt_masks, hist_img_fts, hist_ang_fts,
                                                       hist_pano_img_fts, hist_pano_ang_fts, hist_masks,
                                                       ob_img_fts, ob_ang_fts, ob_nav_types, ob_masks, need="lang", lang_rows=cls_rows)
        cls = txt_embeds if txt_embeds.dim() == 2 else ops.gather_rows(txt_embeds.reshape(B * L, -1), cls_rows, unique=True)
        prediction_scores = self.regress_action(cls)
        if compute_loss:
            act_targets = torch.cat([ob_act_angles, ob_progress.unsqueeze(1)], dim=1)
            return ops.mse_loss(prediction_scores, act_targets)
        return prediction_scores

    # ---- A18
    def forward_sprel(self, txt_ids, txt_masks, hist_img_fts, hist_ang_fts, hist_pano_img_fts, hist_pano_ang_fts,
                      hist_masks, ob_img_fts, ob_ang_fts, ob_nav_types, ob_masks, sp_anchor_idxs, sp_targets, compute_loss):
        txt_embeds, hist_embeds, ob_embeds = self.bert(txt_ids, txt_masks, hist_img_fts, hist_ang_fts,
                                                       hist_pano_img_fts, hist_pano_ang_fts, hist_masks,
                                                       ob_img_fts, ob_ang_fts, ob_nav_types, ob_masks, need="visn")
        B, S, H = ob_embeds.shape                                  # S = 37; the reference hard-codes 36 views (:211-212)
        flat = ob_embeds.reshape(B * S, H)
        base = ops.const_index("arange_mul", int(B), int(S), device=flat.device)
        # the anchor view's embedding next to each of the 36 views (:211-214).  One row per sample is gathered and BROADCAST: the backward
        # is a sum over the 36 copies and one add per anchor row -- a gather of 36 repeated indices would scatter-add 36 atomics into
        # the same row in whatever order they land, and the bf16 images downstream re-round that noise into every weight gradient
        # (run-to-run differences of up to 5e-3 of a gradient's largest element at small batches: tools/determinism_check.py)
        anchor = ops.gather_rows(flat, base + sp_anchor_idxs, unique=True)
        rest = ops.gather_rows(flat, ops.const_index("rows_but_last", int(B), int(S), device=flat.device), unique=True)
        cat_ob_embeds = torch.cat([anchor[:, None].expand(B, S - 1, H), rest.view(B, S - 1, H)], -1)
        prediction_scores = self.sprel_head(cat_ob_embeds)
        if compute_loss:
            return ops.mse_loss(prediction_scores, sp_targets)
        return prediction_scores

    # ---- A19
    def forward_mrc(self, txt_ids, txt_masks, hist_img_fts, hist_ang_fts, hist_pano_img_fts, hist_pano_ang_fts,
                    hist_masks, hist_mrc_masks, hist_img_probs, compute_loss=True, mrc_idx=None):
        """`mrc_idx` (optional): int64 flat positions b*T+t of the masked steps in row-major order (see forward_mlm)."""
        txt_embeds, hist_embeds, _ = self.bert(txt_ids, txt_masks, hist_img_fts, hist_ang_fts, hist_pano_img_fts,
                                               hist_pano_ang_fts, hist_masks, None, None, None, None, need="visn")
        B, T1, H = hist_embeds.shape
        T = T1 - 1
        if mrc_idx is None:
            mrc_idx = hist_mrc_masks.reshape(-1).nonzero(as_tuple=False).squeeze(1)
        rows = torch.div(mrc_idx, T, rounding_mode='floor') * T1 + mrc_idx % T + 1   # +1: drop the global cls slot (:232)
        masked_output = ops.gather_rows(hist_embeds.reshape(B * T1, H), rows, unique=True)
        prediction_soft_labels = self.image_classifier(masked_output)
        hist_mrc_targets = hist_img_probs.reshape(B * T, -1).index_select(0, mrc_idx)
        if compute_loss:
            return ops.kl_div_logsoftmax(prediction_soft_labels, hist_mrc_targets)
        return prediction_soft_labels, hist_mrc_targets

    # ---- A20
    def forward_itm(self, txt_ids, txt_masks, hist_img_fts, hist_ang_fts, hist_pano_img_fts, hist_pano_ang_fts,
                    hist_masks, num_neg_trajs, compute_loss, neg_idxs=None, shuffled_pos_ids=None):
        fused_embeds = self.bert.forward_itm(txt_ids, txt_masks, hist_img_fts, hist_ang_fts, hist_pano_img_fts,
                                             hist_pano_ang_fts, hist_masks, num_neg_trajs=num_neg_trajs,
                                             neg_idxs=neg_idxs, shuffled_pos_ids=shuffled_pos_ids)
        prediction_scores = self.itm_head(fused_embeds).squeeze(2)
        itm_targets = ops.const_index("zeros", int(fused_embeds.size(0)), device=fused_embeds.device)
        if compute_loss:
            return ops.cross_entropy(prediction_scores, itm_targets)
        return prediction_scores, itm_targets
