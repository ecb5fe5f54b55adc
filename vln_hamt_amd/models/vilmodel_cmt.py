"""MI355X-native mirror of ``finetune_src/models/vilmodel_cmt.py`` (the finetune twin of the trunk).

Shares the HIP building blocks of ``vln_hamt_amd.model.vilmodel`` and restates only what differs in the finetune
file: the `no_lang_ca` switch of the cross-modal layer (vilmodel_cmt.py:365, 382-411), frozen language layers when
`update_lang_bert` is False (:440-442), the *per-step* `HistoryEmbeddings.forward` with dropout in front of the
panorama encoder (:553-594) and the three-mode `NavCMT.forward` with its built-in `next_action` head (:610-728).
The fused attention kernel never materialises attention scores; where the reference returns them
(`output_attentions`, :127-128, :348) a `None` placeholder is returned -- no caller on the path reads them.
"""
from __future__ import annotations

import copy

import torch
from torch import nn

from .. import blocks, ops, streams
from ..model import vilmodel as V
from ..model.pretrain_cmt import _MlpHead
from ..modeling import HamtPreTrainedModel, precision_of

BertLayerNorm = V.BertLayerNorm
BertPreTrainedModel = HamtPreTrainedModel
gelu = V.gelu
BertEmbeddings, BertSelfAttention, BertSelfOutput, BertAttention = V.BertEmbeddings, V.BertSelfAttention, V.BertSelfOutput, V.BertAttention
BertIntermediate, BertOutput, BertLayer, BertEncoder = V.BertIntermediate, V.BertOutput, V.BertLayer, V.BertEncoder
BertOutAttention, ImageEmbeddings = V.BertOutAttention, V.ImageEmbeddings


class BertXAttention(nn.Module):
    """vilmodel_cmt.py:349-358: returns (attention_output, attention_scores)."""

    def __init__(self, config, ctx_dim=None):
        super().__init__()
        self.att = BertOutAttention(config, ctx_dim=ctx_dim)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, ctx_tensor, ctx_att_mask=None):
        if input_tensor.dim() == 3 and blocks.usable(self.att.prec, input_tensor):
            return blocks.cross_attn_block(input_tensor, ctx_tensor, ctx_att_mask, self.att, self.output, self.training), None
        return self.output(self.att(input_tensor, ctx_tensor, ctx_att_mask), input_tensor), None


class LXRTXLayer(nn.Module):
    """vilmodel_cmt.py:360-418.  With `no_lang_ca` the language stream is passed through untouched and only serves
    as keys/values for the vision stream."""

    def __init__(self, config):
        super().__init__()
        self.no_lang_ca = config.no_lang_ca
        self.lang_self_att = BertAttention(config)
        self.lang_inter = BertIntermediate(config)
        self.lang_output = BertOutput(config)
        self.visn_self_att = BertAttention(config)
        self.visn_inter = BertIntermediate(config)
        self.visn_output = BertOutput(config)
        self.visual_attention = BertXAttention(config)

    def cross_att(self, lang_input, lang_attention_mask, visn_input, visn_attention_mask):
        if self.no_lang_ca:
            lang_att = lang_input
        else:
            lang_att, _ = self.visual_attention(lang_input, visn_input, ctx_att_mask=visn_attention_mask)
        visn_att, _ = self.visual_attention(visn_input, lang_input, ctx_att_mask=lang_attention_mask)
        return lang_att, visn_att

    def self_att(self, lang_input, lang_attention_mask, visn_input, visn_attention_mask):
        lang_att = (lang_input,) if self.no_lang_ca else self.lang_self_att(lang_input, lang_attention_mask)
        return lang_att, self.visn_self_att(visn_input, visn_attention_mask)

    def output_fc(self, lang_input, visn_input):
        lang_out = lang_input if self.no_lang_ca else V._ffn(self.lang_inter, self.lang_output, lang_input, self.training)
        return lang_out, V._ffn(self.visn_inter, self.visn_output, visn_input, self.training)

    def forward(self, lang_feats, lang_attention_mask, visn_feats, visn_attention_mask):
        lang, visn = self.cross_att(lang_feats, lang_attention_mask, visn_feats, visn_attention_mask)
        lang, visn = self.self_att(lang, lang_attention_mask, visn, visn_attention_mask)
        return self.output_fc(lang[0], visn[0])


class LxmertEncoder(V.LxmertEncoder):
    """vilmodel_cmt.py:420-486: same forward; language layers are frozen when update_lang_bert is False."""

    def __init__(self, config):
        nn.Module.__init__(self)
        self.num_l_layers = config.num_l_layers
        self.num_r_layers = config.num_r_layers
        self.num_h_layers = config.num_h_layers
        self.num_x_layers = config.num_x_layers
        self.update_lang_bert = config.update_lang_bert
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(self.num_l_layers)])
        if not self.update_lang_bert:
            for _, p in self.layer.named_parameters():
                p.requires_grad = False
        self.h_layers = nn.ModuleList([BertLayer(config) for _ in range(self.num_h_layers)]) if self.num_h_layers > 0 else None
        self.r_layers = nn.ModuleList([BertLayer(config) for _ in range(self.num_r_layers)]) if self.num_r_layers > 0 else None
        self.x_layers = nn.ModuleList([LXRTXLayer(config) for _ in range(self.num_x_layers)])


class HistoryEmbeddings(nn.Module):
    """per-step history embedding (vilmodel_cmt.py:512-594): img (B,D), ang (B,4), pos_ids (1,) or (B,),
    pano (B,36,D) -> (B,H); called without features it returns the cls embedding (1,H)."""

    def __init__(self, config):
        super().__init__()
        self.cls_token = nn.Parameter(torch.zeros(1, 1, config.hidden_size))
        self.img_linear = nn.Linear(config.image_feat_size, config.hidden_size)
        self.img_layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.ang_linear = nn.Linear(config.angle_feat_size, config.hidden_size)
        self.ang_layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.position_embeddings = nn.Embedding(config.max_action_steps, config.hidden_size)
        self.type_embedding = nn.Embedding(1, config.hidden_size)
        self.layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.hist_enc_pano = config.hist_enc_pano
        if config.hist_enc_pano:
            self.pano_img_linear = nn.Linear(config.image_feat_size, config.hidden_size)
            self.pano_img_layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
            self.pano_ang_linear = nn.Linear(config.angle_feat_size, config.hidden_size)
            self.pano_ang_layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
            pano_cfg = copy.copy(config)
            pano_cfg.num_hidden_layers = config.num_h_pano_layers
            self.pano_encoder = BertEncoder(pano_cfg)
        else:
            self.pano_encoder = None
        self.prec = precision_of(config)

    def forward(self, img_feats, ang_feats, pos_ids, pano_img_feats=None, pano_ang_feats=None):
        dev = self.cls_token.device
        H = self.cls_token.shape[-1]
        p = float(self.dropout.p) if self.training else 0.0
        if img_feats is None:
            z = torch.zeros(1, dtype=torch.long, device=dev)
            cls = ops.gather_rows(self.type_embedding.weight, z, base=ops.gather_rows(self.cls_token.view(1, H), z))
            return ops.layer_norm(cls, None, self.layer_norm, p_post=p)
        B = img_feats.size(0)
        e = V._VisualLinears.two_stream(self.img_linear, self.img_layer_norm, self.ang_linear, self.ang_layer_norm,
                                        img_feats, ang_feats, self.prec)
        e = ops.gather_rows(self.position_embeddings.weight, pos_ids.reshape(-1).expand(B), base=e)
        e = ops.gather_rows(self.type_embedding.weight, torch.zeros(B, dtype=torch.long, device=dev), base=e)
        if self.pano_encoder is not None:
            Vn = pano_img_feats.shape[1]
            pe = V._VisualLinears.two_stream(self.pano_img_linear, self.pano_img_layer_norm, self.pano_ang_linear,
                                             self.pano_ang_layer_norm, pano_img_feats, pano_ang_feats, self.prec)
            pe = ops.dropout(pe, float(self.dropout.p), self.training)          # :583
            pe = self.pano_encoder(pe.view(B, Vn, H), None)[0]
            e = ops.add3(e, ops.mean_mid(pe))
        return ops.layer_norm(e, None, self.layer_norm, p_post=p)


class NextActionPrediction(_MlpHead):          # vilmodel_cmt.py:596-607
    def __init__(self, hidden_size, dropout_rate, prec="bf16"):
        super().__init__(hidden_size, hidden_size, 1, dropout_rate, prec)


class NavCMT(BertPreTrainedModel):
    """vilmodel_cmt.py:610-728: `language` (once per episode), `history` (one step), `visual` (one decision)."""
    _hamt_container = True      # (optim.AdamW.attach) direct parameter reads below are behind streams.gate

    def __init__(self, config):
        super().__init__(config)
        self.embeddings = BertEmbeddings(config)
        self.img_embeddings = ImageEmbeddings(config)
        self.hist_embeddings = HistoryEmbeddings(config)
        self.encoder = LxmertEncoder(config)
        self.next_action = NextActionPrediction(config.hidden_size, config.pred_head_dropout_prob, precision_of(config))
        self.init_weights()

    @staticmethod
    def _extend(mask):
        return ops.extend_mask(mask)

    def forward(self, mode, txt_ids=None, txt_embeds=None, txt_masks=None, hist_img_feats=None, hist_ang_feats=None,
                hist_pano_img_feats=None, hist_pano_ang_feats=None, hist_embeds=None, ob_step_ids=None, hist_masks=None,
                ob_img_feats=None, ob_ang_feats=None, ob_nav_types=None, ob_masks=None):
        cfg = self.config
        if mode == 'language':
            txt_m = self._extend(txt_masks)
            txt = self.embeddings(txt_ids)
            for layer in self.encoder.layer:
                txt = layer(txt, txt_m)[0]
            if cfg.fix_lang_embedding:
                txt = txt.detach()
            if cfg.no_lang_ca:      # the text stream is step-invariant: precompute its per-x-layer self-att/FFN outputs
                outs = [txt]
                for layer in self.encoder.x_layers:
                    streams.gate(layer)            # (lang_inter / lang_output / the cross-attention K, V weights are read right here)
                    att = layer.lang_self_att(txt, txt_m)[0]
                    outs.append(V._ffn(layer.lang_inter, layer.lang_output, att, self.training))
                # ... and, beyond the reference, what every later `visual` call would re-derive from them: the key / value
                # projections of x-layer l's cross attention over outs[l].  They ride on the tensors
                # (`_hamt_xkv`); a `visual` call given other tensors simply projects again.
                if blocks.usable(precision_of(cfg), txt):
                    for l, layer in enumerate(self.encoder.x_layers):
                        blocks.precompute_cross_kv(outs[l], layer.visual_attention.att)
                return outs
            return txt

        if mode == 'history':
            h = self.hist_embeddings(hist_img_feats, hist_ang_feats, ob_step_ids,
                                     pano_img_feats=hist_pano_img_feats, pano_ang_feats=hist_pano_ang_feats)
            return h.detach() if cfg.fix_hist_embedding else h

        if mode == 'visual':
            hist_m = self._extend(hist_masks)
            if self.encoder.h_layers is not None:
                for layer in self.encoder.h_layers:
                    hist_embeds = layer(hist_embeds, hist_m)[0]
            ob_m = self._extend(ob_masks)
            B = ob_img_feats.size(0)
            ones = torch.ones(B, dtype=torch.long, device=ob_img_feats.device)
            streams.gate(self.embeddings.token_type_embeddings.weight)
            tt = ops.gather_rows(self.embeddings.token_type_embeddings.weight, ones).view(B, 1, -1)
            ob = self.img_embeddings(ob_img_feats, ob_ang_feats, tt, nav_types=ob_nav_types)
            if self.encoder.r_layers is not None:
                for layer in self.encoder.r_layers:
                    ob = layer(ob, ob_m)[0]
            if cfg.fix_obs_embedding:
                ob = ob.detach()
            n_hist = hist_embeds.size(1)
            vis = torch.cat([hist_embeds, ob], 1)
            vis_m = torch.cat([hist_m, ob_m], -1)
            txt_m = self._extend(txt_masks)
            all_txt = txt_embeds
            for l, layer in enumerate(self.encoder.x_layers):
                if cfg.no_lang_ca:
                    txt_embeds = all_txt[l]
                txt_embeds, vis = layer(txt_embeds, txt_m, vis, vis_m)
            hist_out, ob_out = vis[:, :n_hist], vis[:, n_hist:].contiguous()
            if cfg.no_lang_ca or cfg.act_pred_token == 'ob':
                fuse = ob_out
            elif cfg.act_pred_token == 'ob_txt':
                fuse = ops.mul_bcast(ob_out, txt_embeds[:, 0])
            elif cfg.act_pred_token == 'ob_hist':
                fuse = ops.mul_bcast(ob_out, hist_out[:, 0])
            elif cfg.act_pred_token == 'ob_txt_hist':
                fuse = ops.mul_bcast(ob_out, ops.add3(txt_embeds[:, 0].contiguous(), hist_out[:, 0].contiguous()))
            else:
                raise ValueError(cfg.act_pred_token)
            act_logits = ops.fill_where_zero(self.next_action(fuse).squeeze(-1), ob_nav_types, -float('inf'))
            return act_logits, txt_embeds, hist_out, ob_out
        raise ValueError(mode)


def copy_language_(dst, src):
    """In-place refresh of a kept `language` result (tensor or the `no_lang_ca` list) with a new episode's, INCLUDING the cached
    cross-attention key / value projections riding on the tensors: what a hipGraph-captured `visual` step keeps reading
    (graph.GraphedInference bakes the addresses of `dst` in)."""
    if torch.is_tensor(dst):
        dst, src = [dst], [src]
    for d, s_ in zip(dst, src):
        kd, ks = getattr(d, "_hamt_xkv", None), getattr(s_, "_hamt_xkv", None)
        d.copy_(s_)
        if kd is not None and ks is not None:
            kd[1].copy_(ks[1])
            d._hamt_xkv = (ks[0][:1] + (d._version,) + ks[0][2:], kd[1])      # same weights, `d`'s new version
        elif kd is not None:
            del d._hamt_xkv
    return dst
