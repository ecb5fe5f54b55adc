"""CPU ORACLE for the HAMT hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a from-scratch, *functional* restatement (plain PyTorch fp32 on the
CPU, operating on a flat ``state_dict``) of the arithmetic of the reference's
``pretrain_src/model/{vilmodel,pretrain_cmt}.py`` and of the finetune twin
``finetune_src/models/vilmodel_cmt.py``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package (``vln_hamt_amd/``) never does.

Parity status: **pinned**.  ``oracle/gen_goldens.py`` imports the real reference
from ``/root/reference`` (through ``oracle/ref_shim.py``), runs it on seeded
inputs and commits inputs/outputs under ``tests/golden/``;
``tests/test_oracle_goldens.py`` checks this file against those vectors.

Every function cites the reference lines it follows (paths relative to
``/root/reference``).  Parameter names are the reference's ``state_dict`` keys.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- config
@dataclass
class OracleConfig:
    """Keys of pretrain_src/config/r2r_model_config.json:2-32 (+ pretrain_tasks)."""
    hidden_size: int = 768
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    vocab_size: int = 30522
    type_vocab_size: int = 2
    max_position_embeddings: int = 512
    max_action_steps: int = 100
    image_feat_size: int = 768
    angle_feat_size: int = 4
    image_prob_size: int = 1000
    num_l_layers: int = 9
    num_x_layers: int = 4
    num_h_pano_layers: int = 2
    num_h_layers: int = 0
    num_r_layers: int = 0
    layer_norm_eps: float = 1e-12
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    pred_head_dropout_prob: float = 0.1
    update_lang_bert: bool = True
    pretrain_tasks: Sequence[str] = ("mlm", "sap", "sar", "sprel", "mrc", "itm")
    # finetune-only switches (finetune_src/models/vlnbert_init.py:42-63)
    hist_enc_pano: bool = True
    no_lang_ca: bool = False
    act_pred_token: str = "ob_txt"
    fix_lang_embedding: bool = False
    fix_hist_embedding: bool = False
    fix_obs_embedding: bool = False

    @staticmethod
    def tiny(**kw) -> "OracleConfig":
        base = dict(hidden_size=96, num_attention_heads=12, intermediate_size=384, vocab_size=1000,
                    max_position_embeddings=64, max_action_steps=20, image_feat_size=32,
                    image_prob_size=40, num_l_layers=2, num_x_layers=2, num_h_pano_layers=1)
        base.update(kw)
        return OracleConfig(**base)


# --------------------------------------------------------------------------- parameter inventory
def _bert_layer_shapes(p: str, H: int, I: int) -> Dict[str, tuple]:
    s = {}
    for n in ("query", "key", "value"):
        s[f"{p}.attention.self.{n}.weight"] = (H, H)
        s[f"{p}.attention.self.{n}.bias"] = (H,)
    s[f"{p}.attention.output.dense.weight"] = (H, H)
    s[f"{p}.attention.output.dense.bias"] = (H,)
    s[f"{p}.attention.output.LayerNorm.weight"] = (H,)
    s[f"{p}.attention.output.LayerNorm.bias"] = (H,)
    s[f"{p}.intermediate.dense.weight"] = (I, H)
    s[f"{p}.intermediate.dense.bias"] = (I,)
    s[f"{p}.output.dense.weight"] = (H, I)
    s[f"{p}.output.dense.bias"] = (H,)
    s[f"{p}.output.LayerNorm.weight"] = (H,)
    s[f"{p}.output.LayerNorm.bias"] = (H,)
    return s


def _mlp_head_shapes(p: str, din: int, H: int, dout: int, with_dropout: bool) -> Dict[str, tuple]:
    last = 4 if with_dropout else 3  # index of the final Linear inside nn.Sequential
    return {f"{p}.net.0.weight": (H, din), f"{p}.net.0.bias": (H,),
            f"{p}.net.2.weight": (H,), f"{p}.net.2.bias": (H,),
            f"{p}.net.{last}.weight": (dout, H), f"{p}.net.{last}.bias": (dout,)}


def trunk_param_shapes(cfg: OracleConfig, prefix: str = "bert.") -> Dict[str, tuple]:
    """Names/shapes of NavPreTrainedModel (vilmodel.py:578-589) in registration order."""
    H, I = cfg.hidden_size, cfg.intermediate_size
    s: Dict[str, tuple] = {}
    e = prefix + "embeddings"
    s[f"{e}.word_embeddings.weight"] = (cfg.vocab_size, H)
    s[f"{e}.position_embeddings.weight"] = (cfg.max_position_embeddings, H)
    s[f"{e}.token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
    s[f"{e}.LayerNorm.weight"] = (H,)
    s[f"{e}.LayerNorm.bias"] = (H,)
    e = prefix + "img_embeddings"
    s[f"{e}.img_linear.weight"] = (H, cfg.image_feat_size)
    s[f"{e}.img_linear.bias"] = (H,)
    s[f"{e}.img_layer_norm.weight"] = (H,)
    s[f"{e}.img_layer_norm.bias"] = (H,)
    s[f"{e}.ang_linear.weight"] = (H, cfg.angle_feat_size)
    s[f"{e}.ang_linear.bias"] = (H,)
    s[f"{e}.ang_layer_norm.weight"] = (H,)
    s[f"{e}.ang_layer_norm.bias"] = (H,)
    s[f"{e}.nav_type_embedding.weight"] = (3, H)
    s[f"{e}.layer_norm.weight"] = (H,)
    s[f"{e}.layer_norm.bias"] = (H,)
    e = prefix + "hist_embeddings"
    s[f"{e}.cls_token"] = (1, 1, H)
    for lin, k in (("img", cfg.image_feat_size), ("ang", cfg.angle_feat_size)):
        s[f"{e}.{lin}_linear.weight"] = (H, k)
        s[f"{e}.{lin}_linear.bias"] = (H,)
        s[f"{e}.{lin}_layer_norm.weight"] = (H,)
        s[f"{e}.{lin}_layer_norm.bias"] = (H,)
    if cfg.num_h_pano_layers > 0:
        for lin, k in (("pano_img", cfg.image_feat_size), ("pano_ang", cfg.angle_feat_size)):
            s[f"{e}.{lin}_linear.weight"] = (H, k)
            s[f"{e}.{lin}_linear.bias"] = (H,)
            s[f"{e}.{lin}_layer_norm.weight"] = (H,)
            s[f"{e}.{lin}_layer_norm.bias"] = (H,)
        for i in range(cfg.num_h_pano_layers):
            s.update(_bert_layer_shapes(f"{e}.pano_encoder.layer.{i}", H, I))
    s[f"{e}.position_embeddings.weight"] = (cfg.max_action_steps, H)
    s[f"{e}.type_embedding.weight"] = (1, H)
    s[f"{e}.layer_norm.weight"] = (H,)
    s[f"{e}.layer_norm.bias"] = (H,)
    enc = prefix + "encoder"
    for i in range(cfg.num_l_layers):
        s.update(_bert_layer_shapes(f"{enc}.layer.{i}", H, I))
    for i in range(cfg.num_x_layers):
        x = f"{enc}.x_layers.{i}"
        for side in ("lang", "visn"):
            full = _bert_layer_shapes("L", H, I)
            for k, v in full.items():
                k = k[2:]
                if k.startswith("attention."):
                    s[f"{x}.{side}_self_att.{k[len('attention.'):]}"] = v
            for k, v in full.items():
                k = k[2:]
                if k.startswith("intermediate."):
                    s[f"{x}.{side}_inter.{k[len('intermediate.'):]}"] = v
            for k, v in full.items():
                k = k[2:]
                if k.startswith("output."):
                    s[f"{x}.{side}_output.{k[len('output.'):]}"] = v
        for n in ("query", "key", "value"):
            s[f"{x}.visual_attention.att.{n}.weight"] = (H, H)
            s[f"{x}.visual_attention.att.{n}.bias"] = (H,)
        s[f"{x}.visual_attention.output.dense.weight"] = (H, H)
        s[f"{x}.visual_attention.output.dense.bias"] = (H,)
        s[f"{x}.visual_attention.output.LayerNorm.weight"] = (H,)
        s[f"{x}.visual_attention.output.LayerNorm.bias"] = (H,)
    return s


def pretrain_param_shapes(cfg: OracleConfig) -> Dict[str, tuple]:
    """state_dict of MultiStepNavCMTPreTraining (pretrain_cmt.py:73-94), incl. the tied decoder key."""
    H = cfg.hidden_size
    s = trunk_param_shapes(cfg, "bert.")
    t = set(cfg.pretrain_tasks)
    if "mlm" in t:
        s["mlm_head.predictions.bias"] = (cfg.vocab_size,)
        s["mlm_head.predictions.transform.dense.weight"] = (H, H)
        s["mlm_head.predictions.transform.dense.bias"] = (H,)
        s["mlm_head.predictions.transform.LayerNorm.weight"] = (H,)
        s["mlm_head.predictions.transform.LayerNorm.bias"] = (H,)
        s["mlm_head.predictions.decoder.weight"] = (cfg.vocab_size, H)  # tied, pretrain_cmt.py:96-99
    if "sap" in t:
        s.update(_mlp_head_shapes("next_action", H, H, 1, True))
    if "sar" in t:
        s.update(_mlp_head_shapes("regress_action", H, H, 3, True))
    if "sprel" in t:
        s.update(_mlp_head_shapes("sprel_head", 2 * H, H, 2, True))
    if "mrc" in t:
        s.update(_mlp_head_shapes("image_classifier", H, H, cfg.image_prob_size, False))
    if "itm" in t:
        s.update(_mlp_head_shapes("itm_head", H, H, 1, False))
    return s


TIED_KEYS = ("mlm_head.predictions.decoder.weight", "bert.embeddings.word_embeddings.weight")


def make_state_dict(shapes: Dict[str, tuple], seed: int = 0, scale: float = 0.02,
                    dtype=np.float32) -> Dict[str, Tensor]:
    """Platform-stable numpy-PCG64 weight recipe (SURVEY.md 8c golden set 2).

    Every tensor is drawn in key order; LayerNorm weights are 1+N(0,0.1), biases N(0,0.02) so that no
    parameter is at a value (0 / 1) that could hide an indexing bug.  The tied MLM decoder reuses the
    word-embedding tensor (pretrain_cmt.py:96-99).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, Tensor] = {}
    for k, shp in shapes.items():
        if k == TIED_KEYS[0]:
            continue
        a = rng.standard_normal(size=shp, dtype=np.float32)
        is_ln_w = k.endswith("weight") and ("LayerNorm" in k or "layer_norm" in k or k.endswith("net.2.weight"))
        if is_ln_w:
            a = 1.0 + 0.1 * a
        else:
            a = scale * a
        sd[k] = torch.from_numpy(a.astype(dtype))
    if TIED_KEYS[0] in shapes:
        sd[TIED_KEYS[0]] = sd[TIED_KEYS[1]]
    return sd


# --------------------------------------------------------------------------- primitives
def gelu_erf(x: Tensor) -> Tensor:
    """vilmodel.py:23-29 -- exact erf GELU."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def extend_mask(mask: Tensor, dtype=torch.float32) -> Tensor:
    """(B,S) bool -> additive (B,1,1,S): (1-m)*-10000, vilmodel.py:597-599."""
    return (1.0 - mask[:, None, None, :].to(dtype)) * -10000.0


class HamtOracle:
    """Functional oracle over a flat state_dict ``sd`` (reference key names)."""

    def __init__(self, sd: Dict[str, Tensor], cfg: OracleConfig, training: bool = False):
        self.sd, self.cfg, self.training = sd, cfg, training

    # ---- helpers
    def _lin(self, name: str, x: Tensor) -> Tensor:
        return F.linear(x, self.sd[name + ".weight"], self.sd.get(name + ".bias"))

    def _ln(self, name: str, x: Tensor, eps: float) -> Tensor:
        return F.layer_norm(x, (x.shape[-1],), self.sd[name + ".weight"], self.sd[name + ".bias"], eps)

    def _drop(self, x: Tensor, p: float) -> Tensor:
        return F.dropout(x, p, self.training)

    # ---- A1 BertEmbeddings.forward, vilmodel.py:54-69
    def text_embeddings(self, txt_ids: Tensor, prefix="bert.embeddings") -> Tensor:
        L = txt_ids.shape[1]
        w = self.sd[f"{prefix}.word_embeddings.weight"][txt_ids]
        p = self.sd[f"{prefix}.position_embeddings.weight"][:L][None]
        t = self.sd[f"{prefix}.token_type_embeddings.weight"][0][None, None]
        x = self._ln(f"{prefix}.LayerNorm", w + p + t, self.cfg.layer_norm_eps)
        return self._drop(x, self.cfg.hidden_dropout_prob)

    # ---- A2/A7 attention core, vilmodel.py:96-129 and :322-349
    def attention(self, p: str, x: Tensor, ctx: Tensor, add_mask: Optional[Tensor]) -> Tensor:
        nh = self.cfg.num_attention_heads
        B, Sq, H = x.shape
        Sk = ctx.shape[1]
        d = H // nh
        q = self._lin(f"{p}.query", x).view(B, Sq, nh, d).transpose(1, 2)
        k = self._lin(f"{p}.key", ctx).view(B, Sk, nh, d).transpose(1, 2)
        v = self._lin(f"{p}.value", ctx).view(B, Sk, nh, d).transpose(1, 2)
        s = q @ k.transpose(-1, -2) / math.sqrt(d)
        if add_mask is not None:
            s = s + add_mask
        pr = self._drop(torch.softmax(s, -1), self.cfg.attention_probs_dropout_prob)
        return (pr @ v).transpose(1, 2).reshape(B, Sq, H)

    # ---- A3/A5 dense -> dropout -> LN(x + residual), vilmodel.py:139-143, :181-185
    def out_res_ln(self, p: str, x: Tensor, residual: Tensor) -> Tensor:
        y = self._drop(self._lin(f"{p}.dense", x), self.cfg.hidden_dropout_prob)
        return self._ln(f"{p}.LayerNorm", y + residual, self.cfg.layer_norm_eps)

    def self_att_block(self, p: str, x: Tensor, add_mask: Tensor) -> Tensor:
        """BertAttention, vilmodel.py:152-156."""
        return self.out_res_ln(f"{p}.output", self.attention(f"{p}.self", x, x, add_mask), x)

    def ffn(self, p_inter: str, p_out: str, x: Tensor) -> Tensor:
        """BertIntermediate + BertOutput, vilmodel.py:168-171, :181-185."""
        return self.out_res_ln(p_out, gelu_erf(self._lin(f"{p_inter}.dense", x)), x)

    # ---- A6 BertLayer, vilmodel.py:195-201
    def bert_layer(self, p: str, x: Tensor, add_mask: Tensor) -> Tensor:
        a = self.self_att_block(f"{p}.attention", x, add_mask)
        return self.ffn(f"{p}.intermediate", f"{p}.output", a)

    # ---- A8 LXRTXLayer, vilmodel.py:401-412 (shared cross-attention weights :379-383)
    def x_layer(self, p: str, lang: Tensor, lang_mask: Tensor, visn: Tensor, visn_mask: Tensor,
                no_lang_ca: bool = False):
        """no_lang_ca (finetune only, vilmodel_cmt.py:365, :382-411): the language stream is passed
        through untouched (no cross-att, self-att or FFN) and only serves as context for the vision side."""
        va = f"{p}.visual_attention"
        visn_x = self.out_res_ln(f"{va}.output", self.attention(f"{va}.att", visn, lang, lang_mask), visn)
        visn_s = self.self_att_block(f"{p}.visn_self_att", visn_x, visn_mask)
        visn_o = self.ffn(f"{p}.visn_inter", f"{p}.visn_output", visn_s)
        if no_lang_ca:
            return lang, visn_o
        lang_x = self.out_res_ln(f"{va}.output", self.attention(f"{va}.att", lang, visn, visn_mask), lang)
        lang_s = self.self_att_block(f"{p}.lang_self_att", lang_x, lang_mask)
        return self.ffn(f"{p}.lang_inter", f"{p}.lang_output", lang_s), visn_o

    # ---- A10 ImageEmbeddings, vilmodel.py:496-505
    def image_embeddings(self, img: Tensor, ang: Tensor, type_emb: Tensor, nav_types: Optional[Tensor],
                         p="bert.img_embeddings") -> Tensor:
        x = self._ln(f"{p}.img_layer_norm", self._lin(f"{p}.img_linear", img), 1e-12) \
            + self._ln(f"{p}.ang_layer_norm", self._lin(f"{p}.ang_linear", ang), 1e-12) + type_emb
        if nav_types is not None:
            x = x + self.sd[f"{p}.nav_type_embedding.weight"][nav_types]
        return self._drop(self._ln(f"{p}.layer_norm", x, 1e-12), self.cfg.hidden_dropout_prob)

    # ---- A11 HistoryEmbeddings, vilmodel.py:540-575
    def history_embeddings(self, img, ang, pano_img, pano_ang, pos_ids, batch_size,
                           p="bert.hist_embeddings"):
        type_emb = self.sd[f"{p}.type_embedding.weight"][0][None, None]          # (1,1,H)
        cls = self.sd[f"{p}.cls_token"].expand(batch_size, -1, -1) + type_emb
        cls = self._drop(self._ln(f"{p}.layer_norm", cls, 1e-12), self.cfg.hidden_dropout_prob)
        if img is None:
            return cls, None
        x = self._ln(f"{p}.img_layer_norm", self._lin(f"{p}.img_linear", img), 1e-12) \
            + self._ln(f"{p}.ang_layer_norm", self._lin(f"{p}.ang_linear", ang), 1e-12) + type_emb
        if self.cfg.num_h_pano_layers > 0:
            B, T, V, _ = pano_img.shape
            pe = self._ln(f"{p}.pano_img_layer_norm", self._lin(f"{p}.pano_img_linear", pano_img.reshape(B * T, V, -1)), 1e-12) \
                + self._ln(f"{p}.pano_ang_layer_norm", self._lin(f"{p}.pano_ang_linear", pano_ang.reshape(B * T, V, -1)), 1e-12)
            zero_mask = torch.zeros(B * T, 1, 1, V, dtype=pe.dtype)               # :560 "assume pano all exists"
            for i in range(self.cfg.num_h_pano_layers):
                pe = self.bert_layer(f"{p}.pano_encoder.layer.{i}", pe, zero_mask)
            x = x + pe.view(B, T, V, -1).mean(2)
        if pos_ids is not None:
            x = x + self.sd[f"{p}.position_embeddings.weight"][pos_ids]
            x = self._drop(self._ln(f"{p}.layer_norm", x, 1e-12), self.cfg.hidden_dropout_prob)
        return cls, x

    # ---- A9 LxmertEncoder.forward, vilmodel.py:438-478
    def encoder(self, txt, txt_m, hist, hist_m, ob=None, ob_m=None, p="bert.encoder"):
        for i in range(self.cfg.num_l_layers):
            txt = self.bert_layer(f"{p}.layer.{i}", txt, txt_m)
        if not self.cfg.update_lang_bert:
            txt = txt.detach()
        n_hist = hist.shape[1]
        if ob is None:
            vis, vis_m = hist, hist_m
        else:
            vis, vis_m = torch.cat([hist, ob], 1), torch.cat([hist_m, ob_m], -1)
        for i in range(self.cfg.num_x_layers):
            txt, vis = self.x_layer(f"{p}.x_layers.{i}", txt, txt_m, vis, vis_m)
        return txt, vis[:, :n_hist], (vis[:, n_hist:] if ob is not None else None)

    # ---- A12 NavPreTrainedModel.forward, vilmodel.py:591-638
    def trunk(self, txt_ids, txt_masks, hist_img, hist_ang, hist_pano_img, hist_pano_ang, hist_masks,
              ob_img, ob_ang, ob_nav_types, ob_masks):
        B = txt_ids.shape[0]
        txt_m = extend_mask(txt_masks)
        txt = self.text_embeddings(txt_ids)
        hist_m = extend_mask(hist_masks)
        pos = torch.arange(hist_img.shape[1])[None] if hist_img is not None else None
        cls, steps = self.history_embeddings(hist_img, hist_ang, hist_pano_img, hist_pano_ang, pos, B)
        hist = cls if steps is None else torch.cat([cls, steps], 1)
        if ob_img is not None:
            tt = self.sd["bert.embeddings.token_type_embeddings.weight"][1][None, None]
            ob = self.image_embeddings(ob_img, ob_ang, tt, ob_nav_types)
            ob_m = extend_mask(ob_masks)
        else:
            ob, ob_m = None, None
        return self.encoder(txt, txt_m, hist, hist_m, ob, ob_m)

    # ---- A13 forward_itm, vilmodel.py:640-724.  RNG draws (np.random.choice :684, torch.randperm :698)
    # are *inputs* here (neg_idxs (B,K) and a list of K' (B,T) position tables), see SURVEY 8c item 3.
    def trunk_itm(self, txt_ids, txt_masks, hist_img, hist_ang, hist_pano_img, hist_pano_ang, hist_masks,
                  neg_idxs: Optional[Tensor], shuffled_pos_ids: List[Tensor], num_neg_trajs: int = 4):
        B = txt_ids.shape[0]
        p = "bert.hist_embeddings"
        txt_m = extend_mask(txt_masks)
        txt = self.text_embeddings(txt_ids)
        for i in range(self.cfg.num_l_layers):
            txt = self.bert_layer(f"bert.encoder.layer.{i}", txt, txt_m)
        txt = txt.repeat(1 + num_neg_trajs, 1, 1)
        txt_m = txt_m.repeat(1 + num_neg_trajs, 1, 1, 1)
        hist_m = extend_mask(hist_masks)
        cls, nopos = self.history_embeddings(hist_img, hist_ang, hist_pano_img, hist_pano_ang, None, B)
        T = hist_img.shape[1]

        def with_pos(pos_ids):
            x = nopos + self.sd[f"{p}.position_embeddings.weight"][pos_ids]
            return self._drop(self._ln(f"{p}.layer_norm", x, 1e-12), self.cfg.hidden_dropout_prob)

        hist = torch.cat([cls, with_pos(torch.arange(T)[None])], 1)
        all_h, all_m = [hist], [hist_m]
        if B > 1:
            for k in range(neg_idxs.shape[1]):
                all_h.append(hist[neg_idxs[:, k]])
                all_m.append(hist_m[neg_idxs[:, k]])
        for pos in shuffled_pos_ids:
            all_h.append(torch.cat([cls, with_pos(pos)], 1))
            all_m.append(hist_m)
        vis, vis_m = torch.cat(all_h, 0), torch.cat(all_m, 0)
        for i in range(self.cfg.num_x_layers):
            txt, vis = self.x_layer(f"bert.encoder.x_layers.{i}", txt, txt_m, vis, vis_m)
        fused = txt[:, 0] * vis[:, 0]
        return torch.stack(torch.split(fused, B), 1)                               # (B, 1+K, H)

    # ---- heads
    def mlp_head(self, p: str, x: Tensor, with_dropout: bool) -> Tensor:
        """Linear -> ReLU -> LN(1e-12) -> [Dropout] -> Linear, pretrain_cmt.py:13-71."""
        h = self._ln(f"{p}.net.2", torch.relu(self._lin(f"{p}.net.0", x)), 1e-12)
        if with_dropout:
            return self._lin(f"{p}.net.4", self._drop(h, self.cfg.pred_head_dropout_prob))
        return self._lin(f"{p}.net.3", h)

    def mlm_head(self, x: Tensor) -> Tensor:
        """BertOnlyMLMHead, vilmodel.py:252-295 (decoder tied to word embeddings)."""
        p = "mlm_head.predictions"
        h = self._ln(f"{p}.transform.LayerNorm", gelu_erf(self._lin(f"{p}.transform.dense", x)), self.cfg.layer_norm_eps)
        return F.linear(h, self.sd["bert.embeddings.word_embeddings.weight"]) + self.sd[f"{p}.bias"]

    # ---- A14-A20 MultiStepNavCMTPreTraining.forward, pretrain_cmt.py:101-262
    def forward(self, batch: dict, task: str, compute_loss: bool = True, itm_rng: Optional[dict] = None):
        g = lambda k: batch.get(k)
        hist_args = (g("txt_ids"), g("txt_masks"), g("hist_img_fts"), g("hist_ang_fts"),
                     g("hist_pano_img_fts"), g("hist_pano_ang_fts"), g("hist_masks"))
        ob_args = (g("ob_img_fts"), g("ob_ang_fts"), g("ob_nav_types"), g("ob_masks"))
        if task.startswith("mlm"):                                                 # :142-159
            txt, _, _ = self.trunk(*hist_args, None, None, None, None)
            labels = batch["txt_labels"]
            sel = labels != -1
            scores = self.mlm_head(txt[sel])                                       # row-major compaction :161-165
            return F.cross_entropy(scores, labels[sel], reduction="none") if compute_loss else scores
        if task.startswith("sap"):                                                 # :167-183
            txt, _, ob = self.trunk(*hist_args, *ob_args)
            scores = self.mlp_head("next_action", ob * txt[:, :1], True).squeeze(-1)
            scores = scores.masked_fill(batch["ob_nav_types"] == 0, -float("inf"))
            return F.cross_entropy(scores, batch["ob_action_viewindex"], reduction="none") if compute_loss else scores
        if task.startswith("sar"):                                                 # :185-200
            txt, _, _ = self.trunk(*hist_args, *ob_args)
            scores = self.mlp_head("regress_action", txt[:, 0], True)
            if not compute_loss:
                return scores
            tgt = torch.cat([batch["ob_action_angles"], batch["ob_progress"][:, None]], 1)
            return F.mse_loss(scores, tgt, reduction="none")
        if task.startswith("sprel"):                                               # :202-222
            _, _, ob = self.trunk(*hist_args, *ob_args)
            idx = batch["sp_anchor_idxs"][:, None, None].repeat(1, 36, ob.shape[-1])
            anchor = torch.gather(ob, 1, idx)
            scores = self.mlp_head("sprel_head", torch.cat([anchor, ob[:, :-1]], -1), True)
            return F.mse_loss(scores, batch["sp_targets"], reduction="none") if compute_loss else scores
        if task.startswith("mrc"):                                                 # :224-243
            _, hist, _ = self.trunk(*hist_args, None, None, None, None)
            sel = batch["hist_mrc_masks"]
            pred = self.mlp_head("image_classifier", hist[:, 1:][sel], False)
            tgt = batch["hist_img_probs"][sel]
            if not compute_loss:
                return pred, tgt
            return F.kl_div(F.log_softmax(pred, -1), tgt, reduction="none").sum(1)
        if task.startswith("itm"):                                                 # :245-262
            fused = self.trunk_itm(*hist_args, itm_rng["neg_idxs"], itm_rng["shuffled_pos_ids"], 4)
            scores = self.mlp_head("itm_head", fused, False).squeeze(2)
            tgt = torch.zeros(fused.shape[0], dtype=torch.long)
            return F.cross_entropy(scores, tgt, reduction="none") if compute_loss else (scores, tgt)
        raise ValueError("invalid task")

    # ======================================================================= finetune twin (NavCMT)
    # finetune_src/models/vilmodel_cmt.py:553-594 -- single-step history embedding (keys have no "bert." prefix)
    def ft_history_step(self, img, ang, pos_ids, pano_img=None, pano_ang=None, p="hist_embeddings"):
        type_emb = self.sd[f"{p}.type_embedding.weight"][0][None]
        if img is None:
            cls = self.sd[f"{p}.cls_token"][:, 0] + type_emb                        # (1,H)
            return self._drop(self._ln(f"{p}.layer_norm", cls, 1e-12), self.cfg.hidden_dropout_prob)
        x = self._ln(f"{p}.img_layer_norm", self._lin(f"{p}.img_linear", img), 1e-12) \
            + self._ln(f"{p}.ang_layer_norm", self._lin(f"{p}.ang_linear", ang), 1e-12) \
            + self.sd[f"{p}.position_embeddings.weight"][pos_ids] + type_emb
        if self.cfg.hist_enc_pano:
            pe = self._ln(f"{p}.pano_img_layer_norm", self._lin(f"{p}.pano_img_linear", pano_img), 1e-12) \
                + self._ln(f"{p}.pano_ang_layer_norm", self._lin(f"{p}.pano_ang_linear", pano_ang), 1e-12)
            pe = self._drop(pe, self.cfg.hidden_dropout_prob)                      # :583
            zm = torch.zeros(pe.shape[0], 1, 1, pe.shape[1], dtype=pe.dtype)
            for i in range(self.cfg.num_h_pano_layers):
                pe = self.bert_layer(f"{p}.pano_encoder.layer.{i}", pe, zm)
            x = x + pe.mean(1)
        return self._drop(self._ln(f"{p}.layer_norm", x, 1e-12), self.cfg.hidden_dropout_prob)

    # vilmodel_cmt.py:624-728
    def ft_forward(self, mode: str, **kw):
        cfg = self.cfg
        if mode == "language":                                                      # :632-653
            txt_m = extend_mask(kw["txt_masks"])
            txt = self.text_embeddings(kw["txt_ids"], prefix="embeddings")
            for i in range(cfg.num_l_layers):
                txt = self.bert_layer(f"encoder.layer.{i}", txt, txt_m)
            if cfg.fix_lang_embedding:
                txt = txt.detach()
            if cfg.no_lang_ca:
                outs = [txt]
                for i in range(cfg.num_x_layers):
                    xp = f"encoder.x_layers.{i}"
                    a = self.self_att_block(f"{xp}.lang_self_att", txt, txt_m)
                    outs.append(self.ffn(f"{xp}.lang_inter", f"{xp}.lang_output", a))
                return outs
            return txt
        if mode == "history":                                                       # :656-661
            h = self.ft_history_step(kw.get("hist_img_feats"), kw.get("hist_ang_feats"), kw.get("ob_step_ids"),
                                     kw.get("hist_pano_img_feats"), kw.get("hist_pano_ang_feats"))
            return h.detach() if cfg.fix_hist_embedding else h
        if mode == "visual":                                                        # :664-728
            hist, hist_m = kw["hist_embeds"], extend_mask(kw["hist_masks"])
            ob_m = extend_mask(kw["ob_masks"])
            tt = self.sd["embeddings.token_type_embeddings.weight"][1][None, None]
            ob = self.image_embeddings(kw["ob_img_feats"], kw["ob_ang_feats"], tt, kw["ob_nav_types"], p="img_embeddings")
            if cfg.fix_obs_embedding:
                ob = ob.detach()
            n_hist = hist.shape[1]
            vis, vis_m = torch.cat([hist, ob], 1), torch.cat([hist_m, ob_m], -1)
            txt_m = extend_mask(kw["txt_masks"])
            txt = kw["txt_embeds"]
            all_txt = txt
            for i in range(cfg.num_x_layers):
                if cfg.no_lang_ca:
                    txt = all_txt[i]
                txt, vis = self.x_layer(f"encoder.x_layers.{i}", txt, txt_m, vis, vis_m, cfg.no_lang_ca)
            hist_o, ob_o = vis[:, :n_hist], vis[:, n_hist:]
            if cfg.no_lang_ca or cfg.act_pred_token == "ob":
                fuse = ob_o
            elif cfg.act_pred_token == "ob_txt":
                fuse = ob_o * txt[:, :1]
            elif cfg.act_pred_token == "ob_hist":
                fuse = ob_o * hist_o[:, :1]
            else:                                                                   # ob_txt_hist
                fuse = ob_o * (txt[:, :1] + hist_o[:, :1])
            logits = self.mlp_head("next_action", fuse, True).squeeze(-1)
            logits = logits.masked_fill(kw["ob_nav_types"] == 0, -float("inf"))
            return logits, txt, hist_o, ob_o
        raise ValueError(mode)


def navcmt_param_shapes(cfg: OracleConfig) -> Dict[str, tuple]:
    """state_dict of finetune NavCMT (vilmodel_cmt.py:610-622): prefix-less trunk + next_action head.
    Registration order differs from the pretrain trunk only inside hist_embeddings (pano after layer_norm)."""
    s = trunk_param_shapes(cfg, "")
    if not cfg.hist_enc_pano:
        s = {k: v for k, v in s.items() if ".pano_" not in k}
    s.update(_mlp_head_shapes("next_action", cfg.hidden_size, cfg.hidden_size, 1, True))
    return s


# --------------------------------------------------------------------------- A24 optimiser-side semantics
NO_DECAY_SUBSTR = ("bias", "LayerNorm.bias", "LayerNorm.weight")       # optim/misc.py:14


def decays(name: str) -> bool:
    """optim/misc.py:15-22 -- substring match, so `layer_norm.weight`, `net.2.weight` DO get decay."""
    return not any(nd in name for nd in NO_DECAY_SUBSTR)


def warmup_linear(step: int, warmup: int, total: int) -> float:
    """optim/sched.py:17-21."""
    if step < warmup:
        return step / warmup
    return max(0, (total - step) / (total - warmup))


def lr_at(step: int, lr: float, warmup: int, total: int) -> float:
    """optim/sched.py:24-30."""
    v = lr * warmup_linear(step, warmup, total)
    return v if v > 0 else 1e-8


def clip_grad_norm(grads: List[Tensor], max_norm: float) -> Tensor:
    """torch.nn.utils.clip_grad_norm_ as used at main_r2r.py:271-273 (L2, coef clamped to 1)."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adamw_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], state: dict, lr: float,
               betas=(0.9, 0.98), eps: float = 1e-6, weight_decay: float = 0.01) -> None:
    """optim/adamw.py:53-112, in place.  denom = sqrt(v)+eps (:91), bias-corrected step (:93-97),
    decoupled decay AFTER the update using the *updated* p (:109-110)."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    step_size = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            continue
        m = state.setdefault("m", {}).setdefault(k, torch.zeros_like(p))
        v = state.setdefault("v", {}).setdefault(k, torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        p.addcdiv_(m, v.sqrt().add_(eps), value=-step_size)
        if decays(k) and weight_decay > 0:
            p.add_(p, alpha=-lr * weight_decay)


# =========================================================================== ViT-B/16 backbone
@dataclass
class VitConfig:
    """vision_transformer.py:236-239 defaults of vit_base_patch16_224 (:486-493): 224/16, 768 wide, 12 deep, 12 heads."""
    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: float = 4.0

    @classmethod
    def tiny(cls, **kw):
        d = dict(img_size=32, patch_size=8, in_chans=3, embed_dim=128, depth=2, num_heads=2, mlp_ratio=2.0)
        d.update(kw)
        return cls(**d)


def vit_param_shapes(c: VitConfig) -> Dict[str, tuple]:
    """state_dict of VisionTransformer without classifier head (num_classes=0), in module order (:266-283)."""
    D, P = c.embed_dim, c.patch_size
    n_tok = (c.img_size // P) ** 2 + 1
    Hm = int(D * c.mlp_ratio)
    sh = {"cls_token": (1, 1, D), "pos_embed": (1, n_tok, D),
          "patch_embed.proj.weight": (D, c.in_chans, P, P), "patch_embed.proj.bias": (D,)}
    for i in range(c.depth):
        b = f"blocks.{i}."
        sh.update({b + "norm1.weight": (D,), b + "norm1.bias": (D,), b + "attn.qkv.weight": (3 * D, D), b + "attn.qkv.bias": (3 * D,),
                   b + "attn.proj.weight": (D, D), b + "attn.proj.bias": (D,), b + "norm2.weight": (D,), b + "norm2.bias": (D,),
                   b + "mlp.fc1.weight": (Hm, D), b + "mlp.fc1.bias": (Hm,), b + "mlp.fc2.weight": (D, Hm), b + "mlp.fc2.bias": (D,)})
    sh.update({"norm.weight": (D,), "norm.bias": (D,)})
    return sh


def make_vit_state_dict(c: VitConfig, seed: int = 0) -> Dict[str, Tensor]:
    """numpy-PCG64 recipe like make_state_dict: N(0, 0.02) weights, norm weights 1 + 0.1 N, nothing left at 0 / 1."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for k, shp in vit_param_shapes(c).items():
        a = rng.standard_normal(size=shp, dtype=np.float32)
        is_norm_w = k.endswith("weight") and (".norm" in k or k.startswith("norm"))
        sd[k] = torch.from_numpy((1.0 + 0.1 * a) if is_norm_w else 0.02 * a)
    return sd


def vit_forward_features(sd: Dict[str, Tensor], c: VitConfig, images: Tensor) -> Tensor:
    """VisionTransformer.forward_features (vision_transformer.py:335-348) in eval / zero-dropout form: conv patch embed
    (:201-223), cls token + position embedding (:337-342), `depth` pre-LN blocks (:181-198: x += attn(norm1 x);
    x += mlp(norm2 x)), final LayerNorm eps 1e-6 (:265), return the cls row."""
    D, nh = c.embed_dim, c.num_heads
    x = F.conv2d(images, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=c.patch_size)   # :220
    x = x.flatten(2).transpose(1, 2)
    B, N, _ = x.shape
    x = torch.cat([sd["cls_token"].expand(B, -1, -1), x], 1) + sd["pos_embed"]                                  # :337-342
    for i in range(c.depth):
        b = f"blocks.{i}."
        y = F.layer_norm(x, (D,), sd[b + "norm1.weight"], sd[b + "norm1.bias"], 1e-6)
        S = y.shape[1]
        qkv = F.linear(y, sd[b + "attn.qkv.weight"], sd[b + "attn.qkv.bias"]).reshape(B, S, 3, nh, D // nh).permute(2, 0, 3, 1, 4)  # :167
        att = (qkv[0] @ qkv[1].transpose(-2, -1)) * (D // nh) ** -0.5                                            # :170
        att = att.softmax(-1)
        y = (att @ qkv[2]).transpose(1, 2).reshape(B, S, D)                                                     # :174
        x = x + F.linear(y, sd[b + "attn.proj.weight"], sd[b + "attn.proj.bias"])                               # :196
        y = F.layer_norm(x, (D,), sd[b + "norm2.weight"], sd[b + "norm2.bias"], 1e-6)
        y = F.linear(F.gelu(F.linear(y, sd[b + "mlp.fc1.weight"], sd[b + "mlp.fc1.bias"])), sd[b + "mlp.fc2.weight"], sd[b + "mlp.fc2.bias"])
        x = x + y                                                                                               # :197
    x = F.layer_norm(x, (D,), sd["norm.weight"], sd["norm.bias"], 1e-6)                                         # :344
    return x[:, 0]                                                                                              # :346


def a2c_loss_ref(policy_log_probs, hidden_values, rewards, masks, last_value, ended, entropys=None, gamma=0.9,
                 entropy_loss_weight=0.01, normalize_loss="total"):
    """The agent's A2C loss, statement by statement (finetune_src/r2r/agent_cmt.py:476-518).  Per-step lists as the rollout
    collects them: policy_log_probs[t] (B,), hidden_values[t] = critic(hidden_states[t]) (B,) -- the reference calls the critic
    inside the loop (:491) --, rewards[t] / masks[t] numpy (B,), last_value (B,) the critic's value of the last state
    (detached, :478), ended (B,) bool, entropys[t] (B,) when feedback == 'sample' (:497-498).  Returns (rl_loss, logs)."""
    import numpy as np
    batch_size = len(ended)
    discount_reward = np.zeros(batch_size, np.float32)                  # :479
    for i in range(batch_size):
        if not ended[i]:                                                 # :481-482
            discount_reward[i] = float(last_value[i])
    rl_loss = 0.0
    total = 0
    logs = {"critic_loss": [], "policy_loss": []}
    for t in range(len(rewards) - 1, -1, -1):                           # :486
        discount_reward = discount_reward * gamma + rewards[t]           # :487
        mask_ = torch.from_numpy(masks[t])
        r_ = torch.from_numpy(discount_reward.copy())
        v_ = hidden_values[t]
        a_ = (r_ - v_).detach()                                          # :493
        t_policy_loss = (-policy_log_probs[t] * a_ * mask_).sum()        # :495
        t_critic_loss = (((r_ - v_) ** 2) * mask_).sum() * 0.5           # :496
        rl_loss = rl_loss + t_policy_loss + t_critic_loss
        if entropys is not None:
            rl_loss = rl_loss + (-entropy_loss_weight * entropys[t] * mask_).sum()    # :498
        logs["critic_loss"].append(float(t_critic_loss))
        logs["policy_loss"].append(float(t_policy_loss))
        total = total + np.sum(masks[t])                                 # :503
    if normalize_loss == "total":                                        # :507-513
        rl_loss = rl_loss / float(total)
    elif normalize_loss == "batch":
        rl_loss = rl_loss / batch_size
    else:
        assert normalize_loss == "none"
    return rl_loss, logs
