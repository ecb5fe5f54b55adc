"""MI355X-native mirror of the reference's ``pretrain_src/model/vilmodel.py`` class surface.

Same class names, constructor arguments, ``forward`` signatures and parameter names
(so ``state_dict`` / ``named_parameters`` / ``set_dropout`` / DDP see what they see in the reference),
but every ``forward`` lowers to hand-written HIP kernels through ``vln_hamt_amd.ops`` -- the
``nn.Linear`` / ``nn.LayerNorm`` / ``nn.Embedding`` / ``nn.Dropout`` children are parameter (and
dropout-probability) containers only and are never *called*.

Reference lines are cited per class (paths relative to /root/reference/pretrain_src/model/).
"""
from __future__ import annotations

import copy
import math
import os

import numpy as np
import torch
from torch import nn

from .. import blocks, ops, streams
from ..modeling import HamtPreTrainedModel, precision_of

BertLayerNorm = torch.nn.LayerNorm
BertPreTrainedModel = HamtPreTrainedModel

_ACT = {"gelu": ops.ACT_GELU, "relu": ops.ACT_RELU}


def gelu(x):
    """erf GELU (vilmodel.py:23-29); kept for API parity -- the fused kernels apply it in the GEMM epilogue."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _p(drop: nn.Dropout, module: nn.Module) -> float:
    return float(drop.p) if module.training else 0.0


def _act_code(config) -> int:
    act = config.hidden_act
    if not isinstance(act, str) or act not in _ACT:
        raise ValueError(f"hidden_act={act!r}: the HIP epilogues implement 'gelu' (erf) and 'relu'")
    return _ACT[act]


class BertEmbeddings(nn.Module):
    """word + position + token-type lookup, LayerNorm, dropout (vilmodel.py:40-69)."""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids, token_type_ids=None, position_ids=None):
        B, L = input_ids.shape
        H = self.word_embeddings.weight.shape[1]
        if token_type_ids is None and position_ids is None:
            z = ops.embed_sum(input_ids, self.word_embeddings.weight, self.position_embeddings.weight,
                              self.token_type_embeddings.weight)
        else:  # explicit ids: three gathers (same association order as the reference's sum)
            if position_ids is None:
                position_ids = torch.arange(L, dtype=torch.long, device=input_ids.device)[None].expand(B, L)
            if token_type_ids is None:
                token_type_ids = torch.zeros_like(input_ids)
            z = ops.gather_rows(self.word_embeddings.weight, input_ids)
            z = ops.gather_rows(self.position_embeddings.weight, position_ids.expand(B, L), base=z)
            z = ops.gather_rows(self.token_type_embeddings.weight, token_type_ids, base=z).view(B, L, H)
        return ops.layer_norm(z, None, self.LayerNorm, p_post=_p(self.dropout, self), want16=blocks.ENABLED and H % 64 == 0)


class BertSelfAttention(nn.Module):
    """Q/K/V projections + fused softmax attention (vilmodel.py:72-129)."""

    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.output_attentions = config.output_attentions
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)
        self.prec = precision_of(config)

    def forward(self, hidden_states, attention_mask, head_mask=None):
        if head_mask is not None:
            raise NotImplementedError("head_mask is never used on the HAMT path (always None)")
        B, S, H = hidden_states.shape
        qkv = ops.packed_linear(hidden_states, self.prec, self.query, self.key, self.value)     # [B*S, 3H]
        ctx = ops.attention(qkv, None, attention_mask, B, self.num_attention_heads, _p(self.dropout, self), self.prec)
        ctx = ctx.view(B, S, H)
        # the fused kernel never materialises the S x S probabilities; the reference only returns them when
        # config.output_attentions is set and no caller on the path reads them
        return (ctx, None) if self.output_attentions else (ctx,)


class BertSelfOutput(nn.Module):
    """dense -> dropout -> LayerNorm(x + residual) (vilmodel.py:132-143)."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.prec = precision_of(config)

    def forward(self, hidden_states, input_tensor):
        y = ops.linear(hidden_states, self.dense.weight, self.dense.bias, ops.ACT_NONE, self.prec)
        return ops.layer_norm(y, input_tensor, self.LayerNorm, p_pre=_p(self.dropout, self))


class BertAttention(nn.Module):
    """vilmodel.py:146-156."""

    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, attention_mask, head_mask=None):
        packed = input_tensor.dim() == 2 and getattr(input_tensor, "_hamt_seq", None) is not None
        if head_mask is None and (input_tensor.dim() == 3 or packed) and blocks.usable(self.self.prec, input_tensor):
            # bf16 path: the whole sub-block is one autograd Function (vln_hamt_amd/blocks.py)
            y = blocks.self_attn_block(input_tensor, attention_mask, self.self, self.output, self.training)
            return (y, None) if self.self.output_attentions else (y,)
        so = self.self(input_tensor, attention_mask, head_mask)
        return (self.output(so[0], input_tensor),) + so[1:]


class BertIntermediate(nn.Module):
    """dense + erf-GELU in the GEMM epilogue (vilmodel.py:159-171)."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        self.act = _act_code(config)
        self.prec = precision_of(config)

    def forward(self, hidden_states):
        return ops.linear(hidden_states, self.dense.weight, self.dense.bias, self.act, self.prec)


class BertOutput(nn.Module):
    """vilmodel.py:174-185."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.prec = precision_of(config)

    def forward(self, hidden_states, input_tensor):
        y = ops.linear(hidden_states, self.dense.weight, self.dense.bias, ops.ACT_NONE, self.prec)
        return ops.layer_norm(y, input_tensor, self.LayerNorm, p_pre=_p(self.dropout, self))


class BertLayer(nn.Module):
    """post-LN transformer block (vilmodel.py:188-201)."""

    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, attention_mask, head_mask=None):
        att = self.attention(hidden_states, attention_mask, head_mask)
        if blocks.usable(self.output.prec, att[0]) and self.intermediate.act == ops.ACT_GELU:
            out = blocks.ffn_block(att[0], self.intermediate, self.output, self.training)
        else:
            out = self.output(self.intermediate(att[0]), att[0])
        return (out,) + att[1:]


class BertEncoder(nn.Module):
    """stack of BertLayer (vilmodel.py:204-234)."""
    _hamt_container = True      # forward reads parameters only through its layers' __call__ (optim.AdamW.attach)

    def __init__(self, config):
        super().__init__()
        self.output_attentions = config.output_attentions
        self.output_hidden_states = config.output_hidden_states
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])

    def forward(self, hidden_states, attention_mask, head_mask=None):
        all_hidden, all_att = (), ()
        for i, layer in enumerate(self.layer):
            if self.output_hidden_states:
                all_hidden += (hidden_states,)
            outs = layer(hidden_states, attention_mask, None if head_mask is None else head_mask[i])
            hidden_states = outs[0]
            if self.output_attentions:
                all_att += (outs[1],)
        if self.output_hidden_states:
            all_hidden += (hidden_states,)
        res = (hidden_states,)
        if self.output_hidden_states:
            res += (all_hidden,)
        if self.output_attentions:
            res += (all_att,)
        return res


class BertPredictionHeadTransform(nn.Module):
    """dense -> GELU -> LayerNorm (vilmodel.py:252-266)."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.act = _act_code(config)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.prec = precision_of(config)

    def forward(self, hidden_states):
        h = ops.linear(hidden_states, self.dense.weight, self.dense.bias, self.act, self.prec)
        return ops.layer_norm(h, None, self.LayerNorm)


class BertLMPredictionHead(nn.Module):
    """transform + decoder tied to the word embeddings + output bias (vilmodel.py:269-285)."""

    def __init__(self, config):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))
        self.prec = precision_of(config)

    def forward(self, hidden_states):
        h = self.transform(hidden_states)
        return ops.linear(h, self.decoder.weight, self.bias, ops.ACT_NONE, self.prec)


class BertOnlyMLMHead(nn.Module):
    """vilmodel.py:288-295."""

    def __init__(self, config):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)

    def forward(self, sequence_output):
        return self.predictions(sequence_output)


class BertOutAttention(nn.Module):
    """cross-attention: queries from `hidden_states`, keys/values from `context` (vilmodel.py:298-349)."""

    def __init__(self, config, ctx_dim=None):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        if ctx_dim is None:
            ctx_dim = config.hidden_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(ctx_dim, self.all_head_size)
        self.value = nn.Linear(ctx_dim, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)
        self.prec = precision_of(config)

    def forward(self, hidden_states, context, attention_mask=None):
        B, Sq, H = hidden_states.shape
        q = ops.linear(hidden_states.reshape(B * Sq, H), self.query.weight, self.query.bias, ops.ACT_NONE, self.prec)
        kv = ops.packed_linear(context, self.prec, self.key, self.value)                         # [B*Sk, 2H]
        ctx = ops.attention(q, kv, attention_mask, B, self.num_attention_heads, _p(self.dropout, self), self.prec)
        return ctx.view(B, Sq, H)


class BertXAttention(nn.Module):
    """vilmodel.py:351-360."""

    def __init__(self, config, ctx_dim=None):
        super().__init__()
        self.att = BertOutAttention(config, ctx_dim=ctx_dim)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, ctx_tensor, ctx_att_mask=None):
        packed = input_tensor.dim() == 2 and getattr(input_tensor, "_hamt_seq", None) is not None      # (the packed text of a ragged batch)
        if (input_tensor.dim() == 3 or packed) and blocks.usable(self.att.prec, input_tensor):
            return blocks.cross_attn_block(input_tensor, ctx_tensor, ctx_att_mask, self.att, self.output, self.training)
        return self.output(self.att(input_tensor, ctx_tensor, ctx_att_mask), input_tensor)


def _ffn(inter, out, x, training):
    """BertIntermediate + BertOutput, fused into one block Function on the bf16 path."""
    if blocks.usable(out.prec, x) and inter.act == ops.ACT_GELU:
        return blocks.ffn_block(x, inter, out, training)
    return out(inter(x), x)


# HAMT_NO_X_PACK=1: a packed text stream (ragged batches, NavPreTrainedModel._text) is scattered back to [B, L, H] BEFORE the
# cross-modal layers instead of behind them (measurement switch)
X_PACK = os.environ.get("HAMT_NO_X_PACK") != "1"
# the side of the LAST cross-modal layer whose output the task head does not read is not computed (LXRTXLayer.forward `need`); HAMT_NO_DCE=1: always both
DEAD_SIDE_ELIMINATION = os.environ.get("HAMT_NO_DCE") is None
PACK_MAX_LEN = 128      # longest sequence hamt_attn_varlen_* / hamt_attn_varlen_cross_* serve (include/hamt.h)


def _rows_of(x, rows):
    """x [B, S, H] (or a packed [M, H]) -> its rows `rows` as [R, H]; x itself when rows is None."""
    if rows is None:
        return x
    return ops.gather_rows(x.reshape(-1, x.shape[-1]), rows, unique=True)


class LXRTXLayer(nn.Module):
    """LXMERT cross-modality layer: ONE shared cross-attention applied in both directions on the
    pre-update inputs, then per-stream self-attention and FFN (vilmodel.py:362-412)."""

    def __init__(self, config):
        super().__init__()
        self.lang_self_att = BertAttention(config)
        self.lang_inter = BertIntermediate(config)
        self.lang_output = BertOutput(config)
        self.visn_self_att = BertAttention(config)
        self.visn_inter = BertIntermediate(config)
        self.visn_output = BertOutput(config)
        self.visual_attention = BertXAttention(config)

    def cross_att(self, lang_input, lang_attention_mask, visn_input, visn_attention_mask):
        xa = self.visual_attention
        lang_att = xa(lang_input, visn_input, ctx_att_mask=visn_attention_mask)
        visn_att = xa(visn_input, lang_input, ctx_att_mask=lang_attention_mask)
        return lang_att, visn_att

    def self_att(self, lang_input, lang_attention_mask, visn_input, visn_attention_mask):
        return (self.lang_self_att(lang_input, lang_attention_mask),
                self.visn_self_att(visn_input, visn_attention_mask))

    def output_fc(self, lang_input, visn_input):
        return _ffn(self.lang_inter, self.lang_output, lang_input, self.training), _ffn(self.visn_inter, self.visn_output, visn_input, self.training)

    def forward(self, lang_feats, lang_attention_mask, visn_feats, visn_attention_mask, need="both", lang_rows=None, visn_rows=None):
        """`need` = "lang" / "visn" and `lang_rows` / `visn_rows` are for the LAST cross-modal layer only (LxmertEncoder.forward): what
        the caller reads of this layer's outputs.  After the shared cross-attention the two sides are independent chains, and behind a
        side's self-attention its feed-forward block is row-wise -- so a side nobody reads (`need`) is not launched at all (None is
        returned in its place), and of a side of which only some rows are read (`*_rows`: int64 flat row indices into its [B * S, H]
        self-attention output -- the [CLS] rows, the masked positions) the feed-forward block runs on those rows only and [R, H] is
        returned.  Dead code in the reference (it computes everything and indexes afterwards; autograd never ran the unread rows'
        backward either): every result is unchanged."""
        if need == "lang":
            lang = self.visual_attention(lang_feats, visn_feats, ctx_att_mask=visn_attention_mask)
            lang = self.lang_self_att(lang[0] if isinstance(lang, tuple) else lang, lang_attention_mask)
            return _ffn(self.lang_inter, self.lang_output, _rows_of(lang[0], lang_rows), self.training), None
        if need == "visn":
            visn = self.visual_attention(visn_feats, lang_feats, ctx_att_mask=lang_attention_mask)
            visn = self.visn_self_att(visn[0] if isinstance(visn, tuple) else visn, visn_attention_mask)
            return None, _ffn(self.visn_inter, self.visn_output, _rows_of(visn[0], visn_rows), self.training)
        if lang_feats.is_cuda and streams.two_stream_enabled():
            return self._forward_two_streams(lang_feats, lang_attention_mask, visn_feats, visn_attention_mask, lang_rows, visn_rows)
        lang, visn = self.cross_att(lang_feats, lang_attention_mask, visn_feats, visn_attention_mask)
        lang, visn = self.self_att(lang, lang_attention_mask, visn, visn_attention_mask)
        return self.output_fc(_rows_of(lang[0], lang_rows), _rows_of(visn[0], visn_rows))

    def _forward_two_streams(self, lang_feats, lang_mask, visn_feats, visn_mask, lang_rows=None, visn_rows=None):
        """Same computation with the vision-side chain (cross <- lang, self, FFN) on a second stream: after the shared
        cross-attention both sides are independent until the next layer, and the vision side is a chain of tiny kernels
        (6 history tokens per sample when there is no observation) that would otherwise leave the chip mostly idle.
        autograd replays each backward node on its forward stream, so backward overlaps the same way."""
        main = torch.cuda.current_stream()
        side = streams.side_stream(lang_feats.device)
        xa = self.visual_attention
        # the shared cross-attention is applied on BOTH streams: build its lazily cached bf16 weight images (only used when
        # the optimizer's bf16 arena is not there) on `main` before the fork, not concurrently on whichever stream is first
        for lin in (xa.att.query, xa.att.key, xa.att.value, xa.output.dense):
            ops.weight_operand(lin.weight, xa.att.prec)
        streams.fork(main, side)
        for t in (lang_feats, visn_feats, lang_mask, visn_mask, visn_rows):
            streams.share(t, side)                      # inputs produced on `main`, read by the vision-side kernels
        with torch.cuda.stream(side):
            visn = self.visual_attention(visn_feats, lang_feats, ctx_att_mask=lang_mask)
            visn = self.visn_self_att(visn[0] if isinstance(visn, tuple) else visn, visn_mask)
            visn_out = _ffn(self.visn_inter, self.visn_output, _rows_of(visn[0], visn_rows), self.training)
        lang = self.visual_attention(lang_feats, visn_feats, ctx_att_mask=visn_mask)
        lang = self.lang_self_att(lang[0] if isinstance(lang, tuple) else lang, lang_mask)
        lang_out = _ffn(self.lang_inter, self.lang_output, _rows_of(lang[0], lang_rows), self.training)
        streams.join(main, side)
        streams.share(visn_out, main)                   # produced on `side`, consumed on `main` from here on
        return lang_out, visn_out


class LxmertEncoder(nn.Module):
    """text layers, then cross-modal layers over text <-> {history (+) observation} (vilmodel.py:414-478)."""
    _hamt_container = True

    def __init__(self, config):
        super().__init__()
        self.num_l_layers = config.num_l_layers
        self.num_r_layers = config.num_r_layers
        self.num_h_layers = config.num_h_layers
        self.num_x_layers = config.num_x_layers
        self.update_lang_bert = config.update_lang_bert
        # `layer` (not l_layers) so plain BERT checkpoints load, as in the reference (:424)
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(self.num_l_layers)])
        self.h_layers = nn.ModuleList([BertLayer(config) for _ in range(self.num_h_layers)]) if self.num_h_layers > 0 else None
        self.r_layers = nn.ModuleList([BertLayer(config) for _ in range(self.num_r_layers)]) if self.num_r_layers > 0 else None
        self.x_layers = nn.ModuleList([LXRTXLayer(config) for _ in range(self.num_x_layers)])

    def text_layers(self, txt_embeds, extended_txt_masks):
        for layer in self.layer:
            txt_embeds = layer(txt_embeds, extended_txt_masks)[0]
        return txt_embeds

    def forward(self, txt_embeds, extended_txt_masks, hist_embeds, extended_hist_masks,
                img_embeds=None, extended_img_masks=None, text_done=False, need="both", lang_rows=None):
        """`need` = "lang" / "visn": the caller reads only the text output / only the history + observation outputs (the MLM and SAR heads;
        the MRC and SPREL heads) -- the unread side of the last cross-modal layer is not computed and returned as None.  `lang_rows` (int64
        flat positions b * L + l): the caller reads only these text rows (the masked positions, the [CLS] rows) -- the text output is then
        [R, H], those rows in that order, and the last layer's feed-forward block ran on them only (LXRTXLayer.forward)."""
        if not text_done:       # (the caller may have run them already, next to the vision-side embedders)
            txt_embeds = self.text_layers(txt_embeds, extended_txt_masks)
        # a PACKED text stream [M, H] (NavPreTrainedModel._text(keep_packed=True): the real tokens of a ragged batch back to back) stays
        # packed through the cross-modal layers -- their attentions take one packed side (blocks.CrossAttnBlockFn), the rest is
        # row-wise -- and is scattered back into [B, L, H] behind the last one
        seq, unpack = getattr(txt_embeds, "_hamt_seq", None), getattr(txt_embeds, "_hamt_unpack", None)
        if not self.update_lang_bert:
            txt_embeds = txt_embeds.detach()
            if seq is not None:
                txt_embeds._hamt_seq = seq
        if img_embeds is not None and self.r_layers is not None:
            for layer in self.r_layers:
                img_embeds = layer(img_embeds, extended_img_masks)[0]
        if self.h_layers is not None:
            for layer in self.h_layers:
                hist_embeds = layer(hist_embeds, extended_hist_masks)[0]
        n_hist = hist_embeds.size(1)
        if img_embeds is None:
            vis, vis_masks = hist_embeds, extended_hist_masks
        else:
            vis = torch.cat([hist_embeds, img_embeds], 1)
            vis_masks = torch.cat([extended_hist_masks, extended_img_masks], -1)
        if unpack is not None and vis.size(1) > PACK_MAX_LEN:
            # more visual tokens than the packed cross attention holds per sample (candidate views on top of the panorama): back to the
            # padded layout in front of the cross-modal layers
            txt_embeds = ops.gather_rows(txt_embeds, unpack[0]).view(unpack[1], unpack[2], -1)
            unpack = None
        last = len(self.x_layers) - 1
        if not DEAD_SIDE_ELIMINATION or need == "visn":
            lang_rows = None
        if lang_rows is not None and unpack is not None:
            lang_rows = unpack[0].index_select(0, lang_rows)          # padded position -> row of the packed text
        for i, layer in enumerate(self.x_layers):
            if i == last and DEAD_SIDE_ELIMINATION:
                txt_embeds, vis = layer(txt_embeds, extended_txt_masks, vis, vis_masks, need=need, lang_rows=lang_rows)
            else:
                txt_embeds, vis = layer(txt_embeds, extended_txt_masks, vis, vis_masks)
        if unpack is not None and txt_embeds is not None and lang_rows is None:
            txt_embeds = ops.gather_rows(txt_embeds, unpack[0]).view(unpack[1], unpack[2], -1)
        if vis is None:
            return txt_embeds, None, None
        hist_embeds = vis[:, :n_hist]
        if img_embeds is not None:
            img_embeds = vis[:, n_hist:]
        return txt_embeds, hist_embeds, img_embeds


def _bcast_rows(n_outer: int, n_inner: int, device):
    """row -> outer index map for broadcasting a (n_outer, H) table over n_inner rows each."""
    return ops.const_index("bcast", int(n_outer), int(n_inner), device=device)


class _VisualLinears(nn.Module):
    """shared helper: LN(Linear(img)) + LN(Linear(ang)) used by both embedders."""

    @staticmethod
    def two_stream(img_lin, img_ln, ang_lin, ang_ln, img, ang, prec, want16=False):
        if ops.vis_embed_ok(img, ang, img_lin, ang_lin):      # one launch behind the dense layer (csrc/vis_embed.hip)
            return ops.vis_embed(img, ang, img_lin, img_ln, ang_lin, ang_ln, prec, want16)
        a = ops.layer_norm(ops.linear(img, img_lin.weight, img_lin.bias, ops.ACT_NONE, prec), None, img_ln)
        # K = angle_feat_size (4): exact fp32 contraction, it is 4 FMAs per output
        b = ops.layer_norm(ops.linear(ang, ang_lin.weight, ang_lin.bias, ops.ACT_NONE, "fp32"), None, ang_ln)
        return ops.add3(a, b)


class ImageEmbeddings(nn.Module):
    """observation embedding: LN(W img) + LN(W ang) + token-type + nav-type -> LN -> dropout (vilmodel.py:482-505)."""

    def __init__(self, config):
        super().__init__()
        self.img_linear = nn.Linear(config.image_feat_size, config.hidden_size)
        self.img_layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.ang_linear = nn.Linear(config.angle_feat_size, config.hidden_size)
        self.ang_layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.nav_type_embedding = nn.Embedding(3, config.hidden_size)   # 0 non-navigable, 1 navigable, 2 stop
        self.layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.prec = precision_of(config)

    def forward(self, img_feat, ang_feat, type_embeddings, nav_types=None):
        B, S = img_feat.shape[:2]
        H = self.img_linear.weight.shape[0]
        e = _VisualLinears.two_stream(self.img_linear, self.img_layer_norm, self.ang_linear, self.ang_layer_norm,
                                      img_feat, ang_feat, self.prec)
        te = type_embeddings.reshape(-1, H)
        if te.shape[0] == B * S:
            e = ops.add3(e, te.view(B, S, H))
        else:                                                    # (B,1,H) broadcast over the views
            idx = _bcast_rows(te.shape[0], (B * S) // te.shape[0], img_feat.device)
            e = ops.gather_rows(te, idx, base=e)
        if nav_types is not None:
            e = ops.gather_rows(self.nav_type_embedding.weight, nav_types, base=e)
        return ops.layer_norm(e.view(B, S, H), None, self.layer_norm, p_post=_p(self.dropout, self))


class HistoryEmbeddings(nn.Module):
    """hierarchical history encoder (vilmodel.py:507-575): per step LN(W img)+LN(W ang)+type, plus a
    `num_h_pano_layers`-layer BERT over the step's 36 panorama views mean-pooled to ONE token, plus the step
    position; a learned cls token in front."""

    def __init__(self, config):
        super().__init__()
        self.cls_token = nn.Parameter(torch.zeros(1, 1, config.hidden_size))
        self.img_linear = nn.Linear(config.image_feat_size, config.hidden_size)
        self.img_layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.ang_linear = nn.Linear(config.angle_feat_size, config.hidden_size)
        self.ang_layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
        if config.num_h_pano_layers > 0:
            self.pano_img_linear = nn.Linear(config.image_feat_size, config.hidden_size)
            self.pano_img_layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
            self.pano_ang_linear = nn.Linear(config.angle_feat_size, config.hidden_size)
            self.pano_ang_layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
            pano_cfg = copy.copy(config)
            pano_cfg.num_hidden_layers = config.num_h_pano_layers
            self.pano_encoder = BertEncoder(pano_cfg)
        else:
            self.pano_encoder = None
        self.position_embeddings = nn.Embedding(config.max_action_steps, config.hidden_size)
        self.type_embedding = nn.Embedding(1, config.hidden_size)
        self.layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.prec = precision_of(config)

    @property
    def device(self):
        return self.cls_token.device

    def add_position(self, embeddings, pos_ids):
        """LN(dropout-free sum + position) then dropout (vilmodel.py:568-571; also used by forward_itm :669-670)."""
        B, T, H = embeddings.shape
        idx = pos_ids.expand(B, T) if pos_ids.dim() == 2 else pos_ids[None].expand(B, T)
        e = ops.gather_rows(self.position_embeddings.weight, idx, base=embeddings).view(B, T, H)
        return ops.layer_norm(e, None, self.layer_norm, p_post=_p(self.dropout, self))

    def forward(self, img_feats, ang_feats, pano_img_feats, pano_ang_feats, pos_ids=None, batch_size=None):
        dev = self.device
        H = self.cls_token.shape[-1]
        zeros_b = ops.const_index("zeros", int(batch_size), device=dev)
        cls = ops.gather_rows(self.cls_token, zeros_b)        # (the parameter itself, [1, 1, H]: its gradient rows go straight into its arena slot)
        cls = ops.gather_rows(self.type_embedding.weight, zeros_b, base=cls)
        cls = ops.layer_norm(cls.view(batch_size, 1, H), None, self.layer_norm, p_post=_p(self.dropout, self))
        if img_feats is None:
            return cls, None
        B, T = img_feats.shape[:2]
        e = _VisualLinears.two_stream(self.img_linear, self.img_layer_norm, self.ang_linear, self.ang_layer_norm,
                                      img_feats, ang_feats, self.prec)
        e = ops.gather_rows(self.type_embedding.weight, ops.const_index("zeros", int(B * T), device=dev), base=e)
        if self.pano_encoder is not None:
            V = pano_img_feats.shape[2]
            pe = _VisualLinears.two_stream(self.pano_img_linear, self.pano_img_layer_norm, self.pano_ang_linear,
                                           self.pano_ang_layer_norm, pano_img_feats.reshape(B * T, V, -1),
                                           pano_ang_feats.reshape(B * T, V, -1), self.prec, want16=self.prec == "bf16")
            # all 36 views exist: the reference's mask is all-zero (:560) == no mask
            if pe.shape != (B * T, V, H):
                pe = pe.view(B * T, V, H)
            pe = self.pano_encoder(pe, None)[0]
            e = ops.add3(e.view(B * T, H), ops.mean_mid(pe))
        e = e.view(B, T, H)
        if pos_ids is not None:
            e = self.add_position(e, pos_ids)
        return cls, e


class NavPreTrainedModel(BertPreTrainedModel):
    """trunk: text / history / observation embedders + LxmertEncoder (vilmodel.py:578-724)."""
    _hamt_container = True      # (its one direct parameter read, the token-type row of the observation tokens, is gated explicitly)

    def __init__(self, config):
        super().__init__(config)
        self.embeddings = BertEmbeddings(config)
        self.img_embeddings = ImageEmbeddings(config)
        self.hist_embeddings = HistoryEmbeddings(config)
        self.encoder = LxmertEncoder(config)
        self.init_weights()

    def _text(self, txt_ids, txt_m, keep_packed=False):
        """Text embedder + the text-only layers (vilmodel.py:601, 441-443).  With a packing plan on `txt_ids` (`_hamt_pack` = (pack_idx
        [M], cu_seqlens int32 [n + 1], unpack_idx [B L]), put there by MultiStepNavCMTPreTraining.forward from the batch's `txt_pack_idx`
        / `txt_cu` / `txt_unpack_idx`) the nine layers run on the REAL tokens only -- the instructions back to back, self-attention per
        sequence (hamt_attn_varlen_*), everything else row-wise -- and the result is scattered back into the padded [B, L, H] layout
        for the cross-modal layers.  Padded positions get their sequence's first row: any finite value serves, they are masked as
        keys and nothing reads what they produce as queries (the reference computes them and throws them away)."""
        pack = getattr(txt_ids, "_hamt_pack", None)
        H = self.config.hidden_size      # (no parameter is touched here: optim.AdamW.attach's read gates sit in the child modules)
        # (the packed attention kernels hold one sequence per workgroup: instructions of up to 128 tokens -- R2R's 80; RxR pretraining
        # pads to 250, config/pretrain_rxr.json: such batches take the padded kernels, which serve up to 256 keys)
        if pack is None or txt_ids.shape[1] > PACK_MAX_LEN or not (blocks.ENABLED and precision_of(self.config) == "bf16" and txt_ids.is_cuda and H % 64 == 0):
            x = self.embeddings(txt_ids)
            for layer in self.encoder.layer:
                x = layer(x, txt_m)[0]
            return x
        pack_idx, cu, unpack_idx = pack
        B, L = txt_ids.shape
        ids = txt_ids.reshape(-1)[pack_idx]
        x = self.embeddings(ids[None], position_ids=(pack_idx % L)[None]).view(-1, H)
        x._hamt_seq = (cu, int(cu.shape[0]) - 1, L)
        for layer in self.encoder.layer:
            x = layer(x, None)[0]
        if keep_packed and X_PACK:      # (the cross-modal layers go on with the packed rows: LxmertEncoder.forward)
            x._hamt_unpack = (unpack_idx, B, L)
            return x
        return ops.gather_rows(x, unpack_idx).view(B, L, H)

    @staticmethod
    def _extend(mask):
        """(B,S) bool -> additive (B,1,1,S) = (1 - m) * -10000 (vilmodel.py:597-599)."""
        return ops.extend_mask(mask)

    def forward(self, txt_ids, txt_masks, hist_img_feats, hist_ang_feats, hist_pano_img_feats, hist_pano_ang_feats,
                hist_masks, ob_img_feats, ob_ang_feats, ob_nav_types, ob_masks, need="both", lang_rows=None):
        """`need`, `lang_rows` (not in the reference's signature; defaults = its behaviour): "lang" / "visn" when the caller reads only the
        text output / only the history + observation outputs; the text rows it reads -- see LxmertEncoder.forward.  With `lang_rows` the
        text output is [R, H] (2-D) unless HAMT_NO_DCE=1, in which case it is the full [B, L, H]: callers check `dim()`."""
        B = txt_ids.size(0)
        txt_m = self._extend(txt_masks)
        hist_m = self._extend(hist_masks)
        ob_m = self._extend(ob_masks) if ob_img_feats is not None else None

        def vision_side():
            step_ids = None
            if hist_img_feats is not None:
                step_ids = ops.const_index("arange", int(hist_img_feats.size(1)), device=txt_ids.device)[None]
            cls, steps = self.hist_embeddings(hist_img_feats, hist_ang_feats, hist_pano_img_feats, hist_pano_ang_feats, step_ids, batch_size=B)
            hist = cls if steps is None else torch.cat([cls, steps], 1)
            ob = None
            if ob_img_feats is not None:
                ones = ops.const_index("ones", int(B), device=txt_ids.device)
                streams.gate(self.embeddings.token_type_embeddings.weight)
                tt = ops.gather_rows(self.embeddings.token_type_embeddings.weight, ones).view(B, 1, -1)
                ob = self.img_embeddings(ob_img_feats, ob_ang_feats, tt, nav_types=ob_nav_types)
            return hist, ob

        if txt_ids.is_cuda and streams.two_stream_enabled("trunk"):
            # history / observation embedders (incl. the panorama encoder) on the second stream, next to the text embedder
            # and the text-only layers: the two chains do not meet before the first cross-modal layer
            main = torch.cuda.current_stream()
            side = streams.side_stream(txt_ids.device)
            streams.fork(main, side)
            for t in (hist_img_feats, hist_ang_feats, hist_pano_img_feats, hist_pano_ang_feats, ob_img_feats, ob_ang_feats, ob_nav_types):
                streams.share(t, side)
            with torch.cuda.stream(side):
                hist, ob = vision_side()
            txt = self._text(txt_ids, txt_m, keep_packed=True)
            streams.join(main, side)
            streams.share(hist, main)
            streams.share(ob, main)
            return self.encoder(txt, txt_m, hist, hist_m, ob, ob_m, text_done=True, need=need, lang_rows=lang_rows)
        txt = self._text(txt_ids, txt_m, keep_packed=True)
        hist, ob = vision_side()
        return self.encoder(txt, txt_m, hist, hist_m, ob, ob_m, text_done=True, need=need, lang_rows=lang_rows)

    def forward_itm(self, txt_ids, txt_masks, hist_img_feats, hist_ang_feats, hist_pano_img_feats, hist_pano_ang_feats,
                    hist_masks, num_neg_trajs=4, neg_idxs=None, shuffled_pos_ids=None):
        """vilmodel.py:640-724.  `neg_idxs` (B,K) / `shuffled_pos_ids` (list of (B,T)) inject the negatives;
        when None they are drawn like the reference does (np.random.choice :684, torch.randperm :698)."""
        B, T = hist_img_feats.shape[:2]
        dev = txt_ids.device
        n_rep = 1 + num_neg_trajs
        txt_m1 = self._extend(txt_masks)
        hist_m = self._extend(hist_masks)

        cls_rows = [None]

        def text_side():
            txt = self._text(txt_ids, txt_m1, keep_packed=True)
            if txt.dim() == 2:
                # PACKED text [Mb, H] (a ragged batch): n_rep copies back to back, each with the plan's sequences -- real ones and
                # fillers -- so the fillers lie BETWEEN the copies: the cross attentions get the pairing explicitly (`_hamt_pair`:
                # packed sequence r n + j <-> candidate sample r B + j for j < B, no keys for the fillers); everything else is row-wise.
                # Only the [CLS] row of each real sequence is read behind the last layer.
                cu, n, _ = txt._hamt_seq
                Mb, H = txt.shape
                r = torch.arange(n_rep, device=dev, dtype=torch.int32)[:, None]
                j = torch.arange(n, device=dev, dtype=torch.int32)[None]
                rep = txt[None].expand(n_rep, Mb, H).reshape(n_rep * Mb, H)      # (broadcast: the backward sums the copies in a fixed order)
                rep._hamt_seq = (torch.cat([(cu[:-1][None] + r * Mb).reshape(-1), cu[-1:] + (n_rep - 1) * Mb]), n_rep * n, txt._hamt_seq[2])
                rep._hamt_pair = (torch.where(j < B, r * B + j, torch.full_like(j, -1)).reshape(-1).contiguous(),
                                  (r * n + j[:, :B]).reshape(-1).contiguous())
                cls_rows[0] = (cu[:B][None] + r * Mb).reshape(-1).long()
                return rep, txt_m1.repeat(n_rep, 1, 1, 1)
            L, H = txt.shape[1:]
            # n_rep copies by broadcast: the backward is a sum over the copies in a fixed order (a gather of repeated row indices
            # would scatter-add them with atomics in whatever order they land)
            return txt[None].expand(n_rep, B, L, H).reshape(n_rep * B, L, H), txt_m1.repeat(n_rep, 1, 1, 1)

        def vision_side(neg_idxs, shuffled_pos_ids):
            cls, nopos = self.hist_embeddings(hist_img_feats, hist_ang_feats, hist_pano_img_feats, hist_pano_ang_feats, pos_ids=None, batch_size=B)
            hist = torch.cat([cls, self.hist_embeddings.add_position(nopos, torch.arange(T, device=dev)[None])], 1)
            if self.encoder.h_layers is not None:
                for layer in self.encoder.h_layers:
                    hist = layer(hist, hist_m)[0]
            neg_h, neg_m = [], []
            K = num_neg_trajs // 2
            if B > 1:
                if neg_idxs is None:
                    neg_idxs = torch.from_numpy(np.stack(
                        [np.random.choice([j for j in range(B) if j != i], K) for i in range(B)], 0)).to(dev)
                for k in range(K):
                    neg_h.append(ops.gather_rows(hist.reshape(B, -1), neg_idxs[:, k]).view(hist.shape))
                    neg_m.append(hist_m[neg_idxs[:, k]])
            else:
                K = num_neg_trajs
            if shuffled_pos_ids is None:
                lens = (hist_masks.sum(1) - 1).tolist()
                shuffled_pos_ids = []
                for _ in range(K):
                    rows = [torch.cat([torch.randperm(n), torch.arange(n, T, dtype=torch.long)], 0) for n in lens]
                    shuffled_pos_ids.append(torch.stack(rows, 0).to(dev))
            for pos in shuffled_pos_ids:
                sh = torch.cat([cls, self.hist_embeddings.add_position(nopos, pos)], 1)
                if self.encoder.h_layers is not None:
                    for layer in self.encoder.h_layers:
                        sh = layer(sh, hist_m)[0]
                neg_h.append(sh)
                neg_m.append(hist_m)
            return torch.cat([hist] + neg_h, 0), torch.cat([hist_m] + neg_m, 0)

        if txt_ids.is_cuda and streams.two_stream_enabled("trunk"):     # history side next to the text side, as in forward()
            main = torch.cuda.current_stream()
            side = streams.side_stream(dev)
            streams.fork(main, side)
            for t in (hist_img_feats, hist_ang_feats, hist_pano_img_feats, hist_pano_ang_feats, hist_m, neg_idxs,
                      *(shuffled_pos_ids or [])):
                streams.share(t, side)
            with torch.cuda.stream(side):
                vis, vis_m = vision_side(neg_idxs, shuffled_pos_ids)
            txt, txt_m = text_side()
            streams.join(main, side)
            streams.share(vis, main)
            streams.share(vis_m, main)
        else:
            txt, txt_m = text_side()
            vis, vis_m = vision_side(neg_idxs, shuffled_pos_ids)
        H = txt.shape[-1]
        xl = self.encoder.x_layers
        if DEAD_SIDE_ELIMINATION and len(xl) > 0:
            # only the [CLS] rows of both outputs are read: the last layer's feed-forward blocks run on those rows (LXRTXLayer.forward)
            for layer in xl[:-1]:
                txt, vis = layer(txt, txt_m, vis, vis_m)
            lang_rows = cls_rows[0] if cls_rows[0] is not None else ops.const_index("arange_mul", int(n_rep * B), int(txt.shape[1]), device=dev)
            visn_rows = ops.const_index("arange_mul", int(vis.shape[0]), int(vis.shape[1]), device=dev)
            cls, vis0 = xl[-1](txt, txt_m, vis, vis_m, lang_rows=lang_rows, visn_rows=visn_rows)
            fused = ops.mul_bcast(cls.view(n_rep * B, 1, H), vis0)
            return fused.view(n_rep, B, H).transpose(0, 1)
        for layer in xl:
            txt, vis = layer(txt, txt_m, vis, vis_m)
        cls = ops.gather_rows(txt, cls_rows[0]).view(n_rep * B, 1, H) if cls_rows[0] is not None else txt[:, :1].contiguous()
        fused = ops.mul_bcast(cls, vis[:, 0])                            # txt[:,0] * hist[:,0]
        return fused.view(n_rep, B, H).transpose(0, 1)                   # == stack(split(fused, B), 1)
