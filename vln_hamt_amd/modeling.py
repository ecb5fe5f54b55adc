"""Minimal stand-in for the HF `BertPreTrainedModel` base the reference subclasses
(pretrain_src/model/vilmodel.py:15, 578; pretrain_cmt.py:73; finetune_src/models/vilmodel_cmt.py:610).

It reproduces only what the reference's call sites use -- `config`, `init_weights()` (+`tie_weights`),
`_tie_or_clone_weights`, `device` / `dtype`, and `from_pretrained(None, config=..., state_dict=...)` with the
HF prefix rules for `base_model_prefix = "bert"` (main_r2r.py:146-148, vlnbert_init.py:65-68) -- without
depending on a particular transformers version.
"""
from __future__ import annotations

import json
import logging
import types

import torch
from torch import nn

logger = logging.getLogger(__name__)


class HamtConfig(types.SimpleNamespace):
    """Attribute bag with the keys of pretrain_src/config/r2r_model_config.json (any object with the same
    attributes, e.g. a transformers PretrainedConfig, works too)."""

    DEFAULTS = dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, vocab_size=30522,
                    type_vocab_size=2, max_position_embeddings=512, max_action_steps=100, image_feat_size=768,
                    angle_feat_size=4, image_prob_size=1000, num_l_layers=9, num_x_layers=4, num_h_pano_layers=2,
                    num_h_layers=0, num_r_layers=0, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
                    attention_probs_dropout_prob=0.1, pred_head_dropout_prob=0.1, hidden_act="gelu",
                    initializer_range=0.02, update_lang_bert=True, output_attentions=False,
                    output_hidden_states=False, num_hidden_layers=12, hamt_precision="bf16")

    def __init__(self, **kw):
        d = dict(self.DEFAULTS)
        d.update(kw)
        super().__init__(**d)

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls(**json.load(f))


class HamtPreTrainedModel(nn.Module):
    base_model_prefix = "bert"

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        self.config = config

    # ---- initialisation (HF-4.x BERT recipe)
    def _init_weights(self, module):
        std = getattr(self.config, "initializer_range", 0.02)
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.Embedding):
            module.weight._hamt_fp32_read = True      # looked up in fp32 (optim.adamw.shadow_only: not a bf16-only GEMM weight)
            module.weight.data.normal_(mean=0.0, std=std)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    def init_weights(self):
        self.apply(self._init_weights)
        self.tie_weights()

    def tie_weights(self):
        pass

    def _tie_or_clone_weights(self, output_embeddings, input_embeddings):
        output_embeddings.weight = input_embeddings.weight

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def set_precision(self, prec: str):
        """'bf16' (bf16 MFMA operands, fp32 accumulate) or 'fp32' (exact fp32 MFMA) for every contraction."""
        assert prec in ("bf16", "fp32")
        self.config.hamt_precision = prec
        for m in self.modules():
            if hasattr(m, "prec"):
                m.prec = prec
        return self

    # ---- loading
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, *model_args, config=None, state_dict=None, **kwargs):
        if pretrained_model_name_or_path is not None:
            state_dict = torch.load(pretrained_model_name_or_path, map_location="cpu")
        model = cls(config, *model_args)
        if state_dict:
            own = model.state_dict()
            has_base = hasattr(model, cls.base_model_prefix)
            pre = cls.base_model_prefix + "."
            remapped = {}
            for k, v in state_dict.items():
                if k in own:
                    remapped[k] = v
                elif has_base and (pre + k) in own:          # plain-BERT keys into a model that wraps `bert`
                    remapped[pre + k] = v
                elif not has_base and k.startswith(pre) and k[len(pre):] in own:   # pretrain ckpt into NavCMT
                    remapped[k[len(pre):]] = v
                else:
                    remapped[k] = v
            res = model.load_state_dict(remapped, strict=False)
            if res.missing_keys:
                logger.info("Weights of %s not initialized from pretrained model: %s", cls.__name__, res.missing_keys)
            if res.unexpected_keys:
                logger.info("Weights from pretrained model not used in %s: %s", cls.__name__, res.unexpected_keys)
        model.tie_weights()
        model.eval()
        return model


def precision_of(config) -> str:
    return getattr(config, "hamt_precision", "bf16")
