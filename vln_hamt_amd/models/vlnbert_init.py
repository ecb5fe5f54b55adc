"""Mirror of ``finetune_src/models/vlnbert_init.py:13-70``: build the finetune NavCMT from agent args and a
pre-training checkpoint.  Same rules as the reference, written as tables:

* checkpoint keys: a DataParallel ``module.`` prefix is stripped; ``next_action.*`` (a head of the pre-training wrapper) moves
  under ``bert.`` so that the HF prefix rule (base_model_prefix = "bert") drops it again and the head lands in NavCMT;
* config: structural constants of the text encoder (bert-base-uncased, or xlm-roberta-base for RxR -- no hub access needed),
  the agent's own switches copied from ``args``, and the fixed values the reference sets."""
import torch

from ..modeling import HamtConfig

# config key <- args attribute (vlnbert_init.py:42-63)
_FROM_ARGS = {
    "image_feat_size": "image_feat_size", "angle_feat_size": "angle_feat_size",
    "num_l_layers": "num_l_layers", "num_h_layers": "num_h_layers", "num_x_layers": "num_x_layers",
    "hist_enc_pano": "hist_enc_pano", "num_h_pano_layers": "hist_pano_num_layers",
    "fix_lang_embedding": "fix_lang_embedding", "fix_hist_embedding": "fix_hist_embedding", "fix_obs_embedding": "fix_obs_embedding",
    "no_lang_ca": "no_lang_ca", "act_pred_token": "act_pred_token",
}
_FIXED = {"max_action_steps": 100, "num_r_layers": 0, "output_attentions": True, "pred_head_dropout_prob": 0.1}
_TEXT_ENCODER = {False: dict(vocab_size=30522, max_position_embeddings=512, layer_norm_eps=1e-12),     # bert-base-uncased
                 True: dict(vocab_size=250002, max_position_embeddings=514, layer_norm_eps=1e-5)}      # xlm-roberta-base


def _remap_key(key: str) -> str:
    if key.startswith("module"):
        return key[len("module."):]
    return "bert." + key if key.startswith("next_action") else key


def get_vlnbert_models(args, config=None):
    from .vilmodel_cmt import NavCMT
    path = getattr(args, "bert_ckpt_file", None)
    weights = {_remap_key(k): v for k, v in torch.load(path, map_location="cpu").items()} if path is not None else {}
    multilingual = getattr(args, "dataset", None) == "rxr" or getattr(args, "tokenizer", None) == "xlm"
    cfg = HamtConfig(type_vocab_size=2, **_TEXT_ENCODER[multilingual])
    for key, attr in _FROM_ARGS.items():
        setattr(cfg, key, getattr(args, attr))
    for key, value in _FIXED.items():
        setattr(cfg, key, value)
    cfg.update_lang_bert = not args.fix_lang_embedding
    cfg.hamt_precision = getattr(args, "hamt_precision", "bf16")
    return NavCMT.from_pretrained(pretrained_model_name_or_path=None, config=cfg, state_dict=weights)
