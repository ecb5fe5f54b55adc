"""Mirror of ``finetune_src/models/vlnbert_init.py:13-70``: build the finetune NavCMT from agent args and a
pre-training checkpoint (`module.` prefix stripped, `next_action.*` renamed `bert.next_action.*` so that the
HF prefix rule drops the `bert.` again)."""
import torch

from ..modeling import HamtConfig


def get_vlnbert_models(args, config=None):
    from .vilmodel_cmt import NavCMT
    new_ckpt_weights = {}
    if getattr(args, "bert_ckpt_file", None) is not None:
        for k, v in torch.load(args.bert_ckpt_file, map_location="cpu").items():
            if k.startswith('module'):
                new_ckpt_weights[k[7:]] = v
            else:
                if k.startswith('next_action'):
                    k = 'bert.' + k
                new_ckpt_weights[k] = v
    xlm = getattr(args, "dataset", None) == 'rxr' or getattr(args, "tokenizer", None) == 'xlm'
    # bert-base-uncased / xlm-roberta-base structural constants (no hub access needed)
    vis_config = HamtConfig(vocab_size=250002 if xlm else 30522, max_position_embeddings=514 if xlm else 512,
                            type_vocab_size=2, layer_norm_eps=1e-5 if xlm else 1e-12)
    vis_config.max_action_steps = 100
    vis_config.image_feat_size = args.image_feat_size
    vis_config.angle_feat_size = args.angle_feat_size
    vis_config.num_l_layers = args.num_l_layers
    vis_config.num_r_layers = 0
    vis_config.num_h_layers = args.num_h_layers
    vis_config.num_x_layers = args.num_x_layers
    vis_config.hist_enc_pano = args.hist_enc_pano
    vis_config.num_h_pano_layers = args.hist_pano_num_layers
    vis_config.fix_lang_embedding = args.fix_lang_embedding
    vis_config.fix_hist_embedding = args.fix_hist_embedding
    vis_config.fix_obs_embedding = args.fix_obs_embedding
    vis_config.update_lang_bert = not args.fix_lang_embedding
    vis_config.output_attentions = True
    vis_config.pred_head_dropout_prob = 0.1
    vis_config.no_lang_ca = args.no_lang_ca
    vis_config.act_pred_token = args.act_pred_token
    vis_config.hamt_precision = getattr(args, "hamt_precision", "bf16")
    return NavCMT.from_pretrained(pretrained_model_name_or_path=None, config=vis_config, state_dict=new_ckpt_weights)
