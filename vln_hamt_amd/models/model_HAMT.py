"""Mirror of ``finetune_src/models/model_HAMT.py``: `VLNBertCMT` (:11-65) and `Critic` (:258-269).
(The reference's VLNBertCausalCMT / VLNBertMMT / VLNBertCMT3 pass kwargs NavCMT.forward does not accept -- dead
code, not mirrored.)"""
import torch
import torch.nn as nn

from .. import ops
from .vlnbert_init import get_vlnbert_models


def length2mask(length, size=None, device=None):
    """finetune_src/utils/misc.py:12-17 (True = padding)."""
    size = int(max(length)) if size is None else size
    ar = torch.arange(size, dtype=torch.int64, device=device)[None].repeat(len(length), 1)
    return ar > (torch.as_tensor(length, dtype=torch.int64, device=device) - 1)[:, None]


class HistoryCache:
    """Device-resident history of a no-grad rollout.  The reference agent keeps a Python list of per-step
    embeddings and re-stacks ALL of them at every step (agent_cmt.py:305-307, model_HAMT.py:48); here each new embedding is
    written once into its row of a [B, max_len, H] buffer and `visual` reads the first n rows in place -- the buffer's address
    never changes, so hipGraph-captured `visual` steps (one per history length) read it directly and the captured `history` step
    can write into it.  Pass it as `hist_embeds`; training rollouts (autograd) keep using the list."""

    def __init__(self, batch_size: int, max_len: int, hidden: int, device):
        self.buf = torch.zeros(batch_size, max_len, hidden, dtype=torch.float32, device=device)
        self.n = 0

    def reset(self, cls_embed=None):
        self.n = 0
        if cls_embed is not None:
            self.append(cls_embed.expand(self.buf.shape[0], -1))
        return self

    def append(self, h):
        """h: [B, H] (the `history` mode's output of this step)"""
        B, T, H = self.buf.shape
        assert self.n < T, "HistoryCache is full"
        ops.copy_rows_into(self.buf.view(B, T * H), self.n * H, h.detach().reshape(B, H).contiguous())
        self.n += 1
        return self

    def view(self):
        return self.buf[:, :self.n]

    def __len__(self):
        return self.n


class VLNBertCMT(nn.Module):
    _hamt_container = True      # (optim.AdamW.attach) reads parameters only through self.vln_bert's __call__

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.vln_bert = get_vlnbert_models(args, config=None)
        self.drop_env = nn.Dropout(p=args.feat_dropout)

    def _drop(self, x):
        return ops.dropout(x, float(self.drop_env.p), self.training)

    def forward(self, mode, txt_ids=None, txt_masks=None, txt_embeds=None, hist_img_feats=None, hist_ang_feats=None,
                hist_pano_img_feats=None, hist_pano_ang_feats=None, hist_embeds=None, hist_lens=None, ob_step=None,
                ob_img_feats=None, ob_ang_feats=None, ob_nav_types=None, ob_masks=None, return_states=False):
        if mode == 'language':
            return self.vln_bert(mode, txt_ids=txt_ids, txt_masks=txt_masks)
        if mode == 'history':
            if hist_img_feats is not None:
                hist_img_feats = self._drop(hist_img_feats)
            if hist_pano_img_feats is not None:
                hist_pano_img_feats = self._drop(hist_pano_img_feats)
            dev = next(self.parameters()).device
            ob_step_ids = torch.tensor([ob_step], dtype=torch.long, device=dev) if ob_step is not None else None
            return self.vln_bert(mode, hist_img_feats=hist_img_feats, hist_ang_feats=hist_ang_feats, ob_step_ids=ob_step_ids,
                                 hist_pano_img_feats=hist_pano_img_feats, hist_pano_ang_feats=hist_pano_ang_feats)
        if mode == 'visual':
            hist_embeds = hist_embeds.view() if isinstance(hist_embeds, HistoryCache) else torch.stack(hist_embeds, 1)
            hist_masks = length2mask(hist_lens, size=hist_embeds.size(1), device=hist_embeds.device).logical_not()
            ob_img_feats = self._drop(ob_img_feats)
            act_logits, txt_embeds, hist_embeds, ob_embeds = self.vln_bert(
                mode, txt_embeds=txt_embeds, txt_masks=txt_masks, hist_embeds=hist_embeds, hist_masks=hist_masks,
                ob_img_feats=ob_img_feats, ob_ang_feats=ob_ang_feats, ob_nav_types=ob_nav_types, ob_masks=ob_masks)
            if return_states:
                if self.args.no_lang_ca:
                    states = hist_embeds[:, 0]
                else:
                    states = ops.mul_bcast(txt_embeds[:, :1].contiguous(), hist_embeds[:, 0]).squeeze(1)   # [CLS] product
                return act_logits, states
            return (act_logits,)
        raise ValueError(mode)


class Critic(nn.Module):
    """state-value head 768 -> 512 -> 1 (model_HAMT.py:258-269)."""

    def __init__(self, args):
        super().__init__()
        self.state2value = nn.Sequential(nn.Linear(768, 512), nn.ReLU(), nn.Dropout(args.dropout), nn.Linear(512, 1))
        self.prec = getattr(args, "hamt_precision", "bf16")

    def forward(self, state):
        l0, drop, l1 = self.state2value[0], self.state2value[2], self.state2value[3]
        h = ops.linear(state, l0.weight, l0.bias, ops.ACT_RELU, self.prec)
        h = ops.dropout(h, float(drop.p), self.training)
        return ops.linear(h, l1.weight, l1.bias, ops.ACT_NONE, self.prec).squeeze()
