"""The six proxy-task datasets of R2R pretraining behind the reference's names (pretrain_src/data/r2r_tasks.py: `MlmDataset` :55-93,
`MrcDataset` :155-200, `ItmDataset` :230-266, `SapDataset` :290-341, `SarDataset` :383-442, `SprelDataset` :486-557, `random_word`
:12-53).  Each `__getitem__` turns one `MultiStepNavData.get_input` sample into the dict of tensors its `*_collate` function
(data/collate.py) takes, applying the task's random corruption on the host.

One base class carries what the reference repeats six times (instruction, history, observation tensors, the random view / angle
"kill"); the subclasses only say which sample they draw and what they add.  The host RNG streams are consumed in the reference's
order -- python `random` for word masking and the kills, numpy's global stream for region masks and the SPREL anchor -- so the same
seeds give the same samples: pinned by tests/golden/r2r_tasks.npz (the reference's own classes on tests/golden/r2r_tiny/).
"""
from __future__ import annotations

import math
import random

import numpy as np
import torch
from torch.utils.data import Dataset


def random_word(tokens, vocab_range, mask):
    """BERT word masking (r2r_tasks.py:12-53): each token is selected with probability 0.15; a selected token becomes `mask` (80 %), a
    uniformly drawn vocabulary id (10 %) or stays (10 %); its label is the original id, every other label -1; at least one token
    (the first) is masked.  Draws: one `random.random()` per token, one `random.choice` per replaced token."""
    out, labels = [], []
    vocab = range(*vocab_range)
    for tok in tokens:
        u = random.random()
        if u >= 0.15:
            out.append(tok)
            labels.append(-1)
            continue
        u /= 0.15
        out.append(mask if u < 0.8 else (random.choice(vocab) if u < 0.9 else tok))
        labels.append(tok)
    if all(l == -1 for l in labels):
        labels[0], out[0] = tokens[0], mask
    return out, labels


def _standardize_radians(x):
    """angles into [-pi, pi) (r2r_tasks.py:438-442)"""
    x = np.mod(x, 2 * np.pi)
    return np.where(x >= np.pi, x - 2 * np.pi, x)


class _NavTaskDataset(Dataset):
    """`per_step`: one item per (trajectory, instruction, step) -- the observation tasks -- instead of one per (trajectory,
    instruction) with the whole path as history."""
    per_step = False
    input_flags: dict = {}

    def __init__(self, nav_db, tok):
        self.nav_db, self.tok = nav_db, tok
        self.cls_token_id, self.sep_token_id, self.pad_token_id = tok.cls_token_id, tok.sep_token_id, tok.pad_token_id

    def __len__(self):
        return len(self.nav_db.traj_step_refer if self.per_step else self.nav_db.traj_refer)

    def _inputs(self, i):
        ref = (self.nav_db.traj_step_refer if self.per_step else self.nav_db.traj_refer)[i]
        return self.nav_db.get_input(*ref, **self.input_flags)

    @staticmethod
    def _text(inputs, out):
        out["txt_ids"] = torch.LongTensor(inputs["instr_encoding"])
        out["txt_lens"] = out["txt_ids"].size(0)

    @staticmethod
    def _history(inputs, out):
        out["hist_img_fts"] = torch.from_numpy(inputs["hist_img_fts"])
        out["hist_ang_fts"] = torch.from_numpy(inputs["hist_ang_fts"])
        if "hist_pano_img_fts" in inputs:
            out["hist_pano_img_fts"] = torch.from_numpy(inputs["hist_pano_img_fts"])
            out["hist_pano_ang_fts"] = torch.from_numpy(inputs["hist_pano_ang_fts"])
        out["hist_lens"] = inputs["hist_lens"]

    def _observation(self, inputs, out):
        """current panorama; with probability random_kill_v the view features are zeroed, else with random_kill_a the angles"""
        out["ob_img_fts"] = torch.from_numpy(inputs["ob_img_fts"])
        seen = True
        if random.random() < self.random_kill_v:
            out["ob_img_fts"][...] = 0
            seen = False
        out["ob_ang_fts"] = torch.from_numpy(inputs["ob_ang_fts"])
        if seen and random.random() < self.random_kill_a:
            out["ob_ang_fts"][...] = 0
        out["ob_nav_types"] = torch.LongTensor(inputs["ob_nav_types"])
        out["ob_lens"] = out["ob_img_fts"].size(0)


class MlmDataset(_NavTaskDataset):
    """masked language modelling over the instruction, the whole path as history"""
    input_flags = dict(return_ob=False, return_ob_action=False, return_hist_img_probs=False, return_ob_progress=False)

    def __init__(self, nav_db, tok):
        super().__init__(nav_db, tok)
        self.vocab_range = [1996, 29611]            # bert-base-uncased word pieces (r2r_tasks.py:60)
        self.mask_token_id = tok.mask_token_id

    def __getitem__(self, i):
        inputs, out = self._inputs(i), {}
        ids, labels = random_word(inputs["instr_encoding"], self.vocab_range, self.mask_token_id)
        out["txt_ids"], out["txt_labels"] = torch.LongTensor(ids), torch.LongTensor(labels)
        out["txt_lens"] = out["txt_ids"].size(0)
        self._history(inputs, out)
        return out


class MrcDataset(_NavTaskDataset):
    """masked region classification: history views zeroed with probability `mask_prob` (at least one), their soft labels kept"""
    input_flags = dict(return_ob=False, return_ob_action=False, return_hist_img_probs=True, return_ob_progress=False)

    def __init__(self, nav_db, tok, mask_prob):
        super().__init__(nav_db, tok)
        self.mask_prob = mask_prob

    def __getitem__(self, i):
        inputs, out = self._inputs(i), {}
        self._text(inputs, out)
        n = inputs["hist_img_probs"].shape[0]
        picked = [np.random.rand() < self.mask_prob for _ in range(n)]
        if not any(picked):
            picked[np.random.randint(n)] = True
        masks = torch.tensor(picked)
        out["hist_img_fts"] = torch.from_numpy(inputs["hist_img_fts"]).masked_fill(masks[:, None], 0)
        if "hist_pano_img_fts" in inputs:
            out["hist_pano_img_fts"] = torch.from_numpy(inputs["hist_pano_img_fts"]).masked_fill(masks[:, None, None], 0)
        out["hist_img_probs"] = torch.from_numpy(inputs["hist_img_probs"])
        out["hist_mrc_masks"] = masks
        out["hist_ang_fts"] = torch.from_numpy(inputs["hist_ang_fts"])
        if "hist_pano_ang_fts" in inputs:
            out["hist_pano_ang_fts"] = torch.from_numpy(inputs["hist_pano_ang_fts"])
        out["hist_lens"] = inputs["hist_lens"]
        return out


class ItmDataset(_NavTaskDataset):
    """instruction-trajectory matching: the positive pair; negatives are drawn inside the model (vilmodel.py:681-704)"""
    input_flags = dict(return_ob=False, return_ob_action=False, return_hist_img_probs=False, return_ob_progress=False)

    def __getitem__(self, i):
        inputs, out = self._inputs(i), {}
        self._text(inputs, out)
        self._history(inputs, out)
        return out


class SapDataset(_NavTaskDataset):
    """single-step action prediction: which view to take next"""
    per_step = True
    input_flags = dict(return_ob=True, return_ob_action=True, return_hist_img_probs=False, return_ob_progress=False)

    def __init__(self, nav_db, tok, random_kill_v, random_kill_a):
        super().__init__(nav_db, tok)
        self.random_kill_v, self.random_kill_a = random_kill_v, random_kill_a

    def _labels(self, inputs, out):
        out["ob_action_viewindex"] = inputs["ob_action_viewindex"]

    def __getitem__(self, i):
        inputs, out = self._inputs(i), {}
        self._text(inputs, out)
        self._observation(inputs, out)
        self._labels(inputs, out)
        self._history(inputs, out)
        return out


class SarDataset(SapDataset):
    """single-step action regression: heading / elevation of the action and the progress towards the goal"""
    input_flags = dict(return_ob=True, return_ob_action=True, return_hist_img_probs=False, return_ob_progress=True)
    _standardize_radians = staticmethod(_standardize_radians)

    def _labels(self, inputs, out):
        out["ob_action_angles"] = _standardize_radians(inputs["ob_action_angles"])
        out["ob_progress"] = inputs["ob_progress"]


class SprelDataset(SapDataset):
    """spatial relation regression: relative heading / elevation of every view to a random anchor view (always the 36 + STOP layout)"""
    input_flags = dict(return_ob=True, return_ob_action=False, return_hist_img_probs=False, return_ob_progress=False, ob_cand_pano_view=False)
    _standardize_radians = staticmethod(_standardize_radians)

    def __init__(self, nav_db, tok, random_kill_v, random_kill_a):
        super().__init__(nav_db, tok, random_kill_v, random_kill_a)
        step = math.radians(30)
        head = np.array([(v % 12) * step for v in range(36)])
        elev = np.array([(v // 12 - 1) * step for v in range(36)])
        # sp_targets[anchor, view] = (heading, elevation) of `view` relative to `anchor`, wrapped into [-pi, pi)
        self.sp_targets = _standardize_radians(np.stack([head[None, :] - head[:, None], elev[None, :] - elev[:, None]], -1))

    def _labels(self, inputs, out):
        pass

    def __getitem__(self, i):
        out = super().__getitem__(i)
        anchor = np.random.randint(36)
        out["sp_anchor_idxs"] = anchor
        out["sp_targets"] = self.sp_targets[anchor]
        return out


from .collate import itm_collate, mlm_collate, mrc_collate, sap_collate, sar_collate, sprel_collate  # noqa: E402,F401  (the reference keeps them here)
