"""R2R pretraining data: trajectory (jsonl) and view-feature (HDF5) readers and the per-sample input builder, behind the
reference's names (pretrain_src/data/r2r_data.py: `MultiStepNavData` :98-345, `angle_feature` :14-17, the 36-view angle tables
:19-54, `load_nav_graphs` :56-89, `softmax` :91-94).

What differs from the reference in HOW, not in WHAT (outputs are pinned bit for bit by tests/golden/r2r_data.npz, produced by the
reference's own class on the committed tiny dataset tests/golden/r2r_tiny/):

* the view-feature file is opened ONCE per process (the reference re-opens the HDF5 file for every viewpoint it reads,
  r2r_data.py:322) behind a small store interface -- HDF5 through h5py when it is installed, `.npz` archives or a directory of
  `.npy` files (memory mapped) otherwise -- and a viewpoint's [36, feat + prob] block is cast to float32 once and cached;
* the 36 x 36 angle tables are built once as [36, 36, size] arrays and history features are gathered with array indexing instead
  of per-step list appends;
* shortest-path distances (progress targets) come from an own binary-heap Dijkstra over the connectivity files: no networkx.
"""
from __future__ import annotations

import heapq
import json
import math
import os
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np


# ------------------------------------------------------------------------------------------------ angle features
def angle_feature(heading, elevation, angle_feat_size):
    """[sin h, cos h, sin e, cos e] tiled to `angle_feat_size` (r2r_data.py:14-17)"""
    quad = (math.sin(heading), math.cos(heading), math.sin(elevation), math.cos(elevation))
    return np.array(quad * (angle_feat_size // 4), dtype=np.float32)


def _view_angles():
    """(heading, elevation) of the 36 discretised views as the reference accumulates them (r2r_data.py:23-31): 3 elevation rows
    from -30 deg, 12 headings of 30 deg each -- running sums in double precision, not k * 30 deg, so that every bit agrees."""
    step = math.radians(30)
    out, heading, elevation = [], 0.0, math.radians(-30)
    for ix in range(36):
        if ix == 0:
            heading, elevation = 0, math.radians(-30)
        elif ix % 12 == 0:
            heading = 0
            elevation += step
        else:
            heading += step
        out.append((heading, elevation))
    return out


def get_point_angle_feature(angle_feat_size, baseViewId=0):
    """[36, size]: angle features of the 36 views relative to the heading of view `baseViewId` (r2r_data.py:19-32)"""
    base = (baseViewId % 12) * math.radians(30)
    return np.stack([angle_feature(h - base, e, angle_feat_size) for h, e in _view_angles()], 0)


def get_all_point_angle_feature(angle_feat_size):
    return [get_point_angle_feature(angle_feat_size, b) for b in range(36)]


def get_point_rel_angles(baseViewId=0):
    """[36, 2]: (heading relative to the base view, absolute elevation) per view (r2r_data.py:37-51)"""
    base = (baseViewId % 12) * math.radians(30)
    rel = np.zeros((36, 2), np.float32)
    for ix, (h, e) in enumerate(_view_angles()):
        rel[ix, 0] = h - base
        rel[ix, 1] = e
    return rel


def get_all_point_rel_angles():
    return [get_point_rel_angles(b) for b in range(36)]


def softmax(logits, dim=1):
    """(n, d) logits -> probabilities, un-shifted exponentials as the reference computes them (r2r_data.py:91-94)"""
    e = np.exp(logits)
    return e / np.sum(e, axis=dim, keepdims=True)


# ------------------------------------------------------------------------------------------------ navigation graphs
def load_nav_graphs(connectivity_dir):
    """-> (graphs, shortest_distances) per scan (r2r_data.py:56-89).  graphs[scan] = {viewpoint: {neighbour: metres}} over the
    `included` viewpoints joined by `unobstructed` links; shortest_distances[scan][a][b] = length of the shortest path."""
    with open(os.path.join(connectivity_dir, "scans.txt")) as f:
        scans = [ln.strip() for ln in f if ln.strip()]
    graphs, dists = {}, {}
    for scan in scans:
        with open(os.path.join(connectivity_dir, f"{scan}_connectivity.json")) as f:
            nodes = json.load(f)
        xyz = [(n["pose"][3], n["pose"][7], n["pose"][11]) for n in nodes]
        adj: Dict[str, Dict[str, float]] = {}
        for i, a in enumerate(nodes):
            if not a["included"]:
                continue
            for j, linked in enumerate(a["unobstructed"]):
                if linked and nodes[j]["included"]:
                    assert nodes[j]["unobstructed"][i], "Graph should be undirected"
                    w = ((xyz[i][0] - xyz[j][0]) ** 2 + (xyz[i][1] - xyz[j][1]) ** 2 + (xyz[i][2] - xyz[j][2]) ** 2) ** 0.5
                    adj.setdefault(a["image_id"], {})[nodes[j]["image_id"]] = w
                    adj.setdefault(nodes[j]["image_id"], {})[a["image_id"]] = w
        graphs[scan] = adj
        dists[scan] = {src: _dijkstra(adj, src) for src in adj}
    return graphs, dists


def _dijkstra(adj, src):
    dist = {src: 0}
    heap = [(0, 0, src)]
    tick = 1
    done = set()
    while heap:
        d, _, u = heapq.heappop(heap)
        if u in done:
            continue
        done.add(u)
        for v, w in adj[u].items():
            nd = d + w
            if v not in dist or nd < dist[v]:
                dist[v] = nd
                heapq.heappush(heap, (nd, tick, v))
                tick += 1
    return dist


# ------------------------------------------------------------------------------------------------ file readers
def read_jsonl(path) -> Iterator[dict]:
    """One JSON object per non-empty line (what `jsonlines.Reader` yields at r2r_data.py:129)."""
    with open(path, "r") as f:
        for ln in f:
            ln = ln.strip()
            if ln:
                yield json.loads(ln)


class ViewFeatureStore:
    """`"{scan}_{viewpoint}"` -> float32 [36, image_feat + image_prob] (precompute_img_features_vit.py:148-159 writes them;
    r2r_data.py:317-329 reads them).  Backends by file type: `.hdf5` / `.h5` (h5py, one handle per process, opened lazily so that
    DataLoader workers each open their own), `.npz` (numpy archive, one member per key), a directory of `<key>.npy` (memory
    mapped).  `in_memory`: keep every block that was read (r2r_data.py:113-115)."""

    def __init__(self, path: str, in_memory: bool = False):
        self.path, self.in_memory = path, in_memory
        self._cache: Dict[str, np.ndarray] = {}
        self._h = None
        if os.path.isdir(path):
            self.kind = "npy_dir"
        elif path.endswith(".npz"):
            self.kind = "npz"
        else:
            self.kind = "hdf5"

    def _handle(self):
        if self._h is None:
            if self.kind == "hdf5":
                try:
                    import h5py
                except ImportError as e:        # loud: there is no silent substitute for a missing reader
                    raise ImportError(f"ViewFeatureStore: '{self.path}' is an HDF5 file and h5py is not installed; convert it with "
                                      "tools/h5_to_npz.py on a machine that has h5py, or install h5py") from e
                self._h = h5py.File(self.path, "r")
            elif self.kind == "npz":
                self._h = np.load(self.path, mmap_mode="r")
            else:
                self._h = self.path
        return self._h

    def __getstate__(self):            # DataLoader workers: never share a file handle across processes
        st = dict(self.__dict__)
        st["_h"] = None
        return st

    def __contains__(self, key):
        h = self._handle()
        return os.path.exists(os.path.join(h, key + ".npy")) if self.kind == "npy_dir" else key in h

    def get(self, key: str) -> np.ndarray:
        fts = self._cache.get(key)
        if fts is None:
            h = self._handle()
            if self.kind == "hdf5":
                fts = h[key][...].astype(np.float32)
            elif self.kind == "npz":
                fts = np.asarray(h[key]).astype(np.float32)
            else:
                fts = np.load(os.path.join(h, key + ".npy"), mmap_mode="r").astype(np.float32)
            if self.in_memory:
                self._cache[key] = fts
        return fts


# ------------------------------------------------------------------------------------------------ the dataset
class MultiStepNavData(object):
    """Trajectories x instructions x time steps of R2R-style pretraining data; `get_input` builds one sample's arrays
    (instruction tokens, history view / panorama features up to the current step, the current observation, action and progress
    targets) for the task datasets (r2r_data.py:98-345)."""

    def __init__(self, traj_files, img_ft_file, scanvp_cands_file, connectivity_dir,
                 image_prob_size=1000, image_feat_size=2048, angle_feat_size=4,
                 max_txt_len=80, max_act_len=100, hist_enc_pano=True, val_sample_num=None,
                 in_memory=False, ob_cand_pano_view=False):
        self.traj_files, self.img_ft_file = traj_files, img_ft_file
        self.image_feat_size, self.image_prob_size, self.angle_feat_size = image_feat_size, image_prob_size, angle_feat_size
        self.max_txt_len = max_txt_len
        self.max_act_len = min(30, max_act_len)                 # "due to memory issue" (r2r_data.py:109)
        self.hist_enc_pano, self.ob_cand_pano_view = hist_enc_pano, ob_cand_pano_view
        self.in_memory = in_memory
        self.features = ViewFeatureStore(img_ft_file, in_memory=in_memory)
        with open(scanvp_cands_file) as f:
            self.scanvp_cands = json.load(f)
        self.graphs, self.shortest_distances = load_nav_graphs(connectivity_dir)
        self.angle_features = get_all_point_angle_feature(angle_feat_size)
        self.rel_angles = get_all_point_rel_angles()
        self._angle_table = np.stack(self.angle_features, 0)     # [36 base views, 36 views, size]

        # index: (trajectory, instruction, path length) per training sequence; (trajectory, instruction, step) per step
        self.traj_data: List[dict] = []
        self.traj_refer, self.traj_step_refer = [], []
        for path in self.traj_files:
            for item in read_jsonl(path):
                n = len(self.traj_data)
                self.traj_data.append(item)
                path_len = min(len(item["path"]), self.max_act_len - 1)
                for j in range(len(item["instr_encodings"])):
                    self.traj_refer.append((n, j, path_len))
                    self.traj_step_refer.extend((n, j, t) for t in range(path_len))
        if val_sample_num:          # validation on a random subset (r2r_data.py:138-145: two draws from numpy's global stream)
            sel = np.random.permutation(len(self.traj_refer))[:val_sample_num]
            self.traj_refer = [self.traj_refer[s] for s in sel]
            sel = np.random.permutation(len(self.traj_step_refer))[:val_sample_num]
            self.traj_step_refer = [self.traj_step_refer[s] for s in sel]

    # ---- features
    def get_image_feature(self, scan, viewpoint, pad_stop_token=False):
        fts = self.features.get(f"{scan}_{viewpoint}")
        if pad_stop_token:
            fts = np.vstack([fts, np.zeros((1, fts.shape[-1]), dtype=fts.dtype)])
        return fts

    def get_angle_feature(self, viewindex, pad_stop_token=False):
        fts = self.angle_features[viewindex]
        if pad_stop_token:
            fts = np.vstack([fts, np.zeros((1, fts.shape[-1]), dtype=fts.dtype)])
        return fts

    def get_progress(self, scan, start_vp, cur_vp, end_vp):
        if cur_vp == end_vp:
            return 1
        if start_vp == cur_vp:
            return 0
        d = self.shortest_distances[scan]
        return 1 - d[cur_vp][end_vp] / max(d[start_vp][end_vp], 0.1)

    # ---- one sample
    def get_input(self, i_path, j_instr, t_cur, return_ob=False, return_hist_img_probs=False,
                  return_ob_action=False, return_ob_progress=False, ob_cand_pano_view=None):
        td = self.traj_data[i_path]
        scan = td["scan"]
        path = td["path"][:self.max_act_len - 1]
        views, act_views, rel_act = td["path_viewindex"], td["action_viewindex"], td["rel_act_angles"]
        hist = self.get_history_feature(scan, path, views, rel_act, t_cur, return_img_probs=return_hist_img_probs)
        outs = {"instr_id": td["instr_ids"][j_instr], "instr_encoding": td["instr_encodings"][j_instr][:self.max_txt_len],
                "hist_img_fts": hist[0], "hist_ang_fts": hist[1], "hist_lens": t_cur}
        if self.hist_enc_pano:
            outs["hist_pano_img_fts"], outs["hist_pano_ang_fts"] = hist[2], hist[3]
        if return_hist_img_probs:
            outs["hist_img_probs"] = hist[4]
        if return_ob:
            cand_view = self.ob_cand_pano_view if ob_cand_pano_view is None else ob_cand_pano_view
            build = self.get_ob_cand_pano_view if cand_view else self.get_ob_pano_view
            img, ang, nav, gt_label, gt_angle = build(scan, path, views, act_views, rel_act, t_cur)
            outs.update(ob_img_fts=img, ob_ang_fts=ang, ob_nav_types=nav)
            if return_ob_action:
                outs["ob_action_viewindex"], outs["ob_action_angles"] = gt_label, gt_angle
            if return_ob_progress:
                goal = td["guide_path"][-1] if "guide_path" in td else path[-1]
                outs["ob_progress"] = self.get_progress(scan, path[0], path[t_cur], goal)
        return outs

    def get_ob_pano_view(self, scan, path, path_viewindex, action_viewindex, rel_act_angles, t_cur):
        """the 36 views in their own order + a zero STOP row; nav type 1 at the views that lead to a neighbour, 2 at STOP (:203-221)"""
        img = self.get_image_feature(scan, path[t_cur], pad_stop_token=True)[:, :self.image_feat_size]
        ang = self.get_angle_feature(path_viewindex[t_cur], pad_stop_token=True)
        nav = np.zeros((img.shape[0],), dtype=np.int64)
        nav[-1] = 2
        cands = self.scanvp_cands[f"{scan}_{path[t_cur]}"]
        nav[np.array([v[0] for v in cands.values()])] = 1
        if action_viewindex[t_cur] != -1:
            return img, ang, nav, action_viewindex[t_cur], rel_act_angles[t_cur]
        return img, ang, nav, img.shape[0] - 1, np.zeros((2,), dtype=np.float32)            # stop

    def get_ob_cand_pano_view(self, scan, path, path_viewindex, action_viewindex, rel_act_angles, t_cur):
        """candidate views first (their own relative angles), then STOP, then the remaining views (:223-266)"""
        img36 = self.get_image_feature(scan, path[t_cur], pad_stop_token=False)[:, :self.image_feat_size]
        ang36 = self.get_angle_feature(path_viewindex[t_cur], pad_stop_token=False)
        cands = self.scanvp_cands[f"{scan}_{path[t_cur]}"]
        rel = self.rel_angles[path_viewindex[t_cur]]
        nxt = path[t_cur + 1] if t_cur < len(path) - 1 else None
        cand_views = [v[0] for v in cands.values()]
        gt_label = None
        for k, vp in enumerate(cands):
            if nxt is not None and vp == nxt:
                gt_label = k                              # (the last match wins, as in the reference's loop)
        cand_img = img36[cand_views]
        cand_ang = np.stack([angle_feature(rel[v[0]][0] + v[2], rel[v[0]][1] + v[3], self.angle_feat_size) for v in cands.values()], 0)
        rest = np.ones((36,), dtype=bool)
        rest[cand_views] = False
        nav = np.array([1] * len(cand_views) + [2] + [0] * int(rest.sum()))
        img = np.concatenate([cand_img, np.zeros((1, self.image_feat_size), dtype=np.float32), img36[rest]], 0)
        ang = np.concatenate([cand_ang, np.zeros((1, self.angle_feat_size), dtype=np.float32), ang36[rest]], 0)
        if gt_label is None:
            return img, ang, nav, len(cand_views), np.zeros((2,), dtype=np.float32)           # stop
        return img, ang, nav, gt_label, rel_act_angles[t_cur]

    def get_history_feature(self, scan, path, path_viewindex, rel_act_angles, t_cur, return_img_probs=False):
        """features of the steps BEFORE t_cur: the view taken, its action angle (zeros for the STOP step), the whole panorama
        with the angle table of that view, optionally the soft labels of the view taken (:269-315)"""
        F, A, P = self.image_feat_size, self.angle_feat_size, self.image_prob_size
        if t_cur <= 0:
            pano = (np.zeros((0, 36, F), np.float32), np.zeros((0, 36, A), np.float32)) if self.hist_enc_pano else ([], [])
            out = (np.zeros((0, F), np.float32), np.zeros((0, A), np.float32)) + pano
            return out + (np.zeros((0, P), np.float32),) if return_img_probs else out
        vidx = np.asarray(path_viewindex[:t_cur])
        blocks = np.stack([self.get_image_feature(scan, path[t]) for t in range(t_cur)], 0)           # [t, 36, F + P]
        steps = np.arange(t_cur)
        img = blocks[steps, vidx, :F]
        ang = np.stack([np.zeros((A,), np.float32) if t == len(path) - 1 else angle_feature(rel_act_angles[t][0], rel_act_angles[t][1], A)
                        for t in range(t_cur)])
        pano_img = pano_ang = []
        if self.hist_enc_pano:
            pano_img, pano_ang = blocks[:, :, :F], self._angle_table[vidx]
        if return_img_probs:
            return img, ang, pano_img, pano_ang, softmax(blocks[steps, vidx, F:])
        return img, ang, pano_img, pano_ang
