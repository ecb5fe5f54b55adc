"""Import shim for running the *real* reference (``/root/reference``) in this container.

TEST INFRASTRUCTURE.  Used only by ``oracle/gen_goldens.py`` (never on the GPU box, never by the
product).  The reference pins transformers==4.12.3; this image ships 5.x, where
``transformers.modeling_utils.get_parameter_device`` is gone and ``BertPreTrainedModel.init_weights``
needs ``post_init`` bookkeeping.  We install a minimal 4.12-equivalent base class *before* importing
the reference modules.  Goldens never depend on the shim's random init: every parameter is
overwritten from the numpy recipe with ``load_state_dict(strict=True)``.
"""
import importlib
import importlib.util
import os
import sys
import types

import torch
from torch import nn

REF = os.environ.get("HAMT_REFERENCE", "/root/reference")


class _MiniBertPreTrainedModel(nn.Module):
    base_model_prefix = "bert"

    def __init__(self, config, *a, **kw):
        super().__init__()
        self.config = config

    def _init_weights(self, module):
        std = getattr(self.config, "initializer_range", 0.02)
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.Embedding):
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


def install():
    sys.dont_write_bytecode = True
    import transformers
    import transformers.modeling_utils as mu
    if not hasattr(mu, "get_parameter_device"):
        mu.get_parameter_device = lambda m: next(m.parameters()).device
    transformers.BertPreTrainedModel = _MiniBertPreTrainedModel


def _import_pkg(src_dir: str, pkg: str, mods):
    """Import <src_dir>/<pkg>/<mod>.py as package `pkg` without permanently touching sys.path."""
    install()
    for k in [k for k in sys.modules if k == pkg or k.startswith(pkg + ".")]:
        del sys.modules[k]
    sys.path.insert(0, os.path.join(REF, src_dir))
    try:
        out = [importlib.import_module(f"{pkg}.{m}") for m in mods]
    finally:
        sys.path.pop(0)
    return out


def import_pretrain():
    """-> (vilmodel, pretrain_cmt) modules of pretrain_src/model."""
    return _import_pkg("pretrain_src", "model", ["vilmodel", "pretrain_cmt"])


def import_finetune():
    """-> vilmodel_cmt module of finetune_src/models."""
    return _import_pkg("finetune_src", "models", ["vilmodel_cmt"])[0]


def import_finetune_agent_models():
    """-> (vilmodel_cmt, model_HAMT) of finetune_src/models.  model_HAMT.py imports `utils.misc` (pure torch) and
    `models.vlnbert_init` (whose hub / HF-loader calls sit inside functions we never call: VLNBertCMT is built around an
    already constructed NavCMT, see gen_goldens.gen_finetune)."""
    for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        del sys.modules[k]
    return _import_pkg("finetune_src", "models", ["vilmodel_cmt", "model_HAMT"])


class cuda_is_identity:
    """The agent-side wrappers call `.cuda()` on freshly built index tensors (model_HAMT.py:40, utils/misc.py:15-16); in this
    GPU-less container that call is made the identity for the duration of a forward (arithmetic untouched)."""

    def __enter__(self):
        self._orig = torch.Tensor.cuda
        torch.Tensor.cuda = lambda t, *a, **k: t
        return self

    def __exit__(self, *a):
        torch.Tensor.cuda = self._orig


def make_config(ocfg, **extra):
    """A PretrainedConfig-like namespace carrying r2r_model_config.json's keys."""
    d = dict(vars(ocfg))
    d["pretrain_tasks"] = set(ocfg.pretrain_tasks)
    d.update(hidden_act="gelu", output_attentions=False, output_hidden_states=False,
             initializer_range=0.02, num_hidden_layers=12)
    d.update(extra)
    return types.SimpleNamespace(**d)


def import_vit():
    """-> the reference's pretrain_src/model/vision_transformer.py module.  It is a timm copy that imports a handful of
    timm helpers at module level (vision_transformer.py:30-33); timm is not in the image, so tiny stand-ins for exactly
    those names are installed first (only `to_2tuple`, `trunc_normal_`, `DropPath` are ever *called* by the classes we
    instantiate; goldens never depend on random init -- every parameter is overwritten from the numpy recipe)."""
    install()
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        data = types.ModuleType("timm.data")
        data.IMAGENET_DEFAULT_MEAN, data.IMAGENET_DEFAULT_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
        models = types.ModuleType("timm.models")
        helpers = types.ModuleType("timm.models.helpers")
        helpers.build_model_with_cfg = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("not available in the shim"))
        helpers.overlay_external_default_cfg = lambda cfg, kw: None
        layers = types.ModuleType("timm.models.layers")

        class DropPath(nn.Module):          # only constructed for drop_path > 0 (vision_transformer.py:190)
            def __init__(self, p=0.0):
                super().__init__()
                self.p = p

            def forward(self, x):
                assert self.p == 0.0 or not self.training
                return x

        layers.DropPath = DropPath
        layers.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
        layers.trunc_normal_ = lambda t, std=1.0, **k: nn.init.trunc_normal_(t, std=std)
        layers.lecun_normal_ = lambda t: nn.init.normal_(t, std=0.02)
        registry = types.ModuleType("timm.models.registry")
        registry.register_model = lambda f: f
        for name, mod in (("timm", timm), ("timm.data", data), ("timm.models", models), ("timm.models.helpers", helpers),
                          ("timm.models.layers", layers), ("timm.models.registry", registry)):
            sys.modules[name] = mod
    spec = importlib.util.spec_from_file_location("ref_vision_transformer", os.path.join(REF, "pretrain_src", "model", "vision_transformer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def import_collate():
    """-> the reference's pretrain_src/data/r2r_tasks.py module (the six *_collate functions), loaded under a synthetic
    package so that data/__init__.py -- which imports the jsonlines / h5py readers this image lacks -- is not executed."""
    sys.dont_write_bytecode = True
    d = os.path.join(REF, "pretrain_src", "data")
    pkg = types.ModuleType("_ref_data")
    pkg.__path__ = [d]
    sys.modules["_ref_data"] = pkg
    out = None
    for name in ("common", "r2r_tasks"):
        spec = importlib.util.spec_from_file_location(f"_ref_data.{name}", os.path.join(d, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
        out = mod
    return out


def import_loader():
    """-> the reference's pretrain_src/data/loader.py module (MetaLoader, PrefetchLoader, build_dataloader): it imports only torch."""
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("_ref_loader", os.path.join(REF, "pretrain_src", "data", "loader.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def import_r2r_data(feature_npz):
    """-> the reference's pretrain_src/data/r2r_data.py module.  It imports `jsonlines` and `h5py` at module level, neither of
    which is in this image; two I/O stand-ins are installed for exactly the calls the module makes -- `jsonlines.Reader(f)`
    (iterate the JSON objects of an open text file) and `h5py.File(path, 'r')` used as a context manager whose items support
    `[...]` -- the latter served from the numpy archive `feature_npz` (the same arrays the product's reader is given).  No
    arithmetic of the reference is replaced.  numpy >= 1.24 removed the `np.bool` alias r2r_data.py:232 still uses: restored."""
    import json
    import numpy as np
    sys.dont_write_bytecode = True
    if not hasattr(np, "bool"):
        np.bool = np.bool_
    jl = types.ModuleType("jsonlines")

    class Reader:
        def __init__(self, f):
            self.f = f

        def __iter__(self):
            for ln in self.f:
                if ln.strip():
                    yield json.loads(ln)

    jl.Reader = Reader
    h5 = types.ModuleType("h5py")
    arrays = dict(np.load(feature_npz))

    class _Item:
        def __init__(self, a):
            self.a = a

        def __getitem__(self, idx):
            return self.a[idx]

    class File:
        def __init__(self, path, mode="r"):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def __getitem__(self, key):
            return _Item(arrays[key])

    h5.File = File
    sys.modules["jsonlines"], sys.modules["h5py"] = jl, h5
    spec = importlib.util.spec_from_file_location("_ref_r2r_data", os.path.join(REF, "pretrain_src", "data", "r2r_data.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
