"""``pretrain_src/utils/misc.py:52-65`` -- `wrap_model(model, device, local_rank)` -- for the MI355X path: same name, same
signature, same place in main_r2r.py; returns `parallel.ArenaDataParallel` (the flat-arena RCCL exchange behind the deferred
weight-gradient launches) where the reference returns torch's DistributedDataParallel.  `set_dropout` of the same file works on our
modules unchanged (real nn.Dropout children whose `.p` is read at call time), so it is not restated here."""
from ..parallel import ArenaDataParallel, wrap_model  # noqa: F401
