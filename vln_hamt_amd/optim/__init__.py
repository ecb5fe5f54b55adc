"""Mirror of ``pretrain_src/optim``: HF-style AdamW, warmup/linear schedule, name-based decay groups --
with the update, the global-norm clip and the bf16-shadow refresh running as HIP kernels over flat arenas."""
from .sched import warmup_linear, get_lr_sched
from .adamw import AdamW, clip_grad_norm_
from .misc import build_optimizer
