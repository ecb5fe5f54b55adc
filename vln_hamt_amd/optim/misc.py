"""Optimizer factory with the reference's name-based weight-decay groups (pretrain_src/optim/misc.py:12-37):
a parameter is exempt from decay iff its name contains 'bias', 'LayerNorm.bias' or 'LayerNorm.weight' -- so the
lower-case `layer_norm.weight` / `net.2.weight` LayerNorm gains DO decay, exactly as in the reference."""
from .adamw import AdamW

NO_DECAY = ('bias', 'LayerNorm.bias', 'LayerNorm.weight')


def build_optimizer(model, opts):
    named = list(model.named_parameters())
    groups = [
        {'params': [p for n, p in named if not any(nd in n for nd in NO_DECAY)], 'weight_decay': opts.weight_decay},
        {'params': [p for n, p in named if any(nd in n for nd in NO_DECAY)], 'weight_decay': 0.0},
    ]
    if opts.optim != 'adamw':
        raise ValueError('invalid optimizer (the HIP path implements the reference default, adamw)')
    return AdamW(groups, lr=opts.learning_rate, betas=opts.betas)
