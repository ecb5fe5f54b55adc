"""Learning-rate schedules with the reference's semantics (pretrain_src/optim/sched.py:10-30)."""


def noam_schedule(step, warmup_step=4000):
    if step <= warmup_step:
        return step / warmup_step
    return (warmup_step ** 0.5) * (step ** -0.5)


def warmup_linear(step, warmup_step, tot_step):
    if step < warmup_step:
        return step / warmup_step
    return max(0, (tot_step - step) / (tot_step - warmup_step))


def get_lr_sched(global_step, opts):
    lr = opts.learning_rate * warmup_linear(global_step, opts.warmup_steps, opts.num_train_steps)
    return lr if lr > 0 else 1e-8
