"""The learning-rate schedule the pretraining loop uses (pretrain_src/optim/sched.py:17-30: linear warm-up, linear decay, floor 1e-8;
main_r2r.py:255).  The reference's `noam_schedule` (sched.py:10-14) has no caller on the path and is not mirrored."""


def warmup_linear(step, warmup_step, tot_step):
    if step < warmup_step:
        return step / warmup_step
    return max(0, (tot_step - step) / (tot_step - warmup_step))


def get_lr_sched(global_step, opts):
    lr = opts.learning_rate * warmup_linear(global_step, opts.warmup_steps, opts.num_train_steps)
    return lr if lr > 0 else 1e-8
