"""CPU: host-side logic that needs no GPU (task schedule, the deferred weight-gradient queue's bookkeeping)."""
import warnings

import pytest
import torch


def test_task_schedule_is_a_pure_function_of_seed_and_step():
    """ranks stay on the same task whatever the order / number of task_at() calls each of them makes
    (the reference broadcasts the task id instead, data/loader.py:56-59)."""
    from vln_hamt_amd.parallel import MIX_RATIO, TaskSchedule
    for cyclic in (True, False):
        a, b = TaskSchedule(seed=7, cyclic=cyclic), TaskSchedule(seed=7, cyclic=cyclic)
        fwd = [a.task_at(s) for s in range(200)]
        for s in (5, 5, 199, 0):                     # extra, out-of-order calls on the other "rank"
            b.task_at(s)
        bwd = [b.task_at(s) for s in reversed(range(200))][::-1]
        assert fwd == bwd
    mix = TaskSchedule(seed=1, cyclic=False)
    draws = [mix.task_at(s) for s in range(6000)]
    tot = sum(MIX_RATIO.values())
    for t, r in MIX_RATIO.items():
        assert abs(draws.count(t) / 6000 - r / tot) < 0.03, (t, draws.count(t))
    assert [TaskSchedule(seed=2, cyclic=False).task_at(s) for s in range(50)] != draws[:50]


def test_wgrad_queue_bookkeeping_cpu():
    """Work is keyed by the backward pass that queued it; orphans of a dead pass are dropped loudly by the next pass and
    never flushed; deferring outside a backward pass is an error."""
    from vln_hamt_amd import _lib as L, wgrad
    q = wgrad.WgradQueue()
    with pytest.raises(L.HamtError, match="inside a backward pass"):
        q.current()
    seen = []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, fail):
            ctx.fail = fail
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            ps = q.current()
            ps.items.append(("w", None, "dy", "x"))
            assert q.current() is ps                  # one _Pass per graph task
            if ctx.fail:
                raise RuntimeError("boom")
            return g, None

    q.handler = lambda items: seen.append(list(items))
    x = torch.ones(3, requires_grad=True)
    import vln_hamt_amd.wgrad as W
    orig = W._flush_pass
    W._flush_pass = lambda ps, handler: handler(ps.items) if ps.items else None     # no GPU here: only the routing is under test
    try:
        Probe.apply(x, False).sum().backward()
        assert len(seen) == 1 and q.pending() == 0
        with pytest.raises(RuntimeError, match="boom"):
            Probe.apply(x, True).sum().backward()
        assert q.pending() == 1 and len(seen) == 1   # the dead pass never reached its end-of-pass callback
        n0 = wgrad.stats["dropped_stale"]
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            Probe.apply(x, False).sum().backward()
        assert wgrad.stats["dropped_stale"] == n0 + 1 and any("did not finish" in str(w.message) for w in rec)
        assert len(seen) == 2 and len(seen[1]) == 1 and q.pending() == 0          # only the live pass's own item was flushed
        q.reset()
    finally:
        W._flush_pass = orig


def test_range_finality_and_static_cuts():
    """parallel.range_finality maps a step's plan (rank-local) onto the STATIC ranges: a static range is final after the last
    launch group that writes into any plan range overlapping it; the static ranges themselves depend on the layout only."""
    from vln_hamt_amd.parallel import range_finality, shard_cuts
    cuts = shard_cuts(4096 * 3, 4096 * 2, 2, parts=4)
    assert cuts == sorted(set(cuts)) and cuts[0] == 0 and cuts[-1] == 4096 * 3 and 4096 * 2 in cuts
    static = list(zip(cuts[:-1], cuts[1:]))
    plan_a = [(0, 3000, 0, frozenset({0})), (3000, 9000, 2, frozenset({1, 2})), (9000, 4096 * 3, 3, frozenset({0, 1, 2, 3}))]
    plan_b = [(0, 5000, 1, frozenset({0, 1})), (5000, 4096 * 3, 3, frozenset({2, 3}))]      # another rank: other cuts, other groups
    fa, fb = range_finality(static, plan_a), range_finality(static, plan_b)
    assert [(lo, hi) for lo, hi, _, _ in fa] == [(lo, hi) for lo, hi, _, _ in fb] == static
    assert fa[0][2] == 0 and fa[-1][2] == 3 and fb[0][2] == 1
    for lo, hi, after, touched in fa:
        assert after == max(touched)
