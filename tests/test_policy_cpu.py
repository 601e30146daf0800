"""Host logic of ops.BackwardPolicy that needs no GPU: the cost-balanced partition handed to md_costvol_bwd as `shares`
(include/movedepth_hip.h: one [lo, hi) pair of hypothesis steps per workgroup, covering [0, items * D) exactly once) and the decoding
of the census word."""
import numpy as np

from movedepth_amd.ops import BackwardPolicy


def _policy():
    return BackwardPolicy.__new__(BackwardPolicy)      # (no device buffers: partition / _census are pure host code)


def test_partition_covers_every_step_once_and_equalises_cost():
    rng = np.random.default_rng(0)
    pol = _policy()
    for items, nwg, D in ((720, 720, 96), (2304, 2880, 128), (6, 24, 5), (60, 240, 32)):
        cost = rng.integers(50_000, 120_000, items).astype(np.int32)
        cost[rng.integers(0, items, max(1, items // 20))] *= 4          # a few heavy tiles
        sh = pol.partition(cost, nwg, D)
        assert sh.shape == (nwg, 2) and sh[0, 0] == 0 and sh[-1, 1] == items * D
        assert (sh[1:, 0] == sh[:-1, 1]).all() and (sh[:, 1] >= sh[:, 0]).all()
        inside = sh % D
        assert ((inside % 8 == 0) | (inside == 0)).all()                  # cuts at multiples of 8 steps inside an item (or at its borders)
        if D >= 96:
            # cost carried by each share under the model the partition assumes (uniform inside an item): near-equal -- as far as cuts at
            # multiples of 8 steps allow (a tile 15x heavier than a share's worth at D = 32 can only be cut in four: empty shares are valid)
            per_step = np.repeat(cost.astype(np.float64) / D, D)
            acc = np.concatenate([[0.0], np.cumsum(per_step)])
            load = acc[sh[:, 1]] - acc[sh[:, 0]]
            assert load.max() <= 1.6 * load.mean(), (items, load.max() / load.mean())
            steps = sh[:, 1] - sh[:, 0]
            assert steps.max() > steps[steps > 0].min()                        # equal cost, not equal steps


def test_partition_of_uniform_costs_is_the_equal_split():
    pol = _policy()
    sh = pol.partition(np.full(720, 1000, np.int32), 720, 96)
    assert (sh[:, 1] - sh[:, 0] == 96).all() and (sh[:, 0] == np.arange(720) * 96).all()


def test_census_word_decoding():
    g, t, w, s = 12_345 * 4, 69_120, 929, 720
    word = np.array([((g // 4) << 46) | ((t // 4) << 28) | (w << 14) | s], dtype=np.uint64)
    assert BackwardPolicy._census(word) == (g, t, w, s)


def test_step_outputs_lazy_entries():
    """trainer.StepOutputs: an entry registered with lazy() behaves as stored (`in`, [], get) and is produced once."""
    from movedepth_amd.trainer import StepOutputs

    calls = []
    o = StepOutputs()
    o["a"] = 1
    o.update({"b": 2})
    o.lazy(("sample", -1, 0), lambda: calls.append(1) or 42)
    assert "a" in o and ("sample", -1, 0) in o and "c" not in o
    assert ("sample", -1, 0) not in list(o.keys())
    assert o[("sample", -1, 0)] == 42 and o[("sample", -1, 0)] == 42 and calls == [1]
    assert ("sample", -1, 0) in list(o.keys())
    assert o.get("c") is None and o.get("b") == 2
    try:
        o["c"]
        raise AssertionError("missing key must raise")
    except KeyError:
        pass
