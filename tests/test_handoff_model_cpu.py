"""k_scan_one's hand-off protocol, model-checked (tests/handoff_model.py): every interleaving of the agent-scope memory operations of
2 workgroups x {1, 2} consumer waves, 2 items, 2 launches back to back on the same queue memory.  The protocol as the kernel runs
it has no stuck execution, loses no entry and leaves its memory clean; each of the two bugs round 5 found by stress-testing
(profiles/r05_experiments.md section 1) is found by the search when it is switched back on.  core/pigo.go:212-258 (RunCascade
returns every window that passes: none may be lost)."""
import pytest

from handoff_model import Config, explore


@pytest.mark.parametrize("waves,items,items2", [(1, (1, 1), None), (2, (1, 1), None), (1, (2, 0), None), (2, (0, 1), (1, 1)), (1, (0, 0), (1, 1)),
                                                (1, (0, 0), (2, 1))])
def test_protocol_has_no_stuck_or_lossy_execution(waves, items, items2):
    r = explore(Config(grid=2, waves=waves, items=items, items2=items2, qcap=4, launches=2))
    assert r["violations"] == [], r
    assert r["states"] > 100


@pytest.mark.parametrize("kw", [dict(grid=3, waves=1, items=(1, 1)), dict(grid=3, waves=1, items=(0, 1), items2=(1, 1)),
                                dict(grid=2, waves=2, items=(0, 0), items2=(2, 1)), dict(grid=2, waves=2, items=(1, 0, 1)),
                                dict(grid=4, waves=1, items=(1, 0), items2=(1, 1))])
def test_more_agents_two_launches(kw):
    """Three or four workgroups / four consumer waves, two launches on the same memory (tens of thousands of states each)."""
    r = explore(Config(qcap=4, launches=2, **kw))
    assert r["violations"] == [], r


def test_both_bugs_are_found_with_three_workgroups_too():
    r = explore(Config(grid=3, waves=1, items=(1, 1), qcap=4, launches=1, bug="claim_after_done"))
    assert any("stuck" in v for v in r["violations"]), r
    r = explore(Config(grid=3, waves=1, items=(0, 0), items2=(1, 1), qcap=4, launches=2, bug="no_cleanup"))
    assert any("lost" in v or "stuck" in v for v in r["violations"]), r


def test_a_claim_that_may_follow_its_read_of_done_hangs():
    """Bug 1 of round 5: without the s_waitcnt between the claim and the read of `done` both are in flight together."""
    r = explore(Config(grid=2, waves=1, items=(1, 1), qcap=4, launches=1, bug="claim_after_done"))
    assert any("stuck" in v for v in r["violations"]), r


def test_poison_left_behind_loses_a_window_in_the_next_launch():
    """Bug 2 of round 5: the last workgroup must zero [alloc, head) -- the poison nobody took."""
    one = explore(Config(grid=2, waves=1, items=(1, 1), qcap=4, launches=1, bug="no_cleanup"))
    assert one["violations"] == [], "the bug is invisible within one launch"
    # (the second frame keeps more windows alive than the first: the slot the stale poison sits in gets a real entry)
    two = explore(Config(grid=2, waves=1, items=(0, 0), items2=(1, 1), qcap=4, launches=2, bug="no_cleanup"))
    assert any("lost" in v or "stuck" in v for v in two["violations"]), two
