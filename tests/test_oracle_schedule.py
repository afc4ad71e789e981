"""Oracle schedule logic vs the reference's known answers (tests/golden/schedules.json, fixture G1)."""
import math

import pytest

from oracle.schedule import build_sampling_plan, build_step_tables, parse_sampling_schedule
from tests.helpers import jload

CASES = jload("schedules.json")


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['schedule']}-h{c['h']}-k{c['k']}-f{c['fac']}-b{int(c['before_t1'])}")
def test_step_tables_and_schedules(case):
    tab = build_step_tables(case["h"], case["schedule"], case["k"], case["fac"], case["before_t1"])
    assert tab.num_timesteps == case["num_timesteps"]
    assert {str(d): float(i) for d, i in tab.d_to_i.items()} == pytest.approx(case["d_to_i"], abs=1e-12)
    assert sorted(map(str, tab.dynamical_steps)) == sorted(case["dynamical_steps"])
    assert sorted(map(str, tab.artificial_steps)) == sorted(case["artificial_steps"])
    for name, want in case["schedules"].items():
        spec = None if name == "None" else name
        if not want["ok"]:
            with pytest.raises((AssertionError, ValueError, IndexError)):
                parse_sampling_schedule(tab, spec)
            continue
        got = parse_sampling_schedule(tab, spec)
        assert [float(s) for s in got] == pytest.approx(want["steps"], abs=1e-12), name
        assert all(isinstance(s, int) for s in got) == want["all_int"], name
        plan = build_sampling_plan(tab, got)
        assert plan[-1].is_last == (got[-1] == tab.num_timesteps - 1)


def test_appendix_e_known_answers():
    # SURVEY.md Appendix E (captured from the live reference)
    t = build_step_tables(7, "before_t1_only", 25, 0, True)
    assert t.num_timesteps == 32
    assert math.isclose(t.d_to_i[1], 1 / 26) and t.d_to_i[13] == 0.5 and t.d_to_i[26] == 1 and t.d_to_i[31] == 6
    assert parse_sampling_schedule(t, "only_dynamics") == [0, 26, 27, 28, 29, 30, 31]
    assert parse_sampling_schedule(t, "every5th") == [0, 1, 6, 11, 16, 21, 26, 27, 28, 29, 30, 31]
    assert parse_sampling_schedule(t, "first3") == [0, 1, 2, 3, 26, 27, 28, 29, 30, 31]
    assert len(parse_sampling_schedule(t, "first0.5")) == 20
    t = build_step_tables(5, "linear", 0, 2, False)
    assert t.num_timesteps == 11 and t.dynamical_steps == {1: 1.0, 4: 2.0, 7: 3.0, 10: 4.0}


def test_invalid_arguments_raise_like_reference():
    with pytest.raises(AssertionError):
        build_step_tables(1)
    with pytest.raises(AssertionError):
        build_step_tables(4, "before_t1_only", 0, 0, False)
    with pytest.raises(AssertionError):
        build_step_tables(4, "linear", 1, 0, True)
    with pytest.raises(ValueError):
        build_step_tables(4, "cosine", 0, 0, True)
    t = build_step_tables(4, "before_t1_only", 0, 0, True)
    with pytest.raises(ValueError):
        parse_sampling_schedule(t, "bogus")
    with pytest.raises(AssertionError):
        parse_sampling_schedule(t, [0, 2, 1, 3])
