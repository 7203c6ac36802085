"""Pins the C restatement of the matcher (oracle/lsap_oracle.c) against the reference's real
dependency, scipy.optimize.linear_sum_assignment, called as losses_and_metrics.py:240-243 does."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st
from scipy.optimize import linear_sum_assignment

from oracle import lsap


def check(cost):
    r0, c0 = linear_sum_assignment(cost)
    r1, c1 = lsap.linear_sum_assignment_f32(cost)
    assert np.array_equal(r0, r1) and np.array_equal(c0, c1)


@pytest.mark.parametrize("nr,nc", [(1, 1), (1, 7), (7, 1), (5, 5), (20, 50), (93, 100), (100, 100), (30, 300), (120, 100), (300, 50)])
def test_random(nr, nc):
    rng = np.random.default_rng(nr * 1000 + nc)
    for _ in range(5):
        check(rng.random((nr, nc)).astype(np.float32))


def test_ties():
    rng = np.random.default_rng(0)
    check(np.zeros((3, 5), np.float32))
    check(np.ones((5, 3), np.float32))
    check(np.array([[1, 1, 0], [0, 1, 1.]], np.float32))
    for _ in range(50):
        nr, nc = rng.integers(1, 40, size=2)
        check(rng.integers(0, 3, size=(nr, nc)).astype(np.float32))
        base = rng.integers(0, 4, size=(nr, max(1, nc // 3))).astype(np.float32)
        check(np.tile(base, (1, 3)))


def test_inf_nan():
    c = np.random.default_rng(1).random((4, 6)).astype(np.float32)
    c[1, :3] = np.inf
    check(c)
    c[1, :] = np.inf
    with pytest.raises(ValueError):
        linear_sum_assignment(c)
    with pytest.raises(ValueError):
        lsap.linear_sum_assignment_f32(c)
    c[1, :] = np.nan
    with pytest.raises(ValueError):
        lsap.linear_sum_assignment_f32(c)
    assert lsap.linear_sum_assignment_f32(np.zeros((0, 4), np.float32))[0].size == 0


@settings(max_examples=150, deadline=None)
@given(st.integers(1, 24), st.integers(1, 24), st.integers(0, 2 ** 31 - 1), st.sampled_from([2, 4, 1000, 0]))
def test_hypothesis(nr, nc, seed, levels):
    rng = np.random.default_rng(seed)
    c = rng.random((nr, nc))
    if levels:
        c = np.round(c * levels) / levels       # quantised costs -> many exact ties
    check(c.astype(np.float32))


def test_mask_matches_reference_call_pattern():
    rng = np.random.default_rng(5)
    cost = rng.random((6, 20, 50)).astype(np.float32)
    nobj = np.array([0, 1, 20, 7, 3, 12], np.int32)
    want = np.zeros_like(cost)
    for i in range(6):
        r, c = linear_sum_assignment(cost[i, :nobj[i], :])
        want[i][r, c] = 1.0
    assert np.array_equal(lsap.assignment_mask(cost, nobj), want)
