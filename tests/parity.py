"""Shared parity checker: compares a top-k result with the float64 oracle and
classifies every difference, so near-ties below the tolerance are reported,
not hidden."""
import numpy as np


def check_topk(got_idx, got_score, s64, k, tol, exact_idx=False):
    """got_idx/got_score: (B, k) result; s64: (B, N) oracle scores.

    Asserts
      * every reported score is within `tol` of the oracle score of that row,
      * padding (id -1, score -inf) appears exactly in the columns >= N,
      * wherever the reported row differs from the oracle's row at the same
        rank, the two rows' oracle scores differ by at most `tol` (a near-tie),
      * no row is reported twice,
      * with exact_idx, no differences at all.
    Returns the number of near-tie differences."""
    got_idx = np.asarray(got_idx)
    got_score = np.asarray(got_score, dtype=np.float64)
    b, n = s64.shape
    kk = min(k, n)
    assert got_idx.shape == (b, k) and got_score.shape == (b, k)
    order = np.argsort(-s64, axis=1, kind="stable")[:, :kk]
    want_score = np.take_along_axis(s64, order, axis=1)
    assert (got_idx[:, kk:] == -1).all(), "padding ids"
    assert np.isneginf(got_score[:, kk:]).all(), "padding scores"
    gi = got_idx[:, :kk]
    assert (gi >= 0).all() and (gi < n).all(), "row ids out of range"
    for r in range(b):
        assert len(set(gi[r].tolist())) == kk, f"query {r}: duplicate rows in result"
    at_got = np.take_along_axis(s64, gi, axis=1)
    err = np.abs(got_score[:, :kk] - at_got)
    assert err.max(initial=0.0) <= tol, f"score error {err.max():.3e} > {tol}"
    diff = gi != order
    if exact_idx:
        assert not diff.any(), f"{int(diff.sum())} index differences, first at {np.argwhere(diff)[0]}"
    gap = np.abs(at_got - want_score)
    assert (gap[diff] <= tol).all(), f"index difference with score gap {gap[diff].max():.3e} > {tol}"
    # reported order must be non-increasing in the reported scores, ties by id
    gs = got_score[:, :kk]
    assert (gs[:, :-1] >= gs[:, 1:]).all(), "reported scores not sorted"
    same = gs[:, :-1] == gs[:, 1:]
    assert (gi[:, :-1][same] < gi[:, 1:][same]).all(), "equal scores not ordered by row id"
    return int(diff.sum())
