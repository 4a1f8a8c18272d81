"""The NNLS termination rule, pinned on the oracle alone (no GPU).

The reference's active-set loops (src/utils/update.cpp:95-165) have no iteration bound; in floating point its
exchange rule cycles on some inputs and the function never returns.  The oracle -- and, identically, the device
kernels -- stop a row after max(64, 16 r) passes of any loop and at the first main-loop pass that ends on the
set it started from, and flag it (status 2).  This is a deviation from the reference; what makes it harmless is
the claim checked here: ON EVERY INPUT ON WHICH THE UNBOUNDED LOOPS END, the bounded function returns the same
status (0), the bit-identical row and the same active set -- i.e. neither the bound nor the cycle rule ever
trips on a terminating row.  "Unbounded" = the same code with the bound lifted to 20 000 passes and the cycle
rule switched off (or_nnls_set_termination); a row that has not finished after 20 000 passes counts as
non-terminating and is only required to be flagged by the bounded run as well."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402

LIFTED = 20000


def _problems(seed, n_problems):
    """(H, g rows, warm) triples shaped like what a sweep hands the update: H = Hadamard of two Gramians of
    random factor matrices, g = rows of an MTTKRP-like panel.  Regimes: non-negative data (every component of
    a row starts passive), signed data (many active constraints, collapsing components), near-collinear
    factors (ill-conditioned H), exact zeros in g."""
    rng = np.random.default_rng(seed)
    for _ in range(n_problems):
        r = int(rng.integers(1, 65)) if rng.random() < 0.7 else int(rng.integers(1, 21))
        regime = rng.integers(0, 4)
        rows_a, rows_b = int(rng.integers(max(2, r // 2), 3 * r + 8)), int(rng.integers(max(2, r // 2), 3 * r + 8))
        if regime == 0:      # non-negative factors
            A, B = rng.random((rows_a, r)), rng.random((rows_b, r))
        elif regime == 1:    # signed
            A, B = rng.uniform(-1, 1, (rows_a, r)), rng.uniform(-1, 1, (rows_b, r))
        elif regime == 2:    # near-collinear columns
            base_a, base_b = rng.random((rows_a, 1)), rng.random((rows_b, 1))
            A = base_a + 1e-3 * rng.random((rows_a, r))
            B = base_b + 1e-3 * rng.random((rows_b, r))
        else:                # mixed scales
            A = rng.random((rows_a, r)) * 10.0 ** rng.integers(-3, 4, (1, r))
            B = rng.uniform(-1, 1, (rows_b, r))
        H = (A.T @ A) * (B.T @ B)
        n_rows = int(rng.integers(1, 4))
        x_true = rng.random((n_rows, r)) * (rng.random((n_rows, r)) < 0.6)
        G = x_true @ H + rng.normal(0, 10.0 ** rng.integers(-8, 0), (n_rows, r)) * np.abs(H).max()
        if regime == 1 and rng.random() < 0.5:
            G = rng.uniform(-1, 1, (n_rows, r)) * np.abs(H).max()
        if rng.random() < 0.1:
            G[:, rng.integers(0, r)] = 0.0
        yield H, G, bool(rng.random() < 0.5)


def _solve(H, g_row, active, bound, cycle_rule):
    O.nnls_set_termination(bound, cycle_rule)
    try:
        x, act, st = O.update_factor_nnls(g_row[None, :].copy(), H, None if active is None else active.copy())
        return x[0].copy(), act[0].copy(), st, O.nnls_last_max_passes()
    finally:
        O.nnls_set_termination(0, True)


@pytest.mark.parametrize("seed", [7, 8])
def test_bound_and_cycle_rule_never_trip_on_a_terminating_row(seed):
    n_rows = n_term = n_nonterm = 0
    worst = (0, 0.0)   # (passes, passes / bound) over terminating rows
    for H, G, warm in _problems(seed, 700):
        r = H.shape[0]
        bound = max(64, 16 * r)
        for g in G:
            start = None
            if warm:   # the active set a previous sweep left behind: solve a perturbed row first
                g0 = g * (1.0 + 0.05 * np.cos(np.arange(r))) + 1e-3 * np.abs(g).max() * np.sin(np.arange(r) * 1.7)
                _, start, st0, _ = _solve(H, g0, None, 0, True)
                if st0 != 0:
                    start = None
            xl, al, sl, passes = _solve(H, g, start, LIFTED, False)
            xb, ab, sb, _ = _solve(H, g, start, 0, True)
            n_rows += 1
            if sl == 0:
                n_term += 1
                assert sb == 0, "bound or cycle rule tripped on a terminating row (rank %d, %d passes)" % (r, passes)
                assert np.array_equal(xl, xb) and np.array_equal(al, ab)
                if passes / bound > worst[1]:
                    worst = (passes, passes / bound)
            elif sl & 2:
                n_nonterm += 1
                assert sb & 2, "a row that does not terminate must be flagged"
            else:   # status 1: the Cholesky failure the reference dies on -- same in both runs
                assert sb == sl
    assert n_rows >= 1000 and n_term >= 0.9 * n_rows
    # the margin the bound leaves: no terminating row came anywhere near it
    print("rows %d, terminating %d, non-terminating %d; most passes on a terminating row: %d = %.3f of its bound"
          % (n_rows, n_term, n_nonterm, worst[0], worst[1]))
    assert worst[1] <= 0.5
