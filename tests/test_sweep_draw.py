"""R3 (app.py:699-707): the block-drawn weight generator against the reference's loop as written -- same accepted weight
vectors AND the same position of NumPy's legacy stream afterwards (the four random methods of app.py:682 share it), for
unconstrained, bounded, nearly infeasible and infeasible constraint sets.  CPU only."""
import numpy as np
import pytest

from monte_carlo_portfolio_amd import sweep

CASES = [(3, 2500, None, None), (16, 3000, None, None), (3, 2500, [0.05] * 3, [0.6] * 3), (3, 300, [0.3] * 3, [0.36] * 3),
         (4, 50, [0.9] * 4, [1.0] * 4), (5, 1000, [0.1] * 5, [0.5] * 5), (3, 400, [0.32] * 3, [0.35] * 3), (2, 1, None, None),
         (6, 257, [0.0] * 6, [0.3] * 6), (3, 513, [0.0, 0.0, 0.5], [1.0, 1.0, 1.0])]


@pytest.mark.parametrize("N,P,lo,hi", CASES)
@pytest.mark.parametrize("seed", [0, 12345])
def test_block_draw_equals_the_reference_loop(N, P, lo, hi, seed):
    np.random.seed(seed)
    want = sweep._draw_weights_loop(N, P, lo, hi)
    next_want = np.random.random(3)
    np.random.seed(seed)
    got = sweep.draw_weights(N, P, lo, hi)
    next_got = np.random.random(3)
    assert got.shape == want.shape and np.array_equal(got, want)
    assert np.array_equal(next_got, next_want)                 # the stream continues where the loop would have left it


def test_two_methods_in_a_row_share_the_stream():
    np.random.seed(7)
    a1, a2 = sweep._draw_weights_loop(3, 100, [0.1] * 3, [0.7] * 3), sweep._draw_weights_loop(3, 100, [0.1] * 3, [0.7] * 3)
    np.random.seed(7)
    b1, b2 = sweep.draw_weights(3, 100, [0.1] * 3, [0.7] * 3), sweep.draw_weights(3, 100, [0.1] * 3, [0.7] * 3)
    assert np.array_equal(a1, b1) and np.array_equal(a2, b2) and not np.array_equal(a1, a2)
