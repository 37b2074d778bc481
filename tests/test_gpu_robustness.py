"""Degenerate inputs must not turn into addresses (-m gpu).  A row of logits without a single comparable value (all
NaN, all -inf: a broken checkpoint or an overflow upstream) has no arg-max; the sampler then yields token 0 instead of
the sentinel id, which the kernels behind it (probability gather, next step's embedding) would use as an index."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_gpu_kernels import _run_sampler  # noqa: E402


@pytest.mark.parametrize("temp,top_p", [(0.0, 1.0), (1.0, 0.9), (0.8, 1.0)])
def test_sampler_rows_without_a_maximum_yield_a_valid_token(temp, top_p):
    V = 4096
    lg = np.random.default_rng(0).standard_normal((4, V)).astype(np.float32)
    lg[1] = np.nan
    lg[2] = -np.inf
    lg[3, ::2] = np.nan                                   # half NaN: the comparable half still decides
    toks, lp, p0, ki, kl = _run_sampler(lg, temp, top_p, [0.3, 0.3, 0.3, 0.3], k=2)
    assert ((0 <= toks) & (toks < V)).all() and ((0 <= ki) & (ki < V)).all()
    assert toks[1] == 0 and toks[2] == 0
    if temp == 0.0:
        assert toks[0] == int(np.argmax(lg[0])) and toks[3] == 2 * int(np.argmax(lg[3, 1::2])) + 1
