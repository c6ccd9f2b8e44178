"""-m gpu: the kernels' own special functions against scipy/numpy (fp64, the
accuracies DESIGN.md quotes): digamma (stands for gsl_sf_psi), exp for x <= 0,
reciprocal, table logarithm."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from svinet_amd._svils import Engine
    return Engine(8, 4, ones=1, ones_prob=0.1)


def test_digamma(eng):
    sp = pytest.importorskip("scipy.special")
    x = np.concatenate([np.logspace(-6, 9, 200001), np.linspace(0.001, 25, 100001), 1 / np.arange(2, 2049)])
    got = eng.debug_eval(0, x)
    want = sp.digamma(x)
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert err.max() < 4e-15, (err.max(), x[err.argmax()])


def test_exp_neg(eng):
    x = -np.abs(np.concatenate([np.random.default_rng(0).uniform(0, 60, 200000), np.linspace(0, 745, 100001), [0.0, 1e-300, 750.0, 1e4]]))
    got = eng.debug_eval(1, x)
    want = np.exp(x)
    ok = want > 1e-300
    assert np.max(np.abs(got[ok] - want[ok]) / want[ok]) < 4.5e-16
    assert np.all(got[~ok] < 1e-299)
    assert eng.debug_eval(1, np.array([0.0, -np.inf])).tolist() == [1.0, 0.0]


def test_rcp_and_log(eng):
    x = np.concatenate([np.logspace(-3, 14, 200001), np.linspace(10, 20, 10001)])
    r = eng.debug_eval(2, x)
    assert np.max(np.abs(r * x - 1.0)) < 4.5e-16
    y = x[x >= 1.0]
    lg = eng.debug_eval(3, y)
    assert np.max(np.abs(lg - np.log(y)) / np.maximum(np.log(y), 1.0)) < 4.5e-16
