"""The threaded variant of the oracle (oracle/svinet_oracle_omp.c -- bench.py's cpu_baseline_allcores, an all-cores
CPU figure; NOT the reference's summation order) against the sequential oracle that is pinned on the reference's
golden data: same link counts, flags and tags, gamma / lambda / likelihood rows to rounding."""
import numpy as np
import pytest

from oracle import oracle as O


@pytest.mark.parametrize("key,k,sweeps,threads", [("lfr", 28, 60, 3), ("assort", 4, 25, 2)])
def test_threaded_sweeps_equal_sequential_ones(graph_files, key, k, sweeps, threads):
    net = O.Network(graph_files[key])
    a = O.LinkSampling(net, k)
    b = O.LinkSampling(net, k)
    for it in range(sweeps):
        ra, rb = a.sweep(), b.sweep_omp(threads)
        assert ra == rb
        assert a.link_counts() == b.link_counts(), it
        if ra:
            break
    assert np.array_equal(a.converged, b.converged)
    assert np.array_equal(a.active_comms, b.active_comms)
    np.testing.assert_allclose(b.gamma, a.gamma, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(b.lam, a.lam, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(b.rows, a.rows, rtol=1e-9, atol=1e-12)
    assert np.array_equal(a.communities(), b.communities())
