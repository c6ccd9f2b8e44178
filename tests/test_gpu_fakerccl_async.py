"""-m gpu: the tests' transport itself.  tests/fakerccl stands in for RCCL when several ranks share the one GPU of the
test box.  Its synchronous mode completes every collective at the call, which HIDES a missing event edge between the
caller's streams; its asynchronous mode (FAKERCCL_ASYNC=1) gives RCCL's contract and nothing more -- the payload is read
when the op's stream gets there, the result exists only for work ordered behind the op.  These tests drive the
transport directly with two streams and show that a missing producer- or consumer-side hipStreamWaitEvent gives a WRONG
answer in asynchronous mode (and that the synchronous mode would have hidden the consumer-side one), so that the
native multi-rank tests, which run in asynchronous mode, really test the library's stream ordering."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, case, edge, async_mode, delay_us=0, world=2):
    fake = os.path.join(HERE, "fakerccl", "libfakerccl.so")
    if not os.path.exists(fake):
        import __graft_entry__ as ge
        ge.build_test_transport()
    env = dict(os.environ)
    env.update({"FAKERCCL_ASYNC": "1" if async_mode else "0", "FAKERCCL_DELAY_US": str(delay_us), "FAKERCCL_TIMEOUT_S": "120"})
    out = str(tmp_path / ("%s_%d_%d" % (case, edge, async_mode)))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "fakerccl_async_worker.py"), out, str(r), str(world), case, str(edge)],
                              env=env, stderr=subprocess.PIPE, text=True) for r in range(world)]
    for r, p in enumerate(procs):
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, "rank %d:\n%s" % (r, err[-2000:])
    return [np.load(out + ".%d.npy" % r) for r in range(world)]


WANT = 3.0    # 1 + 2


def test_async_with_edges_is_correct(tmp_path):
    for case in ("producer", "consumer"):
        for res in _run(tmp_path, case, 1, True, delay_us=20000):
            assert np.all(res == WANT), case


def test_async_missing_producer_edge_is_wrong(tmp_path):
    """the all-reduce's stream never waited for the stream that writes the payload: the transport reads it too early"""
    res = _run(tmp_path, "producer", 0, True)
    assert all(np.all(r != WANT) for r in res)


def test_async_missing_consumer_edge_is_wrong(tmp_path):
    """the reader's stream never waited for the all-reduce: it sees the old bytes while the collective is in flight"""
    res = _run(tmp_path, "consumer", 0, True, delay_us=30000)
    for rank, r in enumerate(res):
        assert np.all(r == rank + 1.0)


def test_sync_mode_hides_the_missing_consumer_edge(tmp_path):
    """why the asynchronous mode exists: the same broken caller passes on the synchronous transport"""
    for r in _run(tmp_path, "consumer", 0, False):
        assert np.all(r == WANT)
