"""-m gpu: pipelined reports (svils_report_enqueue / _ready / _fetch, include/svils.h) -- the reference's per-sweep
report block (src/linksampling.cc:777-786: likelihood row, max.txt, communities.txt) as snapshots taken in stream
order, collected while the device is already past them."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("graph,n,k", [("lfr", 1000, 28), ("astroph", 17903, 200)])
def test_report_is_the_state_at_its_place_in_the_stream(graph_files, graph, n, k):
    """three reports enqueued between batches of sweeps, NOT fetched until the device is 20 sweeps further: each one
    holds the control block, rows and community tags of its own point of the run -- those of an engine (and of the
    oracle) stopped there"""
    from svinet_amd.host_api import Setup
    setup = Setup(graph_files[graph], n, k)
    eng = setup.engine(use_validation_stop=False)
    marks, tickets, issued, rows_at = (4, 5, 11), [], 0, 0
    for m in marks:
        eng.sweep(m - issued)
        tickets.append((eng.report_enqueue(rows_at, m - rows_at, True), m, rows_at))
        issued, rows_at = m, m
    eng.sweep(20)                              # the device moves on; the snapshots must not
    ref = O.LinkSampling(O.Network(graph_files[graph], n), k, use_validation_stop=False)
    done = 0
    for t, m, first in tickets:
        other = setup.engine(use_validation_stop=False)
        other.sweep(m)
        while done < m:
            ref.sweep()
            done += 1
        c, rows, member = eng.report_fetch(t, m - first, True)
        oc = other.control()
        assert (c.iter, c.sweeps_done, c.rows, c.annealing) == (oc.iter, oc.sweeps_done, oc.rows, oc.annealing) == (m, m, m, int(ref.annealing))
        assert (c.links_dense, c.links_sparse, c.links_shortcut) == ref.link_counts()
        assert np.array_equal(rows, other.rows()[first:m])
        np.testing.assert_allclose(rows[:, 1:], np.asarray(ref.rows)[1 + first:1 + m, 1:], rtol=1e-7, atol=1e-12)
        assert np.array_equal(member, other.communities()) and np.array_equal(member, ref.communities())
        other.close()
    assert eng.control().sweeps_done == marks[-1] + 20


def test_report_after_the_stop_rule_and_slot_accounting(graph_files):
    """a report that names more rows than exist (the stop rule fired inside the chunk) returns the ones that do; at
    most SVILS_REPORT_SLOTS reports may be outstanding; tickets are single-use"""
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    setup = Setup(graph_files["lfr"], 1000, 28)
    eng = setup.engine(use_validation_stop=True)
    ref = O.LinkSampling(O.Network(graph_files["lfr"], 1000), 28)
    n_ref = 1
    while ref.sweep() != 2:
        n_ref += 1
    eng.sweep(n_ref - 10)
    t0 = eng.report_enqueue(0, 0, False)                  # no rows, no communities: the control block alone
    eng.sweep(40)                                         # the stop rule fires after 10 of them
    t1 = eng.report_enqueue(n_ref - 10, 40, True)
    c0, rows0, m0 = eng.report_fetch(t0, 0, False)
    assert not c0.stopped and c0.sweeps_done == n_ref - 10 and rows0.shape[0] == 0 and m0 is None
    c1, rows1, m1 = eng.report_fetch(t1, 40, True)
    assert c1.stopped == 1 and c1.sweeps_done == n_ref and c1.iter == ref.iter and rows1.shape[0] == 10
    assert np.array_equal(rows1, eng.rows()[n_ref - 10:]) and np.array_equal(m1, ref.communities())
    with pytest.raises(_svils.SvilsError):
        eng.report_fetch(t1, 40, True)                    # the slot is free again: not a ticket any more
    ts = [eng.report_enqueue(0, 1, False) for _ in range(4)]
    with pytest.raises(_svils.SvilsError):
        eng.report_enqueue(0, 1, False)                   # SVILS_REPORT_SLOTS outstanding
    for t in ts:
        eng.report_fetch(t, 1, False)
    with pytest.raises(_svils.SvilsError):
        eng.report_enqueue(0, 65, False)                  # SVILS_REPORT_MAX_ROWS


@pytest.mark.parametrize("graph,n,k", [("lfr", 1000, 28), ("astroph", 17903, 200)])
def test_community_tags_are_the_member_matrix(graph_files, graph, n, k):
    """svils_get_community_tags / svils_report_fetch_tags: the (node, community) pairs of exactly the ones of the member
    matrix, by ascending node, for both layouts (K = 28: lane per link, K = 200: row per wavefront with its interleaved
    lane layout of the bitmask); too little room is an error that keeps the slot; a report without communities has none."""
    from svinet_amd import _svils
    from svinet_amd.host_api import Setup
    setup = Setup(graph_files[graph], n, k)
    eng = setup.engine(use_validation_stop=False)
    eng.sweep(12)
    t = eng.report_enqueue(0, 12, True)
    eng.sweep(3)
    member = eng.communities()                                  # after 15 sweeps
    want = np.argwhere(member)                                  # row-major: ascending node, then community
    assert want.shape[0] > 100
    tags = eng.community_tags()
    assert tags.dtype == np.uint32 and np.all(np.diff(tags[:, 0].astype(np.int64)) >= 0)
    assert np.array_equal(tags[np.lexsort((tags[:, 1], tags[:, 0]))], want)
    with pytest.raises(_svils.SvilsError):
        eng.report_fetch_tags(t, 12, cap=3)                     # does not fit: nothing is freed
    c, rows, rtags = eng.report_fetch_tags(t, 12)               # the snapshot taken after 12 sweeps
    other = setup.engine(use_validation_stop=False)
    other.sweep(12)
    assert c.sweeps_done == 12 and rows.shape[0] == 12 and rtags.shape[0] > 3
    assert np.array_equal(rtags[np.lexsort((rtags[:, 1], rtags[:, 0]))], np.argwhere(other.communities()))
    with pytest.raises(_svils.SvilsError):
        eng.report_fetch_tags(t, 12)                            # the slot went with the successful fetch
    t2 = eng.report_enqueue(0, 1, False)
    with pytest.raises(_svils.SvilsError):
        eng.report_fetch_tags(t2, 1)                            # enqueued without communities
    eng.report_fetch(t2, 1, False)
    other.close()
    eng.close()
