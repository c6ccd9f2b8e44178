import gzip
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # svils_sweep captures its hipGraphs only once a handle has run 128 sweeps (capture costs more than a short run).
    # Most tests run tens of sweeps: let them replay graphs from the first call of >= 4 sweeps on, so that the replay
    # path keeps the coverage it had; tests/test_gpu_parity.py::test_graph_capture_threshold runs the default.
    os.environ.setdefault("SVILS_GRAPH_AFTER", "0")
    # a clean checkout has no built artefacts (they are git-ignored): build them once, in-tree
    need = [os.path.join(ROOT, "svinet_amd", "lib", "libsvils.so"),
            os.path.join(ROOT, "svinet_amd", "lib", "libsvinet_host.so"),
            os.path.join(ROOT, "svinet_amd", "bin", "svinet")]
    if not all(os.path.exists(f) for f in need):
        from svinet_amd import build
        build.build_all()


def pytest_collection_modifyitems(config, items):
    """`-m gpu` on a box without a HIP device: skip instead of failing one test after the other --
    except the tests that check that the product fails LOUDLY there (no CPU fallback)."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no HIP device here; there is no CPU fallback to test instead)")
    for item in items:
        if "gpu" in item.keywords and "loud" not in item.name:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def graph_files(tmp_path_factory):
    """Decompressed copies of the example graphs the reference ships (tests/golden/graphs)."""
    d = tmp_path_factory.mktemp("graphs")
    out = {}
    for name, key in (("LFR-network-n1000-k28.txt.gz", "lfr"), ("ca-AstroPh.csv.gz", "astroph")):
        dst = os.path.join(str(d), name[:-3])
        with gzip.open(os.path.join(GOLDEN, "graphs", name), "rb") as f, open(dst, "wb") as g:
            g.write(f.read())
        out[key] = dst
    out["assort"] = os.path.join(GOLDEN, "graphs", "assort-75-4.txt")
    return out
