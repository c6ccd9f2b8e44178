"""The C-ABI library loads on a CPU-only box, exports every symbol that
include/svils.h declares, and refuses to run without a HIP device."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    hdr = open(os.path.join(ROOT, "include", "svils.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(svils_[a-z_0-9]+)\s*\(", hdr)))


def test_header_symbols_are_exported():
    from svinet_amd import _svils
    lib = _svils.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libsvils.so does not export %s" % n
    assert sorted(_svils.EXPORTS) == names
    assert lib.svils_abi_version() == 8


def test_kernel_names():
    from svinet_amd import _svils
    lib = _svils.load()
    assert [lib.svils_kernel_name(i).decode() for i in range(len(_svils.KERNEL_NAMES))] == list(_svils.KERNEL_NAMES)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from svinet_amd import _svils
    with pytest.raises(_svils.SvilsError) as ei:
        _svils.Engine(10, 4, ones=5, ones_prob=0.1)
    assert ei.value.code == -2 and "no CPU path" in str(ei.value)


def test_argument_checks_do_not_need_a_device():
    from svinet_amd import _svils
    lib = _svils.load()
    cfg = _svils.Config()
    assert lib.svils_config_default(ctypes.byref(cfg), 100, 0) == -1
    assert lib.svils_config_default(ctypes.byref(cfg), 100, 20) == 0
    assert (cfg.alpha, cfg.eta0, cfg.eta1, cfg.epsilon, cfg.link_thresh, cfg.reportfreq) == (1 / 20, 1.0, 1.0, 1e-30, 0.5, 1)
    h = ctypes.c_void_p()
    cfg.k = 70000    # (k in 2049..65535 is a column-tiled handle since ABI 7: it gets as far as the device check)
    assert lib.svils_create(ctypes.byref(cfg), ctypes.byref(h)) == -4      # SVILS_ERR_UNSUPPORTED
    assert b"SVILS_MAX_K_TOTAL" in lib.svils_last_error()
    assert lib.svils_sweep(None, 1) == -1


def test_product_does_not_touch_the_oracle():
    """the shipped path must never import/link anything under oracle/"""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "svinet_amd")):
        for f in files:
            if f.endswith((".py", ".cc", ".hh", ".h", ".hip")):
                txt = open(os.path.join(d, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle|svinet_oracle|orc_ls_|orc_net_", txt, flags=re.M):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_option_table_is_the_only_reader_of_the_environment():
    """every tunable is a row of ONE table (svils_option_table), DESIGN.md prints it, nothing on a sweep
    path reads the environment, and the tests' two hooks are not in the product library."""
    from svinet_amd import _svils
    rows = _svils.option_table()
    keys = [r["key"] for r in rows if r["key"] != "-"]
    assert len(keys) == len(set(keys)) >= 14 and "sharded_graphs" in keys and "xchunks" in keys
    assert "fault_inject" not in keys and "assume_cus" not in keys            # -DSVILS_TESTING builds only
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for r in rows:
        assert r["environment"] in design, "DESIGN.md does not list %s" % r["environment"]
        assert r["key"] == "-" or ("`%s`" % r["key"]) in design, "DESIGN.md does not list option %s" % r["key"]
    # getenv in the library's sources: the option table, the RCCL library name (first svils_comm_init of the process) and
    # the runtime's own CU-mask variables (once per svils_set_graph) -- nowhere else
    csrc = os.path.join(ROOT, "svinet_amd", "csrc")
    hits = {}
    for f in sorted(os.listdir(csrc)):
        n = len(re.findall(r"\bgetenv\s*\(", open(os.path.join(csrc, f)).read()))
        if n:
            hits[f] = n
    assert set(hits) == {"svils_options.hip", "svils_comm.hip", "svils_api.hip"}, hits
    assert hits["svils_comm.hip"] == 1 and hits["svils_api.hip"] == 2
    blob = open(os.path.join(ROOT, "svinet_amd", "lib", "libsvils.so"), "rb").read()
    assert b"SVILS_FAULT_INJECT" not in blob and b"SVILS_ASSUME_CUS" not in blob
    testing = os.path.join(ROOT, "svinet_amd", "lib", "libsvils_testing.so")
    if os.path.exists(testing):
        tb = open(testing, "rb").read()
        assert b"SVILS_FAULT_INJECT" in tb and b"SVILS_ASSUME_CUS" in tb


def test_set_option_argument_checks():
    from svinet_amd import _svils
    lib = _svils.load()
    assert lib.svils_set_option(None, b"xchunks", b"2") == -1
    assert b"key\tenvironment\tdefault" in lib.svils_option_table()
