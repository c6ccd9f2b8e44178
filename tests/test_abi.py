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
    assert lib.svils_abi_version() == 7


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
