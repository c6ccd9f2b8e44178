"""In-tree build of the native pieces (no JIT cache: the .so files must travel
with the repo snapshot to the GPU box).

  python -m svinet_amd.build            # build what is stale
  python -m svinet_amd.build --force
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIBDIR = os.path.join(HERE, "lib")
BINDIR = os.path.join(HERE, "bin")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
CXX = os.environ.get("CXX", "g++")
CXX_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-pthread"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+ " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _glob(d, exts):
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts)) if os.path.isdir(d) else []


def build_svils(force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    out = os.path.join(LIBDIR, "libsvils.so")
    srcs = [os.path.join(CSRC, f) for f in ("svils_api.hip", "svils_device.hip", "svils_lpl.hip", "svils_report.hip")]
    deps = srcs + _glob(CSRC, (".h",)) + [os.path.join(ROOT, "include", "svils.h")]
    if force or _stale(out, deps):
        _run([HIPCC] + HIP_FLAGS + ["-shared", "-o", out] + srcs)
    return out


def build_stamps():
    """libsvils_stamps.so: the same kernels with wall-clock stamps at phase boundaries (tools/stamps.py)"""
    out = os.path.join(LIBDIR, "libsvils_stamps.so")
    srcs = [os.path.join(CSRC, f) for f in ("svils_api.hip", "svils_device.hip", "svils_lpl.hip", "svils_report.hip")]
    _run([HIPCC] + HIP_FLAGS + ["-DSVILS_STAMPS", "-shared", "-o", out] + srcs)
    return out


def build_host(force=False):
    """C++ host side: libsvinet_host.so (C entry points for Python) and the svinet CLI."""
    srcs = _glob(HOST, (".cc",))
    if not srcs:
        return None
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(BINDIR, exist_ok=True)
    deps = srcs + _glob(HOST, (".hh", ".h")) + [os.path.join(ROOT, "include", "svils.h")]
    lib_srcs = [s for s in srcs if not s.endswith("main.cc")]
    hostlib = os.path.join(LIBDIR, "libsvinet_host.so")
    if force or _stale(hostlib, deps):
        _run([CXX] + CXX_FLAGS + ["-shared", "-o", hostlib] + lib_srcs +
             ["-I", os.path.join(ROOT, "include"), "-L", LIBDIR, "-lsvils", "-Wl,-rpath,$ORIGIN"])
    exe = os.path.join(BINDIR, "svinet")
    main = os.path.join(HOST, "main.cc")
    if os.path.exists(main) and (force or _stale(exe, deps + [hostlib])):
        _run([CXX] + CXX_FLAGS + ["-o", exe, main, "-I", os.path.join(ROOT, "include"),
              "-L", LIBDIR, "-lsvinet_host", "-lsvils", "-Wl,-rpath,$ORIGIN/../lib"])
    return hostlib


def build_all(force=False):
    build_svils(force)
    build_host(force)


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        build_stamps()
    else:
        build_all("--force" in sys.argv)
