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


def _svils_sources():
    return _glob(CSRC, (".hip",))


def _compile_objects(srcs, objdir, flags, deps_common, force):
    """one object per translation unit, stale ones compiled in parallel (svils_lpl.hip / svils_device.hip are minutes of
    hipcc each; the host-only units are seconds)"""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + deps_common):
            jobs.append([HIPCC] + flags + ["-c", "-o", obj, src])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(_run, jobs))
    return objs


def _build_svils_variant(name, extra_flags, force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    out = os.path.join(LIBDIR, name + ".so")
    srcs = _svils_sources()
    deps_common = _glob(CSRC, (".h",)) + [os.path.join(ROOT, "include", "svils.h")]
    objs = _compile_objects(srcs, os.path.join(LIBDIR, "obj_" + name), HIP_FLAGS + extra_flags, deps_common, force)
    if force or _stale(out, objs):
        _run([HIPCC] + HIP_FLAGS + ["-shared", "-o", out] + objs)
    return out


def build_svils(force=False):
    return _build_svils_variant("libsvils", [], force)


def build_stamps():
    """libsvils_stamps.so: the same kernels with wall-clock stamps at phase boundaries (tools/stamps.py)"""
    return _build_svils_variant("libsvils_stamps", ["-DSVILS_STAMPS"])


def build_testing(force=False):
    """libsvils_testing.so: the product sources + the two hooks that exist for the tests alone (-DSVILS_TESTING: fault
    injection into an in-launch hand-off, a pretended CU count; svils_options.h).  TEST INFRASTRUCTURE: tests select it
    with SVILS_LIB; the product library does not contain the hooks."""
    return _build_svils_variant("libsvils_testing", ["-DSVILS_TESTING"], force)


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
    elif "--testing" in sys.argv:
        build_testing("--force" in sys.argv)
    else:
        build_all("--force" in sys.argv)
