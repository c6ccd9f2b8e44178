"""The DROP-IN binary at config-5 scale: `svinet -file <planted MMSB graph, n = 1e6> -n 1000000 -k 512 -link-sampling
-no-stop -max-iterations M` on one GPU, end to end -- reading the 12 M-line edge list, the constructor (held-out sample,
6e9 MT19937 draws of init_gamma2), the sweeps with their reports (communities.txt from tag pairs), and the final
gamma.txt / lambda.txt / groups.txt (512 M numbers each).  Prints the CLI's own clocks (SVINET_TIMING_FILE), the wall time
and the sizes of the files; the state is checked through the C ABI run of the same inputs by tests/test_gpu_config5.py, here
only lambda.txt is compared with an engine run of the same sweeps (1e-5 relative: the file carries five decimals).
  python tools/cli_config5.py [max_iterations | stop] [n] [k]        (stop: the default flags, the run ends on its stop rule)
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svinet_amd import mmsbgen_sparse as G          # noqa: E402


def main():
    to_stop = len(sys.argv) > 1 and sys.argv[1] == "stop"
    M = 0 if to_stop else (int(sys.argv[1]) if len(sys.argv) > 1 else 4)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    d = tempfile.mkdtemp(prefix="svinet_cfg5_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        t0 = time.perf_counter()
        pairs = G.generate(n, k, 24)
        path = os.path.join(d, "graph.txt")
        with open(path, "w") as f:                      # "%d\t%d\n" per link (Network::read, src/network.cc:10-116)
            for lo in range(0, pairs.shape[0], 1 << 20):
                blk = pairs[lo:lo + (1 << 20)]
                f.write("\n".join("%d\t%d" % (a, b) for a, b in blk.tolist()) + "\n")
        print("graph: %d links, file %.0f MB, generated + written in %.1f s" % (pairs.shape[0], os.path.getsize(path) / 1e6, time.perf_counter() - t0), flush=True)
        tf = os.path.join(d, "timing.json")
        env = dict(os.environ, SVINET_TIMING_FILE=tf, SVINET_TRACE_LOOP="1")   # (the trace marks go to stderr: where the constructor and the writers spend their time)
        # CFG5_TESTING_LIB=1: the run binds libsvils_testing.so (LD_PRELOAD), whose svils_init_gamma prints where its time went
        if os.environ.get("CFG5_TESTING_LIB"):
            env["LD_PRELOAD"] = os.path.join(ROOT, "svinet_amd", "lib", "libsvils_testing.so")
        # CFG5_THREADS=16,32,64: the same run once per thread count of the host-side pools (init_gamma2, file writers)
        for th in [x for x in os.environ.get("CFG5_THREADS", "").split(",") if x]:
            e2 = dict(env, SVINET_INIT_THREADS=th, SVINET_WRITE_THREADS=th)
            t1 = time.perf_counter()
            r = subprocess.run([os.path.join(ROOT, "svinet_amd", "bin", "svinet"), "-file", path, "-n", str(n), "-k", str(k), "-link-sampling",
                                "-no-stop", "-max-iterations", str(M), "-label", "t" + th], cwd=d, env=e2, capture_output=True, text=True, timeout=3000)
            tm = json.load(open(tf))
            print("threads %s: wall %.1f s read %.2f ctor %.2f sweeps %.2f final %.2f" % (th, time.perf_counter() - t1, tm["read_s"], tm["ctor_s"], tm["sweeps_s"], tm["final_files_s"]), flush=True)
            for line in r.stderr.split("\n"):
                if line.startswith("[final]") or line.startswith("[ctor]"):
                    print("    " + line)
            for x in os.listdir(d):
                if os.path.isdir(os.path.join(d, x)):
                    shutil.rmtree(os.path.join(d, x))
        t1 = time.perf_counter()
        r = subprocess.run([os.path.join(ROOT, "svinet_amd", "bin", "svinet"), "-file", path, "-n", str(n), "-k", str(k), "-link-sampling"]
                           + ([] if to_stop else ["-no-stop", "-max-iterations", str(M)]), cwd=d, env=env, capture_output=True, text=True, timeout=3000)
        wall = time.perf_counter() - t1
        print("svinet rc=%d wall %.1f s" % (r.returncode, wall), flush=True)
        if r.returncode:
            print(r.stderr[-2000:])
            sys.exit(1)
        print("timing:", json.dumps(json.load(open(tf))), flush=True)
        for line in r.stderr.split("\n"):       # SVINET_TRACE_LOOP=1: the binary's own marks
            if line.startswith(("[ctor]", "[final]", "[attach]", "[svils_init_gamma]")):
                print("  " + line)
        outdir = [x for x in os.listdir(d) if os.path.isdir(os.path.join(d, x))][0]
        for fn in sorted(os.listdir(os.path.join(d, outdir))):
            print("  %-28s %12.1f MB" % (fn, os.path.getsize(os.path.join(d, outdir, fn)) / 1e6))
        lam = np.loadtxt(os.path.join(d, outdir, "lambda.txt"))[:, 1:]
        val = np.loadtxt(os.path.join(d, outdir, "validation.txt"))
        print("validation.txt rows:", val.shape, "last:", val[-1].tolist())
        if to_stop:
            print("max.txt:", open(os.path.join(d, outdir, "max.txt")).read().strip())
            print("ok (ran to its stop rule)")
            return
        # the same sweeps through the C ABI
        from svinet_amd.host_api import Setup
        s = Setup(path, n, k)
        eng = s.engine(use_validation_stop=False)
        eng.sweep(M + 1)                                # -max-iterations M => M + 1 sweeps (quirk Q8)
        _, l2, _ = eng.state()
        rel = float(np.max(np.abs(lam - l2) / np.maximum(np.abs(l2), 1e-3)))
        print("lambda.txt vs engine after %d sweeps: max rel diff %.2e" % (M + 1, rel))
        assert rel < 2e-5, rel
        print("ok")
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
