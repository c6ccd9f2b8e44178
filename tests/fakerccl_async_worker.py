"""worker of tests/test_gpu_fakerccl_async.py -- one rank talking to tests/fakerccl DIRECTLY (ctypes), with two torch
streams of its own: does the transport honour stream order and nothing more?

argv: out rank world case edge
case: producer -- the payload is written on stream P (behind ~tens of ms of other work), the all-reduce runs on stream C
      consumer -- the all-reduce runs on stream C (stretched by FAKERCCL_DELAY_US), its result is read on stream P
edge: 1 = the hipStreamWaitEvent a correct caller needs is there, 0 = it is missing
Writes the values the rank ended up with.
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
NCCL_FLOAT64, NCCL_SUM = 8, 0


class UID(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def main():
    out, rank, world, case, edge = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    lib = C.CDLL(os.path.join(HERE, "fakerccl", "libfakerccl.so"))
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    torch.cuda.set_device(0)
    uid = UID()
    f = out + ".id"
    if rank == 0:
        assert lib.ncclGetUniqueId(C.byref(uid)) == 0
        with open(f + ".tmp", "wb") as fh:
            fh.write(bytes(uid))
        os.rename(f + ".tmp", f)
    else:
        for _ in range(6000):
            if os.path.exists(f):
                break
            time.sleep(0.01)
        C.memmove(C.byref(uid), open(f, "rb").read(), 128)
    comm = C.c_void_p()
    assert lib.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0

    n = 1 << 16
    P, Cs = torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    b = torch.empty_like(a)
    buf = torch.zeros(n, dtype=torch.float64, device="cuda")
    seen = torch.zeros(n, dtype=torch.float64, device="cuda")
    if case == "consumer":
        buf.fill_(rank + 1)
    torch.cuda.synchronize()

    if case == "producer":
        with torch.cuda.stream(P):
            for _ in range(24):                      # tens of ms of work ahead of the payload
                torch.matmul(a, a, out=b)
            buf.fill_(rank + 1)
        if edge:
            Cs.wait_stream(P)
        assert lib.ncclAllReduce(buf.data_ptr(), buf.data_ptr(), n, NCCL_FLOAT64, NCCL_SUM, comm, C.c_void_p(Cs.cuda_stream)) == 0
        torch.cuda.synchronize()
        res = buf
    else:
        assert lib.ncclAllReduce(buf.data_ptr(), buf.data_ptr(), n, NCCL_FLOAT64, NCCL_SUM, comm, C.c_void_p(Cs.cuda_stream)) == 0
        if edge:
            P.wait_stream(Cs)
        with torch.cuda.stream(P):
            seen.copy_(buf)
        torch.cuda.synchronize()
        res = seen
    np.save(out + ".%d.npy" % rank, res.cpu().numpy())
    lib.ncclCommDestroy(comm)


if __name__ == "__main__":
    main()
