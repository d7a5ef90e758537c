"""Python binding of modal::SolveBatch (modal/batch.hpp, libmodalbatch.so -- the one host library that links RCCL and the HIP
runtime directly; libmodalhost.so, which everything else binds, does not): a batch of independent meshes dealt over the
ranks' GPUs by the LPT rule, solved by host threads per GPU, the fixed-size records gathered with ONE ncclAllGather -- the host
side in C++, RCCL called directly.  The communicator's 128-byte id is made on rank 0 and shipped by the caller (bench.py: the
launcher's TCP store); nothing else of the data path touches Python.

A process that has imported PyTorch carries a second copy of the ROCm runtime libraries (the wheel bundles its own), and which
copy the system RCCL then binds to depends on load order: create the BatchComm before importing torch, or after torch has
initialised the GPU (bench.py does the latter).  A C++ host has no such second runtime."""
import ctypes as C

import numpy as np

from . import sharding


class _Item(C.Structure):
    _fields_ = [("points", C.c_void_p), ("n_points", C.c_uint32), ("tets", C.c_void_p), ("n_tets", C.c_uint32), ("material", C.c_double * 5),
                ("excite", C.c_void_p), ("n_excite", C.c_uint32), ("num_modes", C.c_uint32), ("num_fem_modes", C.c_uint32)]


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        import os
        from . import bank
        bank.lib()  # libmodalhost.so (and libmodalhip.so under it) first: libmodalbatch.so depends on both
        so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmodalbatch.so")
        if not os.path.exists(so):
            raise RuntimeError("libmodalbatch.so is missing: run `make -C mesheditor_amd/cpp` (or __graft_entry__.build())")
        _LIB = C.CDLL(so)
    L = _LIB
    if not getattr(L, "_batch_bound", False):
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
        L.mhx_batch_last_error.restype, L.mhx_batch_last_error.argtypes = C.c_char_p, []
        L.mhx_batch_make_id.restype, L.mhx_batch_make_id.argtypes = None, [vp]
        L.mhx_batch_comm_create.restype, L.mhx_batch_comm_create.argtypes = vp, [i32, i32, i32, vp]
        L.mhx_batch_comm_destroy.restype, L.mhx_batch_comm_destroy.argtypes = None, [vp]
        L.mhx_batch_record_length.restype, L.mhx_batch_record_length.argtypes = C.c_uint64, [u32, u32]
        L.mhx_solve_batch.restype, L.mhx_solve_batch.argtypes = i32, [vp, C.POINTER(_Item), u32, u32, u32, u32, vp]
        L._batch_bound = True
    return L


def make_id():
    """ncclGetUniqueId (rank 0): 128 bytes to hand to every rank."""
    buf = (C.c_ubyte * 128)()
    _lib().mhx_batch_make_id(buf)
    raw = bytes(buf)
    if not any(raw):
        raise RuntimeError("ncclGetUniqueId failed: " + _lib().mhx_batch_last_error().decode())
    return raw


class BatchComm:
    """The RCCL communicator of the ranks that share a batch (ncclCommInitRank; a world of one works as well)."""

    def __init__(self, world, rank, device, unique_id):
        self.L = _lib()
        self.world, self.rank = world, rank
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        self.h = self.L.mhx_batch_comm_create(world, rank, device, buf)
        if not self.h:
            raise RuntimeError("SolveBatch communicator: " + self.L.mhx_batch_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.mhx_batch_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def solve_batch(comm, meshes, nev_max=256, pos_max=16, threads=3, excite=None):
    """meshes: list of (points, tets, material tuple, config kwargs) -- every rank passes the same list.  Returns the unpacked
    records of ALL meshes (sharding.unpack_record dictionaries, ordered by index); raises after the gather if any failed."""
    L = _lib()
    keep, items = [], (_Item * len(meshes))()
    for i, m in enumerate(meshes):
        pts = np.ascontiguousarray(m[0], np.float64)
        tets = np.ascontiguousarray(m[1], np.uint32)
        ex = np.ascontiguousarray(excite[i] if excite is not None else pts[(np.arange(10) * len(pts)) // 10], np.float32)
        keep += [pts, tets, ex]
        it = items[i]
        it.points, it.n_points, it.tets, it.n_tets = pts.ctypes.data, len(pts), tets.ctypes.data, len(tets)
        for k in range(5):
            it.material[k] = float(m[2][k]) if k < len(m[2]) else 0.0
        it.excite, it.n_excite = ex.ctypes.data, len(ex)
        it.num_modes, it.num_fem_modes = int(m[3].get("num_modes", 30)), int(m[3].get("num_fem_modes", 45))
    reclen = int(L.mhx_batch_record_length(nev_max, pos_max))
    assert reclen == sharding.record_length(nev_max, pos_max), "record layouts of batch.cpp and sharding.py differ"
    out = np.zeros((len(meshes), reclen))
    if L.mhx_solve_batch(comm.h, items, len(meshes), threads, nev_max, pos_max, out.ctypes.data_as(C.c_void_p)) != 0:
        raise RuntimeError("SolveBatch: " + L.mhx_batch_last_error().decode())
    records = [sharding.unpack_record(out[i], nev_max, pos_max) for i in range(len(meshes))]
    failed = [r["index"] for r in records if not r["ok"]]
    if failed:
        raise RuntimeError(f"SolveBatch: mesh(es) {failed} failed (every rank holds the same list)")
    return records
