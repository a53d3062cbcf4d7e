"""ctypes binding of the C ABI in include/mpcgpu.h (muscle_amd/csrc/libmpcgpu.so).

Plumbing only. There is no CPU fallback: if the HIP library is missing or no GPU is present the
calls raise. (`lib_path` exists so the test-suite can point the same binding at the SIMT-emulator
build of the same sources, tests/emu/libmpcgpu_emu.so, to check kernel logic without a GPU.)
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "csrc", "libmpcgpu.so")
NKERNELS = 9
KERNEL_FAMILIES = ["fb", "post", "store_build", "relax", "commit", "buildpost_gen", "buildpost_sort", "buildpost_reduce", "calc_aln"]

SYMBOLS = [
    "mpcgpu_create", "mpcgpu_destroy", "mpcgpu_last_error", "mpcgpu_version", "mpcgpu_set_hmm",
    "mpcgpu_set_seqs", "mpcgpu_set_mega", "mpcgpu_pair_count", "mpcgpu_calc_posteriors", "mpcgpu_build_store",
    "mpcgpu_shard_info", "mpcgpu_shard_export", "mpcgpu_store_import", "mpcgpu_values_info", "mpcgpu_values_slice", "mpcgpu_values_export", "mpcgpu_values_import",
    "mpcgpu_cons_iter", "mpcgpu_cons_commit", "mpcgpu_cons_commit_range", "mpcgpu_get_ea", "mpcgpu_get_nnz", "mpcgpu_get_sparse",
    "mpcgpu_get_sparse_range", "mpcgpu_post_scores", "mpcgpu_calc_aln", "mpcgpu_align_alns", "mpcgpu_align_alns_w", "mpcgpu_build_post", "mpcgpu_get_last_post", "mpcgpu_align_msas", "mpcgpu_align_pairs", "mpcgpu_get_list_sparse", "mpcgpu_stage_a_info", "mpcgpu_set_seqs_registry", "mpcgpu_timers_reset", "mpcgpu_timers_enable", "mpcgpu_timers_get",
    "mpcgpu_work_get", "mpcgpu_synchronize", "mpcgpu_relax_info", "mpcgpu_shard_entries",
    "mpcgpu_set_pair_order", "mpcgpu_pair_position", "mpcgpu_plan_partition", "mpcgpu_store_import_part", "mpcgpu_store_complete", "mpcgpu_store_info", "mpcgpu_align_alns_batch",
    "mpcgpu_group_create", "mpcgpu_group_destroy", "mpcgpu_group_last_error", "mpcgpu_group_size", "mpcgpu_group_ctx",
    "mpcgpu_group_transport", "mpcgpu_group_set_hmm", "mpcgpu_group_set_seqs", "mpcgpu_group_set_mega",
    "mpcgpu_group_calc_posteriors", "mpcgpu_group_cons_iter",
]


class MpcGpuError(RuntimeError):
    pass


_libs = {}


def load(lib_path=None):
    path = lib_path or os.environ.get("MPCGPU_LIB") or DEFAULT_LIB  # MPCGPU_LIB: another build of the same library (A/B runs)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise MpcGpuError("HIP library not built: %s (run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C muscle_amd/csrc`); there is no CPU fallback" % path)
    L = C.CDLL(path)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    L.mpcgpu_create.argtypes = [C.POINTER(vp), i32]
    L.mpcgpu_destroy.argtypes = [vp]
    L.mpcgpu_destroy.restype = None
    L.mpcgpu_last_error.argtypes = [vp]
    L.mpcgpu_last_error.restype = C.c_char_p
    L.mpcgpu_version.restype = C.c_char_p
    L.mpcgpu_set_hmm.argtypes = [vp, vp, vp, vp, vp, C.c_float, i32]
    L.mpcgpu_set_seqs.argtypes = [vp, u32, vp, vp]
    L.mpcgpu_set_mega.argtypes = [vp, u32, vp, vp, vp, vp, vp]
    L.mpcgpu_pair_count.argtypes = [vp]
    L.mpcgpu_pair_count.restype = u64
    L.mpcgpu_calc_posteriors.argtypes = [vp, u64, u64]
    L.mpcgpu_build_store.argtypes = [vp]
    L.mpcgpu_shard_info.argtypes = [vp, C.POINTER(u64), C.POINTER(vp)]
    L.mpcgpu_shard_entries.argtypes = [vp, C.POINTER(u64)]
    L.mpcgpu_shard_export.argtypes = [vp, vp]
    L.mpcgpu_values_export.argtypes = [vp, u64, u64, vp]
    L.mpcgpu_values_import.argtypes = [vp, u64, u64, vp]
    L.mpcgpu_store_import.argtypes = [vp, u32, vp, vp, vp, vp]
    L.mpcgpu_values_info.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.mpcgpu_set_pair_order.argtypes = [vp, u32, vp]
    L.mpcgpu_pair_position.argtypes = [vp, u32, u32, C.POINTER(u64)]
    L.mpcgpu_plan_partition.argtypes = [u32, vp, u32, u32, vp, C.POINTER(u32), vp]
    L.mpcgpu_store_import_part.argtypes = [vp, u32, vp, vp, vp, vp, vp, u64, u64]
    L.mpcgpu_store_complete.argtypes = [vp]
    L.mpcgpu_store_info.argtypes = [vp, vp]
    L.mpcgpu_align_alns_batch.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp, u32, vp, vp, vp]
    L.mpcgpu_values_slice.argtypes = [vp, u64, u64, C.POINTER(u64), C.POINTER(u64)]
    L.mpcgpu_cons_iter.argtypes = [vp, u64, u64]
    L.mpcgpu_cons_commit.argtypes = [vp]
    L.mpcgpu_cons_commit_range.argtypes = [vp, u64, u64]
    L.mpcgpu_get_ea.argtypes = [vp, u64, u64, vp]
    L.mpcgpu_get_nnz.argtypes = [vp, u64, u64, vp]
    L.mpcgpu_get_sparse.argtypes = [vp, u64, vp, vp]
    L.mpcgpu_get_sparse_range.argtypes = [vp, u64, u64, vp, vp]
    L.mpcgpu_post_scores.argtypes = [vp, u32, u32, u32, vp, vp, vp, i32, u32, C.POINTER(C.c_float), C.POINTER(u32), vp, vp]
    L.mpcgpu_calc_aln.argtypes = [vp, vp, u32, u32, vp, C.POINTER(u32), C.POINTER(C.c_float)]
    L.mpcgpu_align_alns.argtypes = [vp, u32, vp, u32, vp, u32, u32, vp, vp, vp, C.POINTER(u32), C.POINTER(C.c_float)]
    L.mpcgpu_align_alns_w.argtypes = [vp, u32, vp, u32, vp, u32, u32, vp, vp, vp, vp, vp, C.POINTER(u32), C.POINTER(C.c_float)]
    L.mpcgpu_build_post.argtypes = [vp, u32, vp, u32, vp, u32, u32, vp, vp, vp, vp, vp]
    L.mpcgpu_get_last_post.argtypes = [vp, u32, u32, vp]
    L.mpcgpu_align_pairs.argtypes = [vp, u32, vp, vp, u32, vp, vp, vp, vp]
    L.mpcgpu_get_list_sparse.argtypes = [vp, u32, C.POINTER(u32), vp, vp]
    L.mpcgpu_set_seqs_registry.argtypes = [vp, u32, vp, vp]
    L.mpcgpu_align_msas.argtypes = [vp, u32, vp, vp, u32, u32, vp, vp, vp, C.POINTER(u32), C.POINTER(C.c_float), vp]
    L.mpcgpu_timers_reset.argtypes = [vp]
    L.mpcgpu_timers_enable.argtypes = [vp, C.c_int]
    L.mpcgpu_timers_get.argtypes = [vp, vp, vp]
    L.mpcgpu_work_get.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.mpcgpu_stage_a_info.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.mpcgpu_synchronize.argtypes = [vp]
    L.mpcgpu_relax_info.argtypes = [vp, C.c_char_p, u32, C.POINTER(i32)]
    L.mpcgpu_group_create.argtypes = [C.POINTER(vp), u32, vp]
    L.mpcgpu_group_destroy.argtypes = [vp]
    L.mpcgpu_group_destroy.restype = None
    L.mpcgpu_group_last_error.argtypes = [vp]
    L.mpcgpu_group_last_error.restype = C.c_char_p
    L.mpcgpu_group_size.argtypes = [vp]
    L.mpcgpu_group_size.restype = u32
    L.mpcgpu_group_ctx.argtypes = [vp, u32]
    L.mpcgpu_group_ctx.restype = vp
    L.mpcgpu_group_transport.argtypes = [vp]
    L.mpcgpu_group_transport.restype = C.c_char_p
    L.mpcgpu_group_set_hmm.argtypes = [vp, vp, vp, vp, vp, C.c_float, i32]
    L.mpcgpu_group_set_seqs.argtypes = [vp, u32, vp, vp]
    L.mpcgpu_group_set_mega.argtypes = [vp, u32, vp, vp, vp, vp, vp]
    L.mpcgpu_group_calc_posteriors.argtypes = [vp]
    L.mpcgpu_group_cons_iter.argtypes = [vp]
    _libs[path] = L
    return L


def plan_partition(lens, world, lib_path=None, L=None):
    """The block partition of the all-pairs schedule (include/mpcgpu.h: mpcgpu_plan_partition; host only, no device):
    -> (rects (nrects, 4) uint32 — empty: InitPairs order —, rank_pos list of world + 1 positions)."""
    L = L or load(lib_path)
    lens = np.ascontiguousarray(lens, np.uint32)
    cap = world * (world // 2 + 3) + 4
    rects = np.zeros((cap, 4), np.uint32)
    nr = C.c_uint32()
    pos = np.zeros(world + 1, np.uint64)
    rc = L.mpcgpu_plan_partition(len(lens), lens.ctypes.data, world, cap, rects.ctypes.data, C.byref(nr), pos.ctypes.data)
    if rc != 0:
        raise MpcGpuError("mpcgpu_plan_partition failed (%d)" % rc)
    return rects[:nr.value].copy(), [int(x) for x in pos]


class MpcGroup:
    """Several GPUs of one node inside one process (include/mpcgpu.h, mpcgpu_group_*): what the drop-in binary uses with
    MUSCLE_GPU_DEVICES. `devices` may repeat an ordinal (tests on a one-GPU box). ctx(r) is a non-owning MpcGpu view of
    rank r's context; rank 0 is the one a host reads results from."""

    def __init__(self, devices, lib_path=None):
        self.L = load(lib_path)
        self.lib_path = lib_path
        devs = np.asarray(devices, np.int32)
        h = C.c_void_p()
        if self.L.mpcgpu_group_create(C.byref(h), len(devs), devs.ctypes.data) != 0:
            raise MpcGpuError(self.L.mpcgpu_group_last_error(None).decode())
        self.h = h
        self.n = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.mpcgpu_group_destroy(self.h)
            self.h = None

    __del__ = close

    def _ck(self, rc):
        if rc != 0:
            raise MpcGpuError(self.L.mpcgpu_group_last_error(self.h).decode())

    @property
    def size(self):
        return int(self.L.mpcgpu_group_size(self.h))

    def transport(self):
        return self.L.mpcgpu_group_transport(self.h).decode()

    def set_hmm(self, start, trans, match, ins, min_sparse_score, expf_variant=-1):
        a = [np.ascontiguousarray(x, np.float32) for x in (start, trans, match, ins)]
        self._ck(self.L.mpcgpu_group_set_hmm(self.h, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data,
                                              float(min_sparse_score), expf_variant))

    def set_seqs(self, seqs):
        bufs = [np.frombuffer(s.encode() if isinstance(s, str) else bytes(s), np.uint8).copy() for s in seqs]
        ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
        lens = np.array([len(b) for b in bufs], np.uint32)
        self._ck(self.L.mpcgpu_group_set_seqs(self.h, len(bufs), ptrs, lens.ctypes.data))
        self.n = len(bufs)
        self.lens = lens

    def calc_posteriors(self):
        self._ck(self.L.mpcgpu_group_calc_posteriors(self.h))

    def cons_iter(self):
        self._ck(self.L.mpcgpu_group_cons_iter(self.h))

    def ctx(self, rank):
        g = MpcGpu.__new__(MpcGpu)
        g.L = self.L
        g._owned = False  # close()/__del__ of the view must not destroy the group's context
        g.h = C.c_void_p(self.L.mpcgpu_group_ctx(self.h, rank))
        g.n = self.n
        g.lens = self.lens
        g.pairs = [(i, j) for i in range(self.n) for j in range(i + 1, self.n)] if self.n <= 4096 else None
        return g


class MpcGpu:
    """One context = one GPU = one MPCFlat run (mirrors the members of class MPCFlat that the
    hot path touches: InitSeqs/InitPairs, CalcPosteriors, ConsIter, m_DistMx, m_SparsePosts)."""

    def __init__(self, device=0, lib_path=None):
        self.L = load(lib_path)
        h = C.c_void_p()
        if self.L.mpcgpu_create(C.byref(h), device) != 0:
            raise MpcGpuError(self.L.mpcgpu_last_error(None).decode())
        self.h = h
        self.n = 0
        self.lens = None

    def close(self):
        if getattr(self, "h", None):
            if getattr(self, "_owned", True):
                self.L.mpcgpu_destroy(self.h)
            self.h = None

    __del__ = close

    def _ck(self, rc):
        if rc != 0:
            raise MpcGpuError(self.L.mpcgpu_last_error(self.h).decode())

    def version(self):
        return self.L.mpcgpu_version().decode()

    def set_hmm(self, start, trans, match, ins, min_sparse_score, expf_variant=-1):
        a = [np.ascontiguousarray(x, np.float32) for x in (start, trans, match, ins)]
        assert a[0].size == 5 and a[1].size == 25 and a[2].size == 65536 and a[3].size == 256
        self._ck(self.L.mpcgpu_set_hmm(self.h, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data,
                                        a[3].ctypes.data, float(min_sparse_score), expf_variant))

    def set_seqs_registry(self, seqs):
        """all sequences any explicit pair list may refer to (no all-pairs tables)"""
        bufs = [np.frombuffer(s.encode() if isinstance(s, str) else bytes(s), np.uint8).copy() for s in seqs]
        ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
        lens = np.array([len(b) for b in bufs], np.uint32)
        self._ck(self.L.mpcgpu_set_seqs_registry(self.h, len(bufs), ptrs, lens.ctypes.data))
        self.n = len(bufs)
        self.lens = lens
        self.pairs = None

    def align_msas(self, seq1, seq2, p2c1, p2c2, C1, C2):
        """pair q aligns sequence seq1[q] (row of MSA1) with seq2[q] (row of MSA2); p2c1[q] / p2c2[q] are
        their position->column maps -> (path, score, EA per pair)"""
        s1, s2 = np.asarray(seq1, np.uint32), np.asarray(seq2, np.uint32)
        m1 = np.concatenate([np.asarray(x, np.uint32) for x in p2c1]).astype(np.uint32)
        m2 = np.concatenate([np.asarray(x, np.uint32) for x in p2c2]).astype(np.uint32)
        path = np.empty(C1 + C2, np.uint8)
        ea = np.empty(len(s1), np.float32)
        n, sc = C.c_uint32(), C.c_float()
        self._ck(self.L.mpcgpu_align_msas(self.h, len(s1), s1.ctypes.data, s2.ctypes.data, C1, C2, m1.ctypes.data,
                                           m2.ctypes.data, path.ctypes.data, C.byref(n), C.byref(sc), ea.ctypes.data))
        return path[:n.value].tobytes().decode(), float(np.float32(sc.value)), ea

    def set_seqs(self, seqs):
        bufs = [np.frombuffer(s.encode() if isinstance(s, str) else bytes(s), np.uint8).copy() for s in seqs]
        ptrs = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
        lens = np.array([len(b) for b in bufs], np.uint32)
        self._ck(self.L.mpcgpu_set_seqs(self.h, len(bufs), ptrs, lens.ctypes.data))
        self.n = len(bufs)
        self.lens = lens
        self.pairs = [(i, j) for i in range(self.n) for j in range(i + 1, self.n)] if self.n <= 4096 else None

    def set_mega(self, alpha, weight, lp, mx, profiles):
        """Structure-profile emissions for stage A (call after set_seqs / set_seqs_registry): alpha[F], weight[F],
        lp = per-feature log-probabilities back to back, mx = per-feature A x A log-probability matrices back to
        back, profiles[i] = len[i] x F letters (position-major). alpha=None switches back to byte sequences."""
        if alpha is None:
            self._ck(self.L.mpcgpu_set_mega(self.h, 0, None, None, None, None, None))
            return
        alpha = np.ascontiguousarray(alpha, np.uint32)
        weight = np.ascontiguousarray(weight, np.float32)
        lp, mx = np.ascontiguousarray(lp, np.float32), np.ascontiguousarray(mx, np.float32)
        F = len(alpha)
        lp_parts, mx_parts, a, b = [], [], 0, 0
        for f in range(F):
            A = int(alpha[f])
            lp_parts.append(lp[a:a + A].copy())
            mx_parts.append(mx[b:b + A * A].copy())
            a += A
            b += A * A
        profs = [np.ascontiguousarray(p, np.uint8) for p in profiles]
        assert len(profs) == self.n and all(len(p) == int(self.lens[i]) * F for i, p in enumerate(profs))
        lpp = (C.c_void_p * F)(*[x.ctypes.data for x in lp_parts])
        mxp = (C.c_void_p * F)(*[x.ctypes.data for x in mx_parts])
        pp = (C.c_void_p * self.n)(*[x.ctypes.data for x in profs])
        self._ck(self.L.mpcgpu_set_mega(self.h, F, alpha.ctypes.data, weight.ctypes.data, lpp, mxp, pp))

    @property
    def npairs(self):
        return int(self.L.mpcgpu_pair_count(self.h))

    def calc_posteriors(self, k0=0, k1=None):
        self._ck(self.L.mpcgpu_calc_posteriors(self.h, k0, self.npairs if k1 is None else k1))

    def build_store(self):
        self._ck(self.L.mpcgpu_build_store(self.h))

    def shard_info(self):
        b, p = C.c_uint64(), C.c_void_p()
        self._ck(self.L.mpcgpu_shard_info(self.h, C.byref(b), C.byref(p)))
        return b.value, p.value

    def shard_entries(self):
        """stored cells of this context's shard (known right after stage A)"""
        e = C.c_uint64()
        self._ck(self.L.mpcgpu_shard_entries(self.h, C.byref(e)))
        return e.value

    def shard_export(self, dev_ptr):
        self._ck(self.L.mpcgpu_shard_export(self.h, dev_ptr))

    def values_export(self, first, count, dev_ptr):
        self._ck(self.L.mpcgpu_values_export(self.h, first, count, dev_ptr))

    def values_import(self, first, count, dev_ptr):
        self._ck(self.L.mpcgpu_values_import(self.h, first, count, dev_ptr))

    def store_import(self, k0s, k1s, nbytes, dev_ptr):
        a, b, c = (np.ascontiguousarray(x, np.uint64) for x in (k0s, k1s, nbytes))
        self._ck(self.L.mpcgpu_store_import(self.h, len(a), a.ctypes.data, b.ctypes.data, c.ctypes.data, dev_ptr))

    def store_import_part(self, k0s, k1s, nbytes, offsets, dev_ptr, own_k0, own_k1):
        """shards anywhere in the buffer (offsets; None: back to back), any order; a PARTIAL store for the positions
        [own_k0, own_k1) this context will relax (include/mpcgpu.h: mpcgpu_store_import_part)"""
        a, b, c = (np.ascontiguousarray(x, np.uint64) for x in (k0s, k1s, nbytes))
        o = None if offsets is None else np.ascontiguousarray(offsets, np.uint64)
        self._ck(self.L.mpcgpu_store_import_part(self.h, len(a), a.ctypes.data, b.ctypes.data, c.ctypes.data,
                                                 None if o is None else o.ctypes.data, dev_ptr, own_k0, own_k1))

    def store_info(self):
        """sizes of the current store (include/mpcgpu.h: mpcgpu_store_info)"""
        o = np.zeros(6, np.uint64)
        self._ck(self.L.mpcgpu_store_info(self.h, o.ctypes.data))
        return dict(zip(("record_bytes", "window_bytes", "packed_bytes", "entries", "own_entries", "sequences_held"), (int(x) for x in o)))

    def store_complete(self):
        self._ck(self.L.mpcgpu_store_complete(self.h))

    def set_pair_order(self, rects):
        """rects: (nrects, 4) array of {xa, xb, ya, yb}, or None / empty for InitPairs order (include/mpcgpu.h: mpcgpu_set_pair_order)"""
        r = np.zeros((0, 4), np.uint32) if rects is None else np.ascontiguousarray(rects, np.uint32).reshape(-1, 4)
        self._ck(self.L.mpcgpu_set_pair_order(self.h, len(r), r.ctypes.data if len(r) else None))

    def plan_partition(self, lens, world):
        return plan_partition(lens, world, L=self.L)

    def pair_position(self, x, y):
        p = C.c_uint64()
        self._ck(self.L.mpcgpu_pair_position(self.h, x, y, C.byref(p)))
        return p.value

    def values_info(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._ck(self.L.mpcgpu_values_info(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def values_slice(self, k0, k1):
        f, n = C.c_uint64(), C.c_uint64()
        self._ck(self.L.mpcgpu_values_slice(self.h, k0, k1, C.byref(f), C.byref(n)))
        return f.value, n.value

    def cons_iter(self, k0=0, k1=None):
        self._ck(self.L.mpcgpu_cons_iter(self.h, k0, self.npairs if k1 is None else k1))

    def cons_commit(self):
        self._ck(self.L.mpcgpu_cons_commit(self.h))

    def cons_commit_range(self, first, count):
        self._ck(self.L.mpcgpu_cons_commit_range(self.h, first, count))

    def synchronize(self):
        self._ck(self.L.mpcgpu_synchronize(self.h))

    def get_ea(self, k0=0, k1=None):
        k1 = self.npairs if k1 is None else k1
        out = np.empty(max(k1 - k0, 1), np.float32)
        self._ck(self.L.mpcgpu_get_ea(self.h, k0, k1, out.ctypes.data))
        return out[:k1 - k0]

    def get_nnz(self, k0=0, k1=None):
        k1 = self.npairs if k1 is None else k1
        out = np.empty(max(k1 - k0, 1), np.uint32)
        self._ck(self.L.mpcgpu_get_nnz(self.h, k0, k1, out.ctypes.data))
        return out[:k1 - k0]

    def _pair(self, k):
        if self.pairs is not None:
            return self.pairs[k]
        n, i = self.n, 0
        while k >= n - 1 - i:
            k -= n - 1 - i
            i += 1
        return i, i + 1 + k

    def get_sparse_range(self, k0=0, k1=None):
        """-> list of (offsets u32[LX+1], values u32[2*nnz] = interleaved {P bits, col}) per pair"""
        k1 = self.npairs if k1 is None else k1
        nnz = self.get_nnz(k0, k1).astype(np.int64)
        lx = np.array([int(self.lens[self._pair(k)[0]]) + 1 for k in range(k0, k1)], np.int64)
        off = np.empty(max(int(lx.sum()), 1), np.uint32)
        val = np.empty(max(int(nnz.sum()) * 2, 1), np.uint32)
        self._ck(self.L.mpcgpu_get_sparse_range(self.h, k0, k1, off.ctypes.data, val.ctypes.data))
        out, po, pv = [], 0, 0
        for q in range(k1 - k0):
            out.append((off[po:po + lx[q]], val[pv:pv + 2 * nnz[q]]))
            po += lx[q]
            pv += 2 * nnz[q]
        return out

    def post_scores(self, LX, LY, rows, cols, scores, kernel=0, batch=64):
        """finishing kernels on one candidate list -> (ea, offsets, values) (include/mpcgpu.h: mpcgpu_post_scores)"""
        rows, cols = np.ascontiguousarray(rows, np.uint32), np.ascontiguousarray(cols, np.uint32)
        scores = np.ascontiguousarray(scores, np.float32)
        ea, nnz = C.c_float(), C.c_uint32()
        off = np.empty(LX + 1, np.uint32)
        val = np.empty(max(len(rows), 1) * 2, np.uint32)
        self._ck(self.L.mpcgpu_post_scores(self.h, LX, LY, len(rows), rows.ctypes.data, cols.ctypes.data, scores.ctypes.data,
                                           kernel, batch, C.byref(ea), C.byref(nnz), off.ctypes.data, val.ctypes.data))
        return np.float32(ea.value), off, val[:2 * nnz.value].copy()

    def calc_aln(self, post):
        """post: (LX, LY) float32 dense matrix in host memory -> (path str of B/X/Y, score)"""
        post = np.ascontiguousarray(post, np.float32)
        LX, LY = post.shape
        path = np.empty(LX + LY, np.uint8)
        n, sc = C.c_uint32(), C.c_float()
        self._ck(self.L.mpcgpu_calc_aln(self.h, post.ctypes.data, LX, LY, path.ctypes.data, C.byref(n), C.byref(sc)))
        return path[:n.value].tobytes().decode(), float(np.float32(sc.value))

    def align_alns(self, seq1, seq2, p2c1, p2c2, C1, C2, w1=None, w2=None):
        """seq1/seq2: sequence indices of the rows of the two alignments; p2c1/p2c2: list of per-row
        position->column arrays; w1/w2: optional sequence weights of the rows -> (path, score) of BuildPost + CalcAlnFlat
        on the device store"""
        s1, s2 = np.asarray(seq1, np.uint32), np.asarray(seq2, np.uint32)
        m1 = np.concatenate([np.asarray(x, np.uint32) for x in p2c1]).astype(np.uint32)
        m2 = np.concatenate([np.asarray(x, np.uint32) for x in p2c2]).astype(np.uint32)
        path = np.empty(C1 + C2, np.uint8)
        n, sc = C.c_uint32(), C.c_float()
        if w1 is not None:
            a1, a2 = np.ascontiguousarray(w1, np.float32), np.ascontiguousarray(w2, np.float32)
            self._ck(self.L.mpcgpu_align_alns_w(self.h, len(s1), s1.ctypes.data, len(s2), s2.ctypes.data, C1, C2, m1.ctypes.data,
                                                 m2.ctypes.data, a1.ctypes.data, a2.ctypes.data, path.ctypes.data, C.byref(n), C.byref(sc)))
            return path[:n.value].tobytes().decode(), float(np.float32(sc.value))
        self._ck(self.L.mpcgpu_align_alns(self.h, len(s1), s1.ctypes.data, len(s2), s2.ctypes.data, C1, C2,
                                           m1.ctypes.data, m2.ctypes.data, path.ctypes.data, C.byref(n), C.byref(sc)))
        return path[:n.value].tobytes().decode(), float(np.float32(sc.value))

    def align_alns_batch(self, joins):
        """joins: list of (seq1, seq2, p2c1, p2c2, C1, C2) as align_alns takes them -> [(path, score)] (mpcgpu_align_alns_batch:
        independent joins, the small ones in two launches together)"""
        nj = len(joins)
        n1 = np.array([len(j[0]) for j in joins], np.uint32)
        n2 = np.array([len(j[1]) for j in joins], np.uint32)
        C1 = np.array([j[4] for j in joins], np.uint32)
        C2 = np.array([j[5] for j in joins], np.uint32)
        seqs = np.concatenate([np.concatenate([np.asarray(j[0], np.uint32), np.asarray(j[1], np.uint32)]) for j in joins]).astype(np.uint32)
        maps = np.concatenate([np.asarray(x, np.uint32) for j in joins for x in list(j[2]) + list(j[3])]).astype(np.uint32)
        stride = int((C1 + C2).max())
        paths = np.zeros(nj * stride, np.uint8)
        plen = np.zeros(nj, np.uint32)
        sc = np.zeros(nj, np.float32)
        self._ck(self.L.mpcgpu_align_alns_batch(self.h, nj, n1.ctypes.data, n2.ctypes.data, C1.ctypes.data, C2.ctypes.data, seqs.ctypes.data,
                                                 maps.ctypes.data, stride, paths.ctypes.data, plen.ctypes.data, sc.ctypes.data))
        return [(paths[q * stride:q * stride + int(plen[q])].tobytes().decode(), float(sc[q])) for q in range(nj)]

    def align_pairs(self, seq1, seq2, sparse=False):
        """AlignPairFlat for a list of pairs of registered sequences -> [(path, score, ea)] (+ (off, val) per pair with sparse=True)"""
        s1, s2 = np.ascontiguousarray(seq1, np.uint32), np.ascontiguousarray(seq2, np.uint32)
        n = len(s1)
        stride = int(max(int(self.lens[a]) + int(self.lens[b]) for a, b in zip(s1, s2))) if n else 1
        paths = np.zeros(max(n, 1) * stride, np.uint8)
        plen = np.zeros(max(n, 1), np.uint32)
        sc = np.zeros(max(n, 1), np.float32)
        ea = np.zeros(max(n, 1), np.float32)
        self._ck(self.L.mpcgpu_align_pairs(self.h, n, s1.ctypes.data, s2.ctypes.data, stride, paths.ctypes.data, plen.ctypes.data,
                                           sc.ctypes.data, ea.ctypes.data))
        out = [(paths[q * stride:q * stride + int(plen[q])].tobytes().decode(), np.float32(sc[q]), np.float32(ea[q])) for q in range(n)]
        if not sparse:
            return out
        sp = []
        for q in range(n):
            nz = C.c_uint32()
            self._ck(self.L.mpcgpu_get_list_sparse(self.h, q, C.byref(nz), None, None))
            off = np.empty(int(self.lens[s1[q]]) + 1, np.uint32)
            val = np.empty(max(nz.value, 1) * 2, np.uint32)
            self._ck(self.L.mpcgpu_get_list_sparse(self.h, q, C.byref(nz), off.ctypes.data, val.ctypes.data))
            sp.append((off, val[:2 * nz.value].copy()))
        return out, sp

    def build_post(self, seq1, seq2, p2c1, p2c2, C1, C2, w1=None, w2=None):
        """MPCFlat::BuildPost on the device store -> (C1, C2) float32 matrix (include/mpcgpu.h: mpcgpu_build_post)"""
        s1, s2 = np.asarray(seq1, np.uint32), np.asarray(seq2, np.uint32)
        m1 = np.concatenate([np.asarray(x, np.uint32) for x in p2c1]).astype(np.uint32)
        m2 = np.concatenate([np.asarray(x, np.uint32) for x in p2c2]).astype(np.uint32)
        post = np.empty((C1, C2), np.float32)
        a1 = None if w1 is None else np.ascontiguousarray(w1, np.float32)
        a2 = None if w2 is None else np.ascontiguousarray(w2, np.float32)
        self._ck(self.L.mpcgpu_build_post(self.h, len(s1), s1.ctypes.data, len(s2), s2.ctypes.data, C1, C2, m1.ctypes.data, m2.ctypes.data,
                                          None if a1 is None else a1.ctypes.data, None if a2 is None else a2.ctypes.data, post.ctypes.data))
        return post

    def last_post(self, C1, C2):
        """the dense matrix the last alignment / BuildPost call on this context built"""
        post = np.empty((C1, C2), np.float32)
        self._ck(self.L.mpcgpu_get_last_post(self.h, C1, C2, post.ctypes.data))
        return post

    def relax_info(self):
        """-> (description of the store layout / relax geometry in use, is_fallback)"""
        buf = C.create_string_buffer(1024)
        fb = C.c_int(0)
        self._ck(self.L.mpcgpu_relax_info(self.h, buf, 1024, C.byref(fb)))
        return buf.value.decode(), bool(fb.value)

    def stage_a_info(self):
        """(pairs, pairs that ran in chains, chains) of the last stage A"""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._ck(self.L.mpcgpu_stage_a_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def timers_reset(self):
        self._ck(self.L.mpcgpu_timers_reset(self.h))

    def timers_enable(self, on=True):
        """hipEvents around the launches on (default) / off (what the drop-in does unless MUSCLE_GPU_TIMING is set)."""
        self._ck(self.L.mpcgpu_timers_enable(self.h, 1 if on else 0))

    def timers_get(self):
        ms = np.zeros(NKERNELS, np.float32)
        ln = np.zeros(NKERNELS, np.uint64)
        self._ck(self.L.mpcgpu_timers_get(self.h, ms.ctypes.data, ln.ctypes.data))
        return {k: (float(ms[i]), int(ln[i])) for i, k in enumerate(KERNEL_FAMILIES)}

    def work_get(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._ck(self.L.mpcgpu_work_get(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"dp_cells": a.value, "relax_entry_z": b.value, "store_entries": c.value}
