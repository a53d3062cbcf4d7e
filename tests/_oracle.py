"""ctypes view of oracle/libmpc_oracle.so (the CPU restatement) — TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ODIR, "libmpc_oracle.so")

u8p = C.POINTER(C.c_ubyte)
f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint)


class Hmm(C.Structure):
    _fields_ = [("start", C.c_float * 5), ("trans", C.c_float * 25),
                ("match", C.c_float * 65536), ("ins", C.c_float * 256)]


def _build():
    src = os.path.join(ODIR, "mpc_oracle.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ODIR, "libmpc_oracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _build()
        L = C.CDLL(LIB)
        L.orc_log_add.restype = C.c_float
        L.orc_log_add.argtypes = [C.c_float, C.c_float]
        L.orc_total.restype = C.c_float
        L.orc_min_sparse_score.restype = C.c_float
        L.orc_aln_score.restype = C.c_float
        L.orc_ea.restype = C.c_float
        L.orc_ea.argtypes = [C.c_float, C.c_uint, C.c_uint]
        L.orc_calc_aln.restype = C.c_float
        L.orc_expf_emul.restype = C.c_float
        L.orc_expf_emul.argtypes = [C.c_float, C.c_int]
        L.orc_libm_expf.restype = C.c_float
        L.orc_libm_expf.argtypes = [C.c_float]
        L.orc_store_new.restype = C.c_void_p
        L.orc_store_new.argtypes = [C.c_uint, C.c_void_p]
        L.orc_store_free.argtypes = [C.c_void_p]
        L.orc_store_nnz.restype = C.c_uint
        L.orc_store_nnz.argtypes = [C.c_void_p, C.c_uint]
        L.orc_store_get.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
        L.orc_store_set.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
        L.orc_calc_posteriors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_uint, C.c_uint, C.c_int]
        L.orc_cons_iter.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int]
        L.orc_pair_index.restype = C.c_uint
        L.orc_pair_index.argtypes = [C.c_uint, C.c_uint, C.c_uint]
        _lib = L
    return _lib


def make_hmm(start, trans, match, ins):
    h = Hmm()
    C.memmove(h.start, np.ascontiguousarray(start, np.float32).ctypes.data, 20)
    C.memmove(h.trans, np.ascontiguousarray(trans, np.float32).ctypes.data, 100)
    C.memmove(h.match, np.ascontiguousarray(match, np.float32).ctypes.data, 65536 * 4)
    C.memmove(h.ins, np.ascontiguousarray(ins, np.float32).ctypes.data, 1024)
    return h


def _seq(s):
    if isinstance(s, str):
        s = s.encode()
    return np.frombuffer(s, dtype=np.uint8).copy()


def fwd(h, x, y):
    x, y = _seq(x), _seq(y)
    F = np.empty(5 * (len(x) + 1) * (len(y) + 1), np.float32)
    lib().orc_fwd(C.byref(h), x.ctypes.data_as(u8p), len(x), y.ctypes.data_as(u8p), len(y), F.ctypes.data_as(f32p))
    return F


def bwd(h, x, y):
    x, y = _seq(x), _seq(y)
    B = np.empty(5 * (len(x) + 1) * (len(y) + 1), np.float32)
    lib().orc_bwd(C.byref(h), x.ctypes.data_as(u8p), len(x), y.ctypes.data_as(u8p), len(y), B.ctypes.data_as(f32p))
    return B


def total(F, B, LX, LY):
    return lib().orc_total(F.ctypes.data_as(f32p), B.ctypes.data_as(f32p), LX, LY)


def post(F, B, LX, LY):
    P = np.empty(max(LX * LY, 1), np.float32)
    lib().orc_post(F.ctypes.data_as(f32p), B.ctypes.data_as(f32p), LX, LY, P.ctypes.data_as(f32p))
    return P[:LX * LY].reshape(LX, LY)


def sparse_from_post(P):
    LX, LY = P.shape
    P = np.ascontiguousarray(P, np.float32)
    off = np.empty(LX + 1, np.uint32)
    val = np.empty(max(LX * LY, 1) * 2, np.uint32)
    lib().orc_sparse_from_post.restype = C.c_uint
    n = lib().orc_sparse_from_post(P.ctypes.data_as(f32p), LX, LY, off.ctypes.data_as(u32p), val.ctypes.data_as(u8p))
    return off, val[:2 * n].copy()


def aln_score(P):
    LX, LY = P.shape
    P = np.ascontiguousarray(P, np.float32)
    return lib().orc_aln_score(P.ctypes.data_as(f32p), LX, LY)


def calc_aln(P):
    LX, LY = P.shape
    P = np.ascontiguousarray(P, np.float32)
    buf = C.create_string_buffer(LX + LY + 8)
    n = C.c_uint(0)
    s = lib().orc_calc_aln(P.ctypes.data_as(f32p), LX, LY, buf, C.byref(n))
    return s, buf.raw[:n.value].decode()


class Store:
    """All-pairs sparse posterior store of the oracle (one MySparseMx-layout matrix per pair)."""

    def __init__(self, seqs):
        self.seqs = [_seq(s) for s in seqs]
        self.n = len(seqs)
        self.len = np.array([len(s) for s in self.seqs], np.uint32)
        self.npairs = self.n * (self.n - 1) // 2
        self.h = lib().orc_store_new(self.n, self.len.ctypes.data)
        self._ptrs = (C.c_void_p * self.n)(*[s.ctypes.data for s in self.seqs])

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_store_free(self.h)
            self.h = None

    def pairs(self):
        if getattr(self, "_pairs", None) is None:
            self._pairs = [(i, j) for i in range(self.n) for j in range(i + 1, self.n)]
        return self._pairs

    def calc_posteriors(self, hmm, k0=0, k1=None, threads=0):
        ea = np.zeros(max(self.npairs, 1), np.float32)
        lib().orc_calc_posteriors(C.byref(hmm), self.h, self._ptrs, ea.ctypes.data, k0,
                                  self.npairs if k1 is None else k1, threads)
        return ea[:self.npairs]

    def get(self, k):
        i, j = self.pairs()[k]
        nnz = lib().orc_store_nnz(self.h, k)
        off = np.empty(int(self.len[i]) + 1, np.uint32)
        val = np.empty(max(nnz, 1) * 2, np.uint32)
        lib().orc_store_get(self.h, k, off.ctypes.data, val.ctypes.data)
        return off, val[:2 * nnz].copy()

    def set(self, k, off, val):
        off = np.ascontiguousarray(off, np.uint32)
        val = np.ascontiguousarray(val, np.uint32)
        lib().orc_store_set(self.h, k, len(off) - 1, off.ctypes.data, val.ctypes.data)

    def cons_iter(self, k0=0, k1=None, threads=0):
        dst = Store.__new__(Store)
        dst.seqs, dst.n, dst.len, dst.npairs, dst._ptrs = self.seqs, self.n, self.len, self.npairs, self._ptrs
        dst.h = lib().orc_store_new(self.n, self.len.ctypes.data)
        lib().orc_cons_iter(self.h, dst.h, k0, self.npairs if k1 is None else k1, threads)
        return dst


def val_probs(val):
    return val.view(np.float32)[0::2]


def val_cols(val):
    return val[1::2]


# ---- structure-profile ("mega") emissions: orc_mega / orc_*_mega ---------------------------------------------
class Mega(C.Structure):
    _fields_ = [("nfeat", C.c_uint), ("alpha", C.c_void_p), ("weight", C.c_void_p), ("lp", C.c_void_p),
                ("lp_off", C.c_void_p), ("mx", C.c_void_p), ("mx_off", C.c_void_p)]


def make_mega(alpha, weight, lp, mx):
    """alpha[F] u32, weight[F] f32, lp = log-probabilities of every feature back to back, mx = A_f x A_f
    log-probability matrices back to back (the reference's parsed Mega statics). Keeps the arrays alive."""
    g = Mega()
    keep = {"alpha": np.ascontiguousarray(alpha, np.uint32), "weight": np.ascontiguousarray(weight, np.float32),
            "lp": np.ascontiguousarray(lp, np.float32), "mx": np.ascontiguousarray(mx, np.float32)}
    a = keep["alpha"].astype(np.uint64)
    keep["lp_off"] = np.concatenate([[0], np.cumsum(a)[:-1]]).astype(np.uint32)
    keep["mx_off"] = np.concatenate([[0], np.cumsum(a * a)[:-1]]).astype(np.uint32)
    g.nfeat = len(keep["alpha"])
    for k in ("alpha", "weight", "lp", "lp_off", "mx", "mx_off"):
        setattr(g, k, keep[k].ctypes.data)
    g._keep = keep
    return g


def mega_ins(g, prof, pos):
    lib().orc_mega_ins.restype = C.c_float
    return lib().orc_mega_ins(C.byref(g), prof.ctypes.data_as(u8p), pos)


def mega_match(g, px, i, py, j):
    lib().orc_mega_match.restype = C.c_float
    return lib().orc_mega_match(C.byref(g), px.ctypes.data_as(u8p), i, py.ctypes.data_as(u8p), j)


def fwd_mega(h, g, px, py):
    LX, LY = len(px) // g.nfeat, len(py) // g.nfeat
    F = np.empty(5 * (LX + 1) * (LY + 1), np.float32)
    lib().orc_fwd_mega(C.byref(h), C.byref(g), px.ctypes.data_as(u8p), LX, py.ctypes.data_as(u8p), LY, F.ctypes.data_as(f32p))
    return F


def bwd_mega(h, g, px, py):
    LX, LY = len(px) // g.nfeat, len(py) // g.nfeat
    B = np.empty(5 * (LX + 1) * (LY + 1), np.float32)
    lib().orc_bwd_mega(C.byref(h), C.byref(g), px.ctypes.data_as(u8p), LX, py.ctypes.data_as(u8p), LY, B.ctypes.data_as(f32p))
    return B


class MegaStore(Store):
    """Store whose stage A runs on structure profiles (calcpost.cpp:14-22); relax is unchanged."""

    def __init__(self, seqs, profs):
        Store.__init__(self, seqs)
        self.profs = [np.ascontiguousarray(p, np.uint8) for p in profs]
        self._pptrs = (C.c_void_p * self.n)(*[p.ctypes.data for p in self.profs])

    def calc_posteriors_mega(self, hmm, g, k0=0, k1=None, threads=0):
        ea = np.zeros(max(self.npairs, 1), np.float32)
        lib().orc_calc_posteriors_mega.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_uint, C.c_uint, C.c_int]
        lib().orc_calc_posteriors_mega(C.byref(hmm), C.byref(g), self.h, self._pptrs, ea.ctypes.data, k0,
                                       self.npairs if k1 is None else k1, threads)
        return ea[:self.npairs]
