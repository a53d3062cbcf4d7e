"""ctypes view of oracle/_ref/libmuscle_ref.so (the compiled reference) — TEST INFRASTRUCTURE ONLY.
Present only where /root/reference was available when oracle/build_ref.sh ran."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libmuscle_ref.so")
MUSCLE = os.path.join(ROOT, "oracle", "_ref", "muscle")
u8p = C.POINTER(C.c_ubyte)
f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint)


def available():
    return os.path.exists(LIB)


_lib = None
_hmm_key = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB)
        L.ref_min_sparse_score.restype = C.c_float
        L.ref_total.restype = C.c_float
        L.ref_aln_score.restype = C.c_float
        L.ref_calc_aln.restype = C.c_float
        L.ref_sparse_from_post.restype = C.c_uint
        L.ref_mpc_ea.restype = C.c_float
        L.ref_mpc_nnz.restype = C.c_uint
        L.ref_mpc_pair_count.restype = C.c_uint
        _lib = L
    return _lib


def init_hmm(nucleo=False, perturb_seed=0):
    global _hmm_key
    if _hmm_key != (nucleo, perturb_seed):
        lib().ref_init_hmm(int(nucleo), perturb_seed)
        _hmm_key = (nucleo, perturb_seed)


def get_hmm():
    start = np.empty(5, np.float32)
    trans = np.empty(25, np.float32)
    match = np.empty(65536, np.float32)
    ins = np.empty(256, np.float32)
    lib().ref_get_hmm(start.ctypes.data_as(f32p), trans.ctypes.data_as(f32p),
                      match.ctypes.data_as(f32p), ins.ctypes.data_as(f32p))
    return start, trans, match, ins


def _seq(s):
    if isinstance(s, str):
        s = s.encode()
    return np.frombuffer(s, dtype=np.uint8).copy()


def fwd(x, y):
    x, y = _seq(x), _seq(y)
    F = np.empty(5 * (len(x) + 1) * (len(y) + 1), np.float32)
    lib().ref_fwd(x.ctypes.data_as(u8p), len(x), y.ctypes.data_as(u8p), len(y), F.ctypes.data_as(f32p))
    return F


def bwd(x, y):
    x, y = _seq(x), _seq(y)
    B = np.empty(5 * (len(x) + 1) * (len(y) + 1), np.float32)
    lib().ref_bwd(x.ctypes.data_as(u8p), len(x), y.ctypes.data_as(u8p), len(y), B.ctypes.data_as(f32p))
    return B


def total(F, B, LX, LY):
    return lib().ref_total(F.ctypes.data_as(f32p), B.ctypes.data_as(f32p), LX, LY)


def post(F, B, LX, LY):
    P = np.empty(max(LX * LY, 1), np.float32)
    lib().ref_post(F.ctypes.data_as(f32p), B.ctypes.data_as(f32p), LX, LY, P.ctypes.data_as(f32p))
    return P[:LX * LY].reshape(LX, LY)


def sparse_from_post(P):
    LX, LY = P.shape
    P = np.ascontiguousarray(P, np.float32)
    off = np.empty(LX + 1, np.uint32)
    val = np.empty(max(LX * LY, 1) * 2, np.uint32)
    n = lib().ref_sparse_from_post(P.ctypes.data_as(f32p), LX, LY, off.ctypes.data_as(u32p), val.ctypes.data_as(u8p))
    return off, val[:2 * n].copy()


def aln_score(P):
    LX, LY = P.shape
    P = np.ascontiguousarray(P, np.float32)
    return lib().ref_aln_score(P.ctypes.data_as(f32p), LX, LY)


def calc_aln(P):
    LX, LY = P.shape
    P = np.ascontiguousarray(P, np.float32)
    buf = C.create_string_buffer(LX + LY + 8)
    n = C.c_uint(0)
    s = lib().ref_calc_aln(P.ctypes.data_as(f32p), LX, LY, buf, C.byref(n))
    return s, buf.raw[:n.value].decode()


def mpc_run(seqs, iters=2, threads=0, nucleo=False, perturb_seed=0):
    """Whole stage through the reference's MPCFlat (ONE call per process). Returns
    stages[0..iters] = list over pairs of (offsets, values-u32), and ea[npairs]."""
    init_hmm(nucleo, perturb_seed)
    L = lib()
    arr = (C.c_char_p * len(seqs))(*[s.encode() if isinstance(s, str) else s for s in seqs])
    rc = L.ref_mpc_begin(len(seqs), arr, threads)
    if rc != 0:
        raise RuntimeError("ref_mpc_begin may only be called once per process")
    lens = [len(s) for s in seqs]
    pairs = [(i, j) for i in range(len(seqs)) for j in range(i + 1, len(seqs))]

    def snap():
        out = []
        for k, (i, j) in enumerate(pairs):
            nnz = L.ref_mpc_nnz(k)
            off = np.empty(lens[i] + 1, np.uint32)
            val = np.empty(max(nnz, 1) * 2, np.uint32)
            L.ref_mpc_sparse(k, off.ctypes.data_as(u32p), val.ctypes.data_as(u8p))
            out.append((off, val[:2 * nnz].copy()))
        return out

    L.ref_mpc_calc_posteriors()
    ea = np.array([L.ref_mpc_ea(i, j) for (i, j) in pairs], np.float32)
    stages = [snap()]
    if len(seqs) >= 3:  # mpcflat.cpp:176
        for it in range(iters):
            L.ref_mpc_cons_iter(it)
            stages.append(snap())
    return stages, ea


def mpc_run_mega(path, iters=2, threads=0):
    """The same through a .mega input (structure profiles): Mega::FromFile + MPCFlat on its chains; CalcPost
    takes the profile branch (calcpost.cpp:14-22). ONE call per process. Returns (tables, stages, ea, probes):
    tables = the reference's own parsed Mega statics (what a caller hands to mpcgpu_set_mega / the oracle)."""
    init_hmm(False, 0)
    L = lib()
    L.ref_mega_weight.restype = C.c_float
    L.ref_mega_ins.restype = C.c_float
    L.ref_mega_match.restype = C.c_float
    rc = L.ref_mpc_begin_mega(path.encode(), threads)
    if rc != 0:
        raise RuntimeError("ref_mpc_begin_mega may only be called once per process")
    F, n = L.ref_mega_feature_count(), L.ref_mega_profile_count()
    alpha = np.array([L.ref_mega_alpha(f) for f in range(F)], np.uint32)
    weight = np.array([L.ref_mega_weight(f) for f in range(F)], np.float32)
    lp, mx = [], []
    for f in range(F):
        a = np.empty(int(alpha[f]), np.float32)
        L.ref_mega_logprobs(f, a.ctypes.data_as(f32p))
        m = np.empty(int(alpha[f]) ** 2, np.float32)
        L.ref_mega_logprobmx(f, m.ctypes.data_as(f32p))
        lp.append(a)
        mx.append(m)
    lens = [L.ref_mega_length(i) for i in range(n)]
    profs, seqs = [], []
    for i in range(n):
        p = np.empty(lens[i] * F, np.uint8)
        L.ref_mega_profile(i, p.ctypes.data_as(u8p))
        profs.append(p)
        buf = C.create_string_buffer(lens[i] + 1)
        L.ref_mega_seq(i, buf)
        seqs.append(buf.raw[:lens[i]].decode())
    tables = {"alpha": alpha, "weight": weight, "lp": np.concatenate(lp), "mx": np.concatenate(mx),
              "profs": profs, "seqs": seqs}
    # direct probes of the reference's emission functions and DP on the first pair (pins the oracle's pieces)
    probes = {}
    if n >= 2:
        l0, l1 = lens[0], lens[1]
        probes["ins0"] = np.array([L.ref_mega_ins(0, p) for p in range(l0)], np.float32)
        probes["match01"] = np.array([[L.ref_mega_match(0, a, 1, b) for b in range(l1)] for a in range(l0)], np.float32)
        Fw = np.empty(5 * (l0 + 1) * (l1 + 1), np.float32)
        Bw = np.empty(5 * (l0 + 1) * (l1 + 1), np.float32)
        L.ref_mega_fwd(0, 1, Fw.ctypes.data_as(f32p))
        L.ref_mega_bwd(0, 1, Bw.ctypes.data_as(f32p))
        probes["F01"], probes["B01"] = Fw, Bw
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]

    def snap():
        out = []
        for k, (i, j) in enumerate(pairs):
            nnz = L.ref_mpc_nnz(k)
            off = np.empty(lens[i] + 1, np.uint32)
            val = np.empty(max(nnz, 1) * 2, np.uint32)
            L.ref_mpc_sparse(k, off.ctypes.data_as(u32p), val.ctypes.data_as(u8p))
            out.append((off, val[:2 * nnz].copy()))
        return out

    L.ref_mpc_calc_posteriors()
    ea = np.array([L.ref_mpc_ea(i, j) for (i, j) in pairs], np.float32)
    stages = [snap()]
    if n >= 3:
        for it in range(iters):
            L.ref_mpc_cons_iter(it)
            stages.append(snap())
    return tables, stages, ea, probes
