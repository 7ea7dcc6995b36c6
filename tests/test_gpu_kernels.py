"""-m gpu: every HIP kernel of the CRS2_LM path against the CPU oracle's statement of the same
contract (oracle/port_kernels.c), through the kernel-level C-ABI of include/nlopt_amd.h.
Integer/index outputs and everything that feeds x must be bit-exact; objective values within
1e-10 relative (device libm vs glibc, reassociated reduction — SURVEY.md §7.3.9)."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
import nlopt_amd
from nlopt_amd import DevBuf

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def words_from_seed(seed, count, skip=0):
    P = O.port()
    P.orc_srand(seed)
    P.orc_k_words.argtypes = [C.c_uint64, C.c_void_p]
    if skip:
        tmp = np.zeros(skip, dtype=np.uint32)
        P.orc_k_words(skip, tmp.ctypes.data)
    w = np.zeros(count, dtype=np.uint32)
    P.orc_k_words(count, w.ctypes.data)
    return w


def close(a, b, scale=None):
    a, b = np.asarray(a), np.asarray(b)
    s = np.maximum(np.abs(b), np.abs(b).mean() if scale is None else scale)
    return np.all(np.abs(a - b) <= RTOL * s)


@pytest.fixture(scope="module")
def L():
    L = nlopt_amd.lib()
    assert nlopt_amd.device_count() > 0, "no HIP device: the product has no CPU fallback"
    return L


@pytest.mark.parametrize("seed,predraw", [(5489, 0), (42, 1000), (123456789, 623), (7, 624)])
def test_mt_stream_matches_serial_generator(L, seed, predraw):
    """jump-ahead + per-segment generation == the serial generator, at ragged offsets that cross
    segment boundaries and reach segments only obtainable through several doubling rounds"""
    SEG = 624 * 1024
    L.nlopt_srand(seed)
    for _ in range(predraw):
        L.nla_genrand_int32()
    total = 6 * SEG + 12345
    ref = words_from_seed(seed, total, skip=predraw)
    s = L.nla_mtstream_create(None)
    assert s
    try:
        for first, count in [(0, 5000), (SEG - 100, 1000), (3, 2 * SEG + 17), (5 * SEG + 999, SEG + 11346), (4 * SEG - 1, 2)]:
            d = DevBuf(4 * count)
            assert L.nla_mtstream_fill(s, first, count, d.ptr) == 0
            assert L.nla_stream_sync(None) == 0
            got = d.to_array(np.uint32, count)
            assert np.array_equal(got, ref[first:first + count]), (first, count)
            d.free()
        # leaving the host generator where the reference's would be after consuming `used` words
        used = 3 * SEG + 4321
        assert L.nla_mtstream_finish(s, used) == 0
        nxt = [L.nla_genrand_int32() for _ in range(2000)]
        assert nxt == list(ref[used:used + 2000])
    finally:
        L.nla_mtstream_destroy(s)


@pytest.mark.parametrize("obj,n,nrows", [("rastrigin", 10, 99), ("griewank", 257, 300), ("ackley", 512, 1000),
                                         ("rosenbrock", 64, 513), ("levy", 33, 77), ("sphere", 4096, 64)])
def test_init_rows_kernel(L, obj, n, nrows):
    P = O.port()
    ld = (n + 1) & ~1
    lo, hi = nlopt_amd.objective_box(obj)
    lb, ub = np.full(n, lo), np.linspace(hi * 0.5, hi, n)
    w = words_from_seed(99, 2 * n * nrows)
    Xr = np.zeros((nrows, ld))
    P.orc_k_init_rows.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    P.orc_k_init_rows(n, ld, lb.ctypes.data, ub.ctypes.data, w.ctypes.data, nrows, Xr.ctypes.data)
    Fr = np.zeros(nrows)
    P.orc_k_eval.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    P.orc_k_eval(O.OBJ[obj], n, ld, Xr.ctypes.data, nrows, Fr.ctypes.data)
    dlb, dub, dw = DevBuf.from_array(lb), DevBuf.from_array(ub), DevBuf.from_array(w)
    dX, dF = DevBuf(8 * ld * (nrows + 2)), DevBuf(8 * (nrows + 2))
    # rows are written at row_first = 2 to check the row offset arithmetic
    assert L.nla_k_crs_init_rows(O.OBJ[obj], n, ld, dlb.ptr, dub.ptr, dw.ptr, 2, nrows, dX.ptr, dF.ptr, None) == 0
    assert L.nla_stream_sync(None) == 0
    X = dX.to_array(np.float64, ld * (nrows + 2)).reshape(nrows + 2, ld)[2:, :n]
    F = dF.to_array(np.float64, nrows + 2)[2:]
    assert np.array_equal(X, Xr[:, :n])            # bit-exact population
    assert close(F, Fr)
    # the stand-alone evaluator on the same rows
    dF2 = DevBuf(8 * nrows)
    dXr = DevBuf.from_array(Xr)
    assert L.nla_k_eval(O.OBJ[obj], n, ld, dXr.ptr, nrows, dF2.ptr, None) == 0
    assert close(dF2.to_array(np.float64, nrows), Fr)


@pytest.mark.parametrize("n,N,nblocks", [(1, 30, 70), (2, 3, 65), (10, 100, 200), (10, 11, 64), (64, 2000, 129),
                                         (257, 600, 40), (512, 100000, 66), (4096, 20000, 8)])
def test_vitter_kernel_bit_exact(L, n, N, nblocks):
    P = O.port()
    w = words_from_seed(2024, 2 * n * nblocks)
    jr, pr, lr = np.zeros(nblocks, np.int32), np.zeros(nblocks * n, np.int32), np.zeros(nblocks, np.int32)
    P.orc_k_vitter.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    P.orc_k_vitter(n, N, w.ctypes.data, nblocks, jr.ctypes.data, pr.ctypes.data, lr.ctypes.data)
    dw = DevBuf.from_array(w)
    dj, dp, dl = DevBuf(4 * nblocks), DevBuf(4 * nblocks * n), DevBuf(4 * nblocks)
    assert L.nla_k_crs_vitter(n, N, dw.ptr, nblocks, dj.ptr, dp.ptr, dl.ptr, None) == 0
    assert L.nla_stream_sync(None) == 0
    assert np.array_equal(dj.to_array(np.int32, nblocks), jr)
    assert np.array_equal(dp.to_array(np.int32, nblocks * n), pr)
    assert np.array_equal(dl.to_array(np.int32, nblocks), lr)


def _spec_inputs(n, N, K, seed, obj):
    P = O.port()
    ld = (n + 1) & ~1
    rng = np.random.default_rng(seed)
    lo, hi = nlopt_amd.objective_box(obj)
    lb, ub = np.full(n, lo), np.full(n, hi)
    X = np.zeros((N, ld))
    X[:, :n] = rng.uniform(lo, hi, size=(N, n))
    w = words_from_seed(seed, 2 * n * (K + 1))
    jn, pos, last = np.zeros(K, np.int32), np.zeros(K * n, np.int32), np.zeros(K, np.int32)
    P.orc_k_vitter.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    P.orc_k_vitter(n, N, w.ctypes.data, K, jn.ctypes.data, pos.ctypes.data, last.ctypes.data)
    return ld, lb, ub, X, w, jn, pos, last


@pytest.mark.parametrize("obj,n,N,K,i0", [("rastrigin", 10, 100, 7, 0), ("rastrigin", 10, 11, 5, 10), ("griewank", 64, 500, 33, 250),
                                          ("ackley", 257, 600, 9, 599), ("levy", 128, 300, 4, 17), ("rosenbrock", 512, 2000, 6, 3),
                                          ("griewank", 4096, 4200, 3, 4199), ("sphere", 1, 9, 8, 4)])
def test_gather_and_post_kernels(L, obj, n, N, K, i0):
    P = O.port()
    ld, lb, ub, X, w, jn, pos, last = _spec_inputs(n, N, K, 31 + n, obj)
    oid = O.OBJ[obj]
    # oracle
    TXr = np.zeros((K, ld))
    P.orc_k_gather.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p]
    P.orc_k_gather(n, ld, X.ctypes.data, i0, jn.ctypes.data, pos.ctypes.data, last.ctypes.data, K, lb.ctypes.data,
                   ub.ctypes.data, TXr.ctypes.data)
    TMr = np.zeros((K, ld))
    P.orc_k_mutate.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for s in range(K):
        P.orc_k_mutate(n, X[i0].ctypes.data, TXr[s].ctypes.data, w[(s + 1) * 2 * n:].ctypes.data, lb.ctypes.data,
                       ub.ctypes.data, TMr[s].ctypes.data)
    fTr, fMr = np.zeros(K), np.zeros(K)
    P.orc_k_eval.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    P.orc_k_eval(oid, n, ld, TXr.ctypes.data, K, fTr.ctypes.data)
    P.orc_k_eval(oid, n, ld, TMr.ctypes.data, K, fMr.ctypes.data)
    # a worst-list that contains rows the slots did and did not read, the best row, and duplicates of nothing
    rng = np.random.default_rng(5)
    nW = min(N, 40)
    W = rng.permutation(N)[:nW].astype(np.int64)
    W[nW // 2] = i0 if i0 not in W else W[nW // 2]
    mhr = np.zeros(K, np.int32)
    P.orc_k_minhz.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    P.orc_k_minhz(n, i0, pos.ctypes.data, last.ctypes.data, K, W.ctypes.data, nW, mhr.ctypes.data)
    # device
    dX, dlb, dub, dw = DevBuf.from_array(X), DevBuf.from_array(lb), DevBuf.from_array(ub), DevBuf.from_array(w)
    dj, dp, dl, dW = DevBuf.from_array(jn), DevBuf.from_array(pos), DevBuf.from_array(last), DevBuf.from_array(W)
    dTX, dTM = DevBuf(8 * ld * K), DevBuf(8 * ld * K)
    dfT, dfM, dmh = DevBuf(8 * K), DevBuf(8 * K), DevBuf(4 * K)
    assert L.nla_k_crs_gather(n, ld, dX.ptr, i0, dj.ptr, dp.ptr, dl.ptr, K, dlb.ptr, dub.ptr, dTX.ptr, None) == 0
    assert L.nla_k_crs_post(oid, n, ld, dX.ptr, i0, dTX.ptr, dTM.ptr, dw.ptr + 4 * 2 * n, K, dW.ptr, nW, dp.ptr, dl.ptr,
                            dlb.ptr, dub.ptr, dfT.ptr, dfM.ptr, dmh.ptr, None) == 0
    assert L.nla_stream_sync(None) == 0
    TX = dTX.to_array(np.float64, K * ld).reshape(K, ld)[:, :n]
    TM = dTM.to_array(np.float64, K * ld).reshape(K, ld)[:, :n]
    assert np.array_equal(TX, TXr[:, :n])          # bit-exact trial points (row order, no FMA)
    assert np.array_equal(TM, TMr[:, :n])          # bit-exact mutations
    scale = np.abs(np.concatenate([fTr, fMr])).mean()
    assert close(dfT.to_array(np.float64, K), fTr, scale)
    assert close(dfM.to_array(np.float64, K), fMr, scale)
    assert np.array_equal(dmh.to_array(np.int32, K), mhr)
    # commit kernel: write two candidates back and read the population
    slot = np.array([0, K - 1], np.int32)
    kind = np.array([1, 2], np.int32)
    rows = np.array([1 if i0 != 1 else 2, N - 1 if i0 != N - 1 else N - 2], np.int64)
    if K == 1:
        slot, kind, rows = slot[:1], kind[:1], rows[:1]
    ds, dk, dr = DevBuf.from_array(slot), DevBuf.from_array(kind), DevBuf.from_array(rows)
    assert L.nla_k_crs_commit(n, ld, dX.ptr, dTX.ptr, dTM.ptr, len(slot), ds.ptr, dk.ptr, dr.ptr, None) == 0
    assert L.nla_stream_sync(None) == 0
    X2 = dX.to_array(np.float64, N * ld).reshape(N, ld)
    Xe = X.copy()
    for s, k, r in zip(slot, kind, rows):
        Xe[r, :n] = (TXr if k == 1 else TMr)[s, :n]
    assert np.array_equal(X2[:, :n], Xe[:, :n])
