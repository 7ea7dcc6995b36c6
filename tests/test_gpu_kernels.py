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


@pytest.mark.parametrize("seed,predraw,pop,rel0", [(42, 0, 1000, 0), (7, 1, 777, 6), (5489, 623, 65, 2 * 64 * 7), (9, 3, 2, 10), (11, 0, 5000, 4)])
def test_ranking_bits_fused_with_the_generator_equal_words_then_bits(L, seed, predraw, pop, rel0):
    """mt_rankbits_kernel (words -> u < PF bits without the words ever reaching memory, all segments in one launch) against the
    two-pass statement it replaces: the serial generator's words, nlopt_urand(0,1) < 0.45 per pair (isres.c:210), bit j of row i for
    step i (pop-1) + j.  Both parities of the ranking's first word (predraw odd: a step then straddles two regenerations, and two
    segments), rows that end inside a 64-bit word, a sub-range of the sweeps (the multi-rank partition), pop = 2."""
    popm1 = pop - 1
    roww = (popm1 + 63) // 64
    L.nlopt_srand(seed)
    for _ in range(predraw):
        L.nla_genrand_int32()
    nsweeps = pop
    total = rel0 + 2 * popm1 * nsweeps
    ref = words_from_seed(seed, total + 8, skip=predraw)
    w = ref[rel0:rel0 + 2 * popm1 * nsweeps].astype(np.uint64)
    u = ((w[0::2] >> np.uint64(5)) * 67108864.0 + (w[1::2] >> np.uint64(6)).astype(np.float64)) * (1.0 / 9007199254740992.0)
    want = np.zeros((nsweeps, roww * 64), bool)
    want[:, :popm1] = (u < 0.45).reshape(nsweeps, popm1)
    want_words = np.packbits(want.reshape(nsweeps, roww, 64), axis=2, bitorder="little").view(np.uint64).reshape(nsweeps, roww)
    s = L.nla_mtstream_create(None)
    assert s
    try:
        for first, last in ((0, nsweeps), (nsweeps // 3, nsweeps - nsweeps // 4)):
            d = DevBuf.from_array(np.zeros(nsweeps * roww, np.uint64))
            assert L.nla_mtstream_rankbits(s, rel0, rel0 + 2 * popm1 * first, 2 * popm1 * (last - first), popm1, roww, d.ptr) == 0
            assert L.nla_stream_sync(None) == 0
            got = d.to_array(np.uint64, nsweeps * roww).reshape(nsweeps, roww)
            assert np.array_equal(got[first:last], want_words[first:last]), (first, last)
            assert not got[:first].any() and not got[last:].any()
            d.free()
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


def _spec_inputs(n, N, K, seed, obj, align=2):
    P = O.port()
    ld = (n + align - 1) & ~(align - 1)
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


class SlotStatus(C.Structure):
    _fields_ = [("fT", C.c_double), ("fM", C.c_double), ("t", C.c_int32), ("pad", C.c_int32)]


@pytest.mark.parametrize("obj,n,N,K,i0", [("rastrigin", 10, 100, 7, 0), ("rastrigin", 10, 11, 9, 10), ("griewank", 64, 70, 60, 33),
                                          ("ackley", 257, 600, 40, 599), ("levy", 128, 140, 130, 17), ("rosenbrock", 512, 700, 256, 3),
                                          ("griewank", 4096, 4200, 24, 4199), ("sphere", 2, 9, 8, 4), ("griewank", 2048, 2100, 20, 77),
                                          ("ackley", 300, 320, 30, 5), ("ackley", 9000, 9100, 6, 5), ("rastrigin", 1000, 1100, 200, 1)])
def test_chain_kernel_resolves_the_window_like_the_sequential_statement(L, obj, n, N, K, i0):
    chain_kernel_case(L, obj, n, N, K, i0)


def chain_kernel_case(L, obj, n, N, K, i0):
    """nla_k_crs_chain (hip/crs_chain.hip): one launch computes every slot of the window, evaluates it, replays the accept / reject
    chain on the window's worst rows and lets later slots read what the chain says a worst row holds at their turn.  Against the
    sequential statement orc_k_crs_chain: bit-exact trial points and mutations, f within 1e-10, the same records of what every
    slot read from where.  Small populations (N barely above n) make every slot depend on MANY earlier slots of the same launch:
    the in-kernel waiting, evaluation and resolution are what is tested.  The objective values of the rows are random, so the
    chain has rejections, accepted mutations and values landing among the worst rows again.  The chain is advanced by the launch's
    resolver wavefront (hip/crs_chain_resolver.h); a trial that becomes the new best point ends the window (control word `halt` =
    2 | (slot + 1) << 8): later slots may come back "not computed" (status.t = 0), never with a wrong point."""
    P = O.port()
    ring = 2 * K + 3
    first = 3 * ring + 2
    mask = 511
    ld, lb, ub, X, w0, jn0, pos0, last0 = _spec_inputs(n, N, ring, 177 + n, obj, align=16)      # the chain kernel's contract: rows on 128-byte lines
    oid = O.OBJ[obj]
    ent = [(first + a) % ring for a in range(ring)]
    w = np.zeros(2 * n * ring, np.uint32)
    jn, pos, last = np.zeros(ring, np.int32), np.zeros(ring * n, np.int32), np.zeros(ring, np.int32)
    for a in range(ring):
        w[ent[a] * 2 * n:(ent[a] + 1) * 2 * n] = w0[a * 2 * n:(a + 1) * 2 * n]
        jn[ent[a]], last[ent[a]] = jn0[a], last0[a]
        pos[ent[a] * n:(ent[a] + 1) * n] = pos0[a * n:(a + 1) * n]
    # f of every row (the real objective), the K worst rows worst first with the reference's tie rule, the best row = i0 by decree
    F = np.zeros(N)
    P.orc_k_eval.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    P.orc_k_eval(oid, n, ld, X.ctypes.data, N, F.ctypes.data)
    F[i0] = F.min() - 1.0
    order = np.lexsort((np.arange(N), F))[::-1]
    nW = min(K, N - 1)
    W = order[:nW].astype(np.int64)
    # make the chain interesting: pull the worst values close together so that trial values fall on both sides of them
    Wf = F[W].copy()
    fbest = float(F[i0])

    class St(C.Structure):
        _fields_ = [("fT", C.c_double), ("fM", C.c_double), ("t", C.c_int32), ("pad", C.c_int32)]
    fwcap = 48
    nslot = mask + 1
    P.orc_k_crs_chain.restype = None
    P.orc_k_crs_chain.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    dX, dlb, dub, dw = DevBuf.from_array(X), DevBuf.from_array(lb), DevBuf.from_array(ub), DevBuf.from_array(w)
    dj, dp, dl = DevBuf.from_array(jn), DevBuf.from_array(pos), DevBuf.from_array(last)
    dW, dWf = DevBuf.from_array(W), DevBuf.from_array(Wf)
    dTX, dTM = DevBuf.from_array(np.zeros(nslot * ld), uncached=True), DevBuf.from_array(np.zeros(nslot * ld), uncached=True)
    cb = L.nla_crs_chain_ctrl_bytes(256, 256)
    dctrl = DevBuf.from_array(np.zeros(cb, np.uint8), uncached=True)      # the kernel's contract: TX, TM, ctrl are uncached memory
    dst = DevBuf(C.sizeof(St) * K)
    dcnt, drec = DevBuf.from_array(np.zeros(K, np.uint32)), DevBuf.from_array(np.zeros(K * fwcap, np.uint32))
    for rep in range(2):                        # twice on the same control block: the ticket base carries over
        assert L.nla_memset(dst.ptr, 0, C.sizeof(St) * K, None) == 0
        assert L.nla_k_crs_chain(oid, n, ld, dX.ptr, i0, fbest, dj.ptr, dp.ptr, dl.ptr, dw.ptr, ring, first, K, dW.ptr, dWf.ptr, nW, 0, mask,
                                 dlb.ptr, dub.ptr, dTX.ptr, dTM.ptr, dctrl.ptr, rep * L.nla_crs_chain_tickets(n, ld, K), dst.ptr, dcnt.ptr,
                                 drec.ptr, fwcap, None) == 0
        assert L.nla_stream_sync(None) == 0
        raw = dst.to_array(np.uint8, C.sizeof(St) * K)
        st = np.frombuffer(raw.tobytes(), dtype=[("fT", "f8"), ("fM", "f8"), ("t", "i4"), ("pad", "i4")])
        craw = dctrl.to_array(np.uint8, cb)
        chead = np.frombuffer(craw[:32].tobytes(), np.uint32)
        # every slot up to a new best point is computed; behind it a slot is either computed in full or not at all
        live = int(chead[3] >> 8) if (chead[3] & 2) else K
        assert np.all(st["t"][:live] == n) and np.all((st["t"] == n) | (st["t"] == 0))
        done = st["t"] == n
        # the sequential statement, its accept / reject decisions taken on the DEVICE's f values (they differ from the host's in the
        # last bits; with N barely above n the trial points are nearly equal and so are their f: a comparison could go either way)
        dev_status = (St * K)()
        C.memmove(dev_status, raw.tobytes(), C.sizeof(St) * K)
        TXr, TMr = np.zeros((nslot, ld)), np.zeros((nslot, ld))
        str_ = (St * K)()
        cntr, recr = np.zeros(K, np.uint32), np.zeros(K * fwcap, np.uint32)
        dbg = np.zeros(8 + 256, np.uint32)
        P.orc_k_crs_chain(oid, n, ld, X.ctypes.data, i0, fbest, jn.ctypes.data, pos.ctypes.data, last.ctypes.data, w.ctypes.data, ring, first, K,
                          W.ctypes.data, Wf.ctypes.data, nW, mask, lb.ctypes.data, ub.ctypes.data, TXr.ctypes.data, TMr.ctypes.data,
                          C.addressof(str_), cntr.ctypes.data, recr.ctypes.data, fwcap, C.addressof(dev_status), dbg.ctypes.data)
        # the chain as the device resolved it (control block: 8 u32, 32 f64 + 32 i64 of landed values, 2K record words, done[K], K unused words, rowstate[nW])
        rs_dev = np.frombuffer(craw[32 + 512 + 16 * K + 8 * K: 32 + 512 + 16 * K + 8 * K + 4 * nW].tobytes(), np.uint32)
        assert np.array_equal(rs_dev, dbg[8:8 + nW]), ("who overwrote which worst row", "device next/halt/naccept/wp/nextra", chead[2:7].tolist(),
                                                     "statement next/halt/wp/nextra", dbg[:4].tolist(),
                                                     "first difference at", int(np.flatnonzero(rs_dev != dbg[8:8 + nW])[0]),
                                                     rs_dev[:12].tolist(), dbg[8:20].tolist())
        fTr = np.array([str_[a].fT for a in range(K)])
        fMr = np.array([str_[a].fM for a in range(K)])
        scale = np.abs(np.concatenate([fTr, fMr])).mean()
        cnt, rec = dcnt.to_array(np.uint32, K), drec.to_array(np.uint32, K * fwcap).reshape(K, fwcap)
        TX = dTX.to_array(np.float64, nslot * ld).reshape(nslot, ld)
        TM = dTM.to_array(np.float64, nslot * ld).reshape(nslot, ld)
        for a in range(K):
            qa = (first + a) & mask
            if not done[a]:
                continue
            assert cnt[a] == cntr[a], (a, cnt[a], cntr[a])
            if cnt[a] <= fwcap:                  # (beyond the capacity which records survive is arbitrary; the caller discards such a slot)
                k = int(cnt[a])
                assert sorted(rec[a, :k].tolist()) == sorted(recr.reshape(K, fwcap)[a, :k].tolist()), a
            assert np.array_equal(TX[qa, :n], TXr[qa, :n]), a
            assert np.array_equal(TM[qa, :n], TMr[qa, :n]), a
        assert close(st["fT"][done], fTr[done], scale) and close(st["fM"][done], fMr[done], scale)
    kinds = (recr >> 16) & 3
    assert K < 8 or (cntr.sum() > 0 and (kinds[recr > 0] > 0).any())      # the case does exercise reading from producers


@pytest.mark.parametrize("obj,n,N,K,i0,variant", [("rastrigin", 10, 100, 7, 0, 0), ("rastrigin", 10, 11, 5, 10, 0),
                                                  ("griewank", 64, 500, 33, 250, 0), ("ackley", 257, 600, 9, 599, 0),
                                                  ("levy", 128, 300, 4, 17, 0), ("rosenbrock", 512, 2000, 6, 3, 0),
                                                  ("griewank", 4096, 4200, 3, 4199, 0), ("sphere", 1, 9, 8, 4, 0),
                                                  ("griewank", 2048, 3000, 5, 77, 832), ("griewank", 1024, 3000, 5, 77, 1616),
                                                  ("ackley", 300, 700, 6, 5, 132), ("ackley", 9000, 9100, 2, 5, 416),
                                                  ("griewank", 2048, 3000, 5, 77, 10816), ("ackley", 300, 700, 6, 5, 10432)])
def test_advance_finish_commit_kernels(L, obj, n, N, K, i0, variant):
    """the resumable gather-sum in two passes (first one stopped by a hazard list), evaluation and
    mutation of the finished slots, and the commit: bit-exact x / partial sums / t, f within 1e-10"""
    P = O.port()
    ring = K + 1                      # block b at ring entry b % ring; the window starts at block `first`
    first = 3 * ring + 2              # exercises the ring wrap and the slot mask
    mask = 63
    ld, lb, ub, X, w0, jn0, pos0, last0 = _spec_inputs(n, N, ring, 31 + n, obj)
    oid = O.OBJ[obj]
    # ring layout: entry (first+a) % ring holds what the oracle digested for window slot a
    ent = [(first + a) % ring for a in range(ring)]
    w = np.zeros(2 * n * ring, np.uint32)
    jn, pos, last = np.zeros(ring, np.int32), np.zeros(ring * n, np.int32), np.zeros(ring, np.int32)
    for a in range(ring):
        w[ent[a] * 2 * n:(ent[a] + 1) * 2 * n] = w0[a * 2 * n:(a + 1) * 2 * n]
        jn[ent[a]], last[ent[a]] = jn0[a], last0[a]
        pos[ent[a] * n:(ent[a] + 1) * n] = pos0[a * n:(a + 1) * n]
    # hazard list: W[a-1] = a row sampled by slot a (so slot a must stop there or earlier), plus the best row
    P.orc_k_advance_slot.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                     C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5 + n)
    W = np.zeros(max(K, 1), np.int64)
    for a in range(1, K):
        t = int(rng.integers(0, n))
        if t < n - 1:
            r = int(pos0[a * n + t]); r += (r >= i0)
        else:
            r = int(pos0[a * n + n - 1]); r += (r >= i0); r += int(last0[a]); r += (r == i0)
        W[a - 1] = r
    if K >= 3:
        W[K - 2] = i0                 # the best row in the list must be ignored
    nW = K - 1
    # oracle, pass 1 and pass 2
    acc = np.zeros((K, ld))
    t1r, t2r = np.zeros(K, np.int32), np.zeros(K, np.int32)
    for a in range(K):
        t1r[a] = P.orc_k_advance_slot(n, ld, X.ctypes.data, i0, int(jn0[a]), pos0[a * n:].ctypes.data, int(last0[a]),
                                      W.ctypes.data, min(a, nW), 0, lb.ctypes.data, ub.ctypes.data, acc[a].ctypes.data)
    acc1 = acc.copy()
    for a in range(K):
        t2r[a] = P.orc_k_advance_slot(n, ld, X.ctypes.data, i0, int(jn0[a]), pos0[a * n:].ctypes.data, int(last0[a]),
                                      W.ctypes.data, 0, int(t1r[a]), lb.ctypes.data, ub.ctypes.data, acc[a].ctypes.data)
    assert np.all(t2r == n)
    TXr = acc
    TMr = np.zeros((K, ld))
    P.orc_k_mutate.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for a in range(K):
        P.orc_k_mutate(n, X[i0].ctypes.data, TXr[a].ctypes.data, w0[(a + 1) * 2 * n:].ctypes.data, lb.ctypes.data,
                       ub.ctypes.data, TMr[a].ctypes.data)
    fTr, fMr = np.zeros(K), np.zeros(K)
    P.orc_k_eval.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    P.orc_k_eval(oid, n, ld, TXr.ctypes.data, K, fTr.ctypes.data)
    P.orc_k_eval(oid, n, ld, TMr.ctypes.data, K, fMr.ctypes.data)
    # device
    dX, dlb, dub, dw = DevBuf.from_array(X), DevBuf.from_array(lb), DevBuf.from_array(ub), DevBuf.from_array(w)
    dj, dp, dl, dW = DevBuf.from_array(jn), DevBuf.from_array(pos), DevBuf.from_array(last), DevBuf.from_array(W)
    nslot = mask + 1
    dTX, dTM = DevBuf(8 * ld * nslot), DevBuf(8 * ld * nslot)
    dfT, dfM, dst = DevBuf(8 * nslot), DevBuf(8 * nslot), DevBuf(C.sizeof(SlotStatus) * K)
    dt0, dt1, dt2 = DevBuf.from_array(np.zeros(K, np.int32)), DevBuf(4 * K), DevBuf(4 * K)
    q = [(first + a) & mask for a in range(K)]

    def status():
        raw = dst.to_array(np.uint8, C.sizeof(SlotStatus) * K)
        return np.frombuffer(raw.tobytes(), dtype=[("fT", "f8"), ("fM", "f8"), ("t", "i4"), ("pad", "i4")])

    def run(t_in, t_out, nw):
        assert L.nla_k_crs_advance(n, ld, dX.ptr, i0, dj.ptr, dp.ptr, dl.ptr, ring, first, K, dW.ptr, nw, t_in.ptr, t_out.ptr,
                                   mask, dlb.ptr, dub.ptr, dTX.ptr, variant, None) == 0
        assert L.nla_k_crs_finish(oid, n, ld, dX.ptr, i0, dTX.ptr, dTM.ptr, dw.ptr, ring, first, K, t_in.ptr, t_out.ptr, mask,
                                  dlb.ptr, dub.ptr, dfT.ptr, dfM.ptr, dst.ptr, None) == 0
        assert L.nla_stream_sync(None) == 0

    run(dt0, dt1, nW)
    st1 = status()
    assert np.array_equal(dt1.to_array(np.int32, K), t1r) and np.array_equal(st1["t"], t1r)
    TX1 = dTX.to_array(np.float64, nslot * ld).reshape(nslot, ld)
    for a in range(K):
        if t1r[a] > 0:
            assert np.array_equal(TX1[q[a], :n], acc1[a, :n]), a        # partial sums are bit-exact too
    scale = np.abs(np.concatenate([fTr, fMr])).mean()
    done1 = t1r == n
    assert close(st1["fT"][done1], fTr[done1], scale) and close(st1["fM"][done1], fMr[done1], scale)
    run(dt1, dt2, 0)
    st2 = status()
    assert np.all(dt2.to_array(np.int32, K) == n) and np.all(st2["t"] == n)
    TX = dTX.to_array(np.float64, nslot * ld).reshape(nslot, ld)[q][:, :n]
    TM = dTM.to_array(np.float64, nslot * ld).reshape(nslot, ld)[q][:, :n]
    assert np.array_equal(TX, TXr[:, :n])          # bit-exact trial points (row order, no FMA)
    assert np.array_equal(TM, TMr[:, :n])          # bit-exact mutations
    assert close(st2["fT"], fTr, scale) and close(st2["fM"], fMr, scale)
    # slots finished in pass 1 keep their first-pass results (not recomputed)
    assert np.array_equal(st2["fT"][done1], st1["fT"][done1])
    # commit kernel: write two candidates back and read the population
    slot = np.array([q[0], q[K - 1]], np.int32)
    kind = np.array([1, 2], np.int32)
    rows = np.array([1 if i0 != 1 else 2, N - 1 if i0 != N - 1 else N - 2], np.int64)
    src = [0, K - 1]
    if K == 1:
        slot, kind, rows, src = slot[:1], kind[:1], rows[:1], src[:1]
    ds, dk, dr = DevBuf.from_array(slot), DevBuf.from_array(kind), DevBuf.from_array(rows)
    assert L.nla_k_crs_commit(n, ld, dX.ptr, dTX.ptr, dTM.ptr, len(slot), ds.ptr, dk.ptr, dr.ptr, None) == 0
    assert L.nla_stream_sync(None) == 0
    X2 = dX.to_array(np.float64, N * ld).reshape(N, ld)
    Xe = X.copy()
    for a, k, r in zip(src, kind, rows):
        Xe[r, :n] = (TXr if k == 1 else TMr)[a, :n]
    assert np.array_equal(X2[:, :n], Xe[:, :n])
