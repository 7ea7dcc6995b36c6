"""-m gpu: an MT19937 device stream cut into SHORTER segments than the 1024 regenerations the other algorithms use (mtstream.c:
nla_mtstream_create_seg; hip/mt_kernels.hip: mt_generate_seg_kernel, nla_k_mt_generate_seg) and MLSL on such a stream
(`nlopt_set_param(opt, "amd_mlsl_seg_regens", s)`; MLSL's default is 64 since round 5: 205 wavefronts generate an iteration's 8 M words
at config 4 instead of 13 — 15.9 -> 14.4 ms per iteration on the MI355X, profiles/r05_staged_ab.txt).  The words a stream delivers do not
depend on how it is cut — that is the whole contract; tests/test_mt_segments_emulated.py is the CPU twin (segment arithmetic of
mtstream.c over the emulated device)."""
import ctypes as C

import numpy as np
import pytest

import nlopt_amd
from nlopt_amd import DevBuf
from test_gpu_kernels import words_from_seed
from test_gpu_mlsl import run_amd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    L = nlopt_amd.lib()
    assert nlopt_amd.device_count() > 0, "no HIP device: the product has no CPU fallback"
    L.nla_mtstream_create_seg.argtypes = [C.c_void_p, C.c_int]
    L.nla_mtstream_create_seg.restype = C.c_void_p
    return L


@pytest.mark.parametrize("seg,seed,predraw", [(1, 5489, 0), (2, 42, 1000), (8, 7, 624), (64, 123456789, 623), (256, 42, 1), (512, 9, 0), (1024, 3, 17)])
def test_short_segment_stream_matches_serial_generator(L, seg, seed, predraw):
    """test_gpu_kernels.py::test_mt_stream_matches_serial_generator for a stream cut into segments of `seg` regenerations: the same words at
    ragged offsets that cross segment boundaries and reach segments only obtainable through several doubling rounds (hundreds of segments
    for the short lengths), fills that go backwards, and the host generator left where the reference's would be"""
    SEG = 624 * seg
    L.nlopt_srand(seed)
    for _ in range(predraw):
        L.nla_genrand_int32()
    nseg = 6 if seg >= 256 else (40 if seg >= 8 else 300)
    total = nseg * SEG + 12345
    ref = words_from_seed(seed, total, skip=predraw)
    s = L.nla_mtstream_create_seg(None, seg)
    assert s
    try:
        for first, count in [(0, 5000), (SEG - 100, 1000), (3, 2 * SEG + 17), ((nseg - 1) * SEG + 999, SEG + 11346), (4 * SEG - 1, 2),
                             (SEG // 2, (nseg // 2) * SEG + 5)]:
            d = DevBuf(4 * count)
            assert L.nla_mtstream_fill(s, first, count, d.ptr) == 0
            assert L.nla_stream_sync(None) == 0
            got = d.to_array(np.uint32, count)
            assert np.array_equal(got, ref[first:first + count]), (first, count)
            d.free()
        used = 3 * SEG + 4321
        assert L.nla_mtstream_finish(s, used) == 0
        nxt = [L.nla_genrand_int32() for _ in range(2000)]
        assert nxt == list(ref[used:used + 2000])
    finally:
        L.nla_mtstream_destroy(s)


def test_segment_lengths_that_are_not_served(L):
    for seg in (0, -4, 3, 96, 2048):
        assert not L.nla_mtstream_create_seg(None, seg)
    s = L.nla_mtstream_create_seg(None, 64)                     # the fused ranking-bits kernel is for the default layout only
    d = DevBuf(8 * 64)
    assert L.nla_mtstream_rankbits(s, 0, 0, 200, 100, 2, d.ptr) != 0
    L.nla_mtstream_destroy(s)
    assert L.nla_k_mt_generate_seg(None, 0, 1, 0, 10, d.ptr, 48, None) != 0


def _run(params, **kw):
    """test_gpu_mlsl.run_amd with nlopt_set_param values on the global optimiser"""
    orig = nlopt_amd.Opt.optimize_raw

    def patched(self, xs):
        for k, v in params.items():
            self.set_param(k, v)
        return orig(self, xs)
    nlopt_amd.Opt.optimize_raw = patched
    try:
        return run_amd(**kw)
    finally:
        nlopt_amd.Opt.optimize_raw = orig


@pytest.mark.parametrize("seg", [64, 1, 256])
@pytest.mark.parametrize("kw", [dict(obj="rastrigin", n=8, nsamples=40, seed=5, stopval=1.5, maxeval=100000),
                                dict(obj="ackley", n=300, nsamples=200, seed=7, maxeval=30000),
                                dict(obj="ackley", n=1200, nsamples=300, seed=11, maxeval=20000, alg=nlopt_amd.G_MLSL_LDS),
                                dict(obj="griewank", n=5, nsamples=0, seed=3, stopval=1e-7, maxeval=100000)])
def test_mlsl_on_short_segments_is_the_same_run(seg, kw):
    """the same words at the same offsets: every evaluation, the minimiser, the result and the stream position of an MLSL run do not
    depend on the segment length of its stream (pseudo-random sampling incl. the LDS variant above Sobol's 1111 dimensions)"""
    a = _run({}, **kw)
    b = _run({"amd_mlsl_seg_regens": seg}, **kw)
    assert a["ret"] == b["ret"] and a["nevals"] == b["nevals"] and a["minf"] == b["minf"] and np.array_equal(a["x"], b["x"])
    assert np.array_equal(a["trace"]["f"], b["trace"]["f"]) and a["stats"]["mt_words"] == b["stats"]["mt_words"]
