"""Where does a CRS2_LM run leave the reference's path?  Helpers for the -m gpu parity tests and tools/stress_crs.py: compare a
run's trace (f, row, kind, accepted per evaluation) with the committed checkpoints of crs_golden.json (machine independent) and
with a live oracle trace, and say what differs FIRST — before any assertion on the end result — so that a one-off failure on a
box nobody can go back to still leaves its evidence (also appended to gpurun_out/crs_divergence.jsonl)."""
import json
import os
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-10


def _decision_bytes(t):
    rec = np.zeros(len(t), dtype=[("row", "<i8"), ("kind", "<i4"), ("accepted", "<i4")])
    rec["row"], rec["kind"], rec["accepted"] = t["row"], t["kind"], t["accepted"]
    return rec


def first_bad_checkpoint(trace, ck):
    """index of the first block of ck['every'] evaluations whose decisions (CRC32) or last f (1e-10) differ from the golden
    run's, or None.  A missing / surplus block counts as different."""
    every, crcs, fl = ck["every"], ck["crc32"], ck["f_last"]
    rec = _decision_bytes(trace)
    scale = float(np.abs(trace["f"]).mean()) if len(trace) else 1.0
    nblk = (len(rec) + every - 1) // every
    for b in range(max(nblk, len(crcs))):
        if b >= nblk or b >= len(crcs):
            return b
        blk = rec[b * every:(b + 1) * every]
        if (zlib.crc32(blk.tobytes()) & 0xffffffff) != crcs[b]:
            return b
        fg = float.fromhex(fl[b])
        if abs(float(trace["f"][min((b + 1) * every, len(rec)) - 1]) - fg) > RTOL * max(abs(fg), scale):
            return b
    return None


def first_divergence(ta, tp):
    """first evaluation at which trace ta differs from the oracle's tp: (index, what) or None"""
    m = min(len(ta), len(tp))
    scale = float(np.abs(tp["f"]).mean()) if len(tp) else 1.0
    bad = (ta["row"][:m] != tp["row"][:m]) | (ta["kind"][:m] != tp["kind"][:m]) | (ta["accepted"][:m] != tp["accepted"][:m]) | \
          (np.abs(ta["f"][:m] - tp["f"][:m]) > RTOL * np.maximum(np.abs(tp["f"][:m]), scale))
    idx = np.flatnonzero(bad)
    if idx.size:
        i = int(idx[0])
        what = [k for k in ("row", "kind", "accepted") if ta[k][i] != tp[k][i]] or ["f"]
        return i, "+".join(what)
    if len(ta) != len(tp):
        return m, "length %d vs %d" % (len(ta), len(tp))
    return None


def explain(name, a, p=None, g=None, N=None, extra=None):
    """a = device run (run_amd dict with trace), p = live oracle run (or None), g = golden record (or None).  Returns a report
    dict; report['ok'] says whether anything differs.  Never raises."""
    rep = dict(case=name, ok=True, ret=int(a["ret"]), nevals=int(a["nevals"]), minf=float(a["minf"]).hex())
    ta = a["trace"]
    try:
        if g is not None and "checkpoints" in g:
            b = first_bad_checkpoint(ta, g["checkpoints"])
            if b is not None:
                rep.update(ok=False, golden_first_bad_block=b, golden_block_evals=[b * g["checkpoints"]["every"], (b + 1) * g["checkpoints"]["every"]])
        if p is not None:
            tp = p["trace"]
            d = first_divergence(ta, tp)
            if d is not None:
                i, what = d
                n_init = int(N) if N else int(np.count_nonzero(tp["kind"] == 0))
                rep.update(ok=False, oracle_first_diff=i, oracle_diff_what=what, phase="init" if i < n_init else "trial", init_evals=n_init)
                lo, hi = max(0, i - 2), min(min(len(ta), len(tp)), i + 3)
                rep["around"] = [dict(i=j, dev=[float(ta["f"][j]).hex(), int(ta["row"][j]), int(ta["kind"][j]), int(ta["accepted"][j])],
                                      orc=[float(tp["f"][j]).hex(), int(tp["row"][j]), int(tp["kind"][j]), int(tp["accepted"][j])]) for j in range(lo, hi)]
                m = min(len(ta), len(tp), n_init)
                scale = float(np.abs(tp["f"][:m]).mean()) if m else 1.0
                badrows = np.flatnonzero(np.abs(ta["f"][:m] - tp["f"][:m]) > RTOL * np.maximum(np.abs(tp["f"][:m]), scale))
                rep["init_rows_differing"] = int(badrows.size)
                if badrows.size:
                    rep["init_rows_first"] = [int(v) for v in badrows[:24]]
                    rep["init_rows_last"] = int(badrows[-1])
                    runs = np.split(badrows, np.flatnonzero(np.diff(badrows) != 1) + 1)
                    rep["init_bad_runs"] = [[int(r[0]), int(r[-1])] for r in runs[:16]]
        if not rep["ok"]:
            rep["stats"] = {k: a["stats"][k] for k in ("rounds", "slots_launched", "slots_used", "slots_invalid", "slots_newbest", "slots_role",
                                                       "accepted", "evals_init", "evals_trial", "evals_mutation", "mt_words") if k in a["stats"]}
            if extra:
                rep.update(extra)
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "crs_divergence.jsonl"), "a") as f:
                f.write(json.dumps(rep) + "\n")
    except Exception as e:          # the report must never hide the assertion that follows it
        rep["explain_error"] = repr(e)
    return rep
