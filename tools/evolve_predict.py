"""evolve_predict.py — how well can the start of an individual's deviates be predicted? (analysis for hip/isres_evolve2.hip)

Input: the dump the emulated device writes with NLA_EMU_EVOLVE_DUMP=<file> (oracle/emu_device.c: per individual of the mutation phase
k, deviates consumed, expected redraws from the parent's x / sigma with sigma' ~ sigma, the same with sigma' = sigma exp(taup z_k)).
Replays the look-up rounds' window logic (256 individuals per round, window of 256 candidate starts centred on the prediction, exact
start of the round's first individual) with different predictors and counts rounds per generation."""
import sys
import numpy as np

rec = np.fromfile(sys.argv[1], dtype=np.float64).reshape(-1, 4)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
EVM, EVD = 256, 256
# split into generations: k restarts
k = rec[:, 0].astype(np.int64)
starts = np.flatnonzero(np.diff(k, prepend=k[0] + 1) < 0).tolist() + [len(k)]
if starts[0] != 0:
    starts = [0] + starts
for g in range(len(starts) - 1):
    r = rec[starts[g]:starts[g + 1]]
    cons = r[:, 1].astype(np.int64)
    red = cons - 1 - 2 * n
    mu0, mu1 = r[:, 2], r[:, 3]
    print("generation %d: %d individuals, redraws per individual mean %.2f sd %.2f; residual sd vs mu(sigma) %.2f, vs mu(sigma exp(taup z)) %.2f; corr %.3f / %.3f"
          % (g + 1, len(r), red.mean(), red.std(), (red - mu0).std(), (red - mu1).std(), np.corrcoef(red, mu0)[0, 1], np.corrcoef(red, mu1)[0, 1]))

    def rounds(pred_kind):
        i, nr, full, rho_r, rho_a, rho_m = 0, 0, 0, 0.0, 0.0, 0.0
        bias = 0.0
        while i < len(r):
            m = min(EVM, len(r) - i)
            c = cons[i:i + m]
            true_start = np.concatenate(([0], np.cumsum(c)[:-1]))            # relative to the round's exact first start
            ab = np.arange(m) * n                                            # mutated coordinates before individual q
            least = np.arange(m) + 2 * ab
            if pred_kind == "rate":
                rhoc = rho_r / rho_a if rho_a > 0 else 0.0
                pred = least + np.floor(rhoc * ab)
            elif pred_kind == "mu0":
                pred = least + np.floor(np.concatenate(([0], np.cumsum(mu0[i:i + m])[:-1])) * (1 + bias))
            elif pred_kind == "mudev":         # what hip/isres_evolve2.hip does: decayed observed / expected sums, 1.0 before anything was resolved
                ratio = rho_r / rho_m if rho_m > 0 else 1.0
                pred = least + np.floor(ratio * np.concatenate(([0], np.cumsum(mu0[i:i + m])[:-1])))
            elif pred_kind == "mu1":
                pred = least + np.floor(np.concatenate(([0], np.cumsum(mu1[i:i + m])[:-1])) * (1 + bias))
            else:
                pred = true_start
            d = true_start - (pred - EVD // 2)
            bad = np.flatnonzero((d < 0) | (d >= EVD))
            res = bad[0] if len(bad) else m
            res = max(res, 1)
            rsum = red[i:i + res].sum()
            rho_r = 0.9 * rho_r + rsum
            rho_a = 0.9 * rho_a + res * n
            rho_m = 0.9 * rho_m + mu0[i:i + res].sum()
            if pred_kind in ("mu0", "mu1"):
                mm = (mu0 if pred_kind == "mu0" else mu1)[i:i + res].sum()
                if mm > 0:
                    bias = 0.8 * bias + 0.2 * (rsum / mm - 1)
            full += res == m
            i += res
            nr += 1
        return nr, full
    for kind in ("rate", "mu0", "mudev", "mu1", "oracle"):
        nr, full = rounds(kind)
        print("   predictor %-6s: %4d rounds (%d full blocks), %.1f individuals per round" % (kind, nr, full, len(r) / nr))
