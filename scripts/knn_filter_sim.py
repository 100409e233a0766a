"""Developer tool (CPU only, numpy): what the compare filters of the float matcher would let through on one image pair of
bench.py's float descriptors (512 * unit-norm): the f16 product with its rigorous error bound E (k_knn2_f16, round 5) and an int8
coarse filter with rounding-error norms - rows passing per query, share of (register slot, wave) tests
with a hit per 256-row window, rows inside the final bound.  The numbers quoted in DESIGN.md section 4 come from here."""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from metricsfm_amd import scene
sc = scene.config_scene(3)
scene.add_features(sc, 4096, images=range(2))
d = [(512.0 * x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32) for x in sc.desc[:2]]
A, B = d[0].astype(np.float64), d[1].astype(np.float64)
vmax = max(np.abs(A).max(), np.abs(B).max())
import math
ex = math.frexp(vmax)[1]; s = 2.0 ** (14 - ex)
print("vmax", vmax, "scale", s)
sa, sb = A * s, B * s
ha, hb = sa.astype(np.float16).astype(np.float64), sb.astype(np.float16).astype(np.float64)
ra = np.linalg.norm(sa - ha, axis=1); rb = np.linalg.norm(sb - hb, axis=1)
a2 = (sa ** 2).sum(1); b2 = (sb ** 2).sum(1)
def ebound(a2max, ramax, b2, rb, shift):
    na, nb = np.sqrt(a2max), np.sqrt(b2)
    dot = ramax * (nb + rb) + na * rb
    return 1.02 * (2 * dot + 1.7e-5 * (2.02 * na * nb + a2max + shift) + 1.2e-7 * (a2max + b2 + shift)) + 1e-5
shift = b2.max()
for it in range(3): shift = b2.max() + 2 * ebound(a2.max(), ra.max(), b2.max(), rb.max(), shift * 1.001)
E = ebound(a2.max(), ra.max(), b2, rb, shift)
print("E mean", E.mean(), "a2 typical", a2.mean(), "shift", shift)
# approximate v = a2 + shift - 2 ha.hb   (queries = B rows, train = A rows)
V = a2[None, :] + shift - 2 * hb @ ha.T     # [query, train]
D = V - (shift - b2)[:, None]
srt = np.sort(D, axis=1)
print("scaled d2: nearest mean", srt[:,0].mean(), "2nd", srt[:,1].mean(), "3rd", srt[:,2].mean(), "median row", np.median(D))
print("gap 2->3 mean", (srt[:,2]-srt[:,1]).mean(), "median", np.median(srt[:,2]-srt[:,1]))
# simulate filter: threshold from list state at last flush (every 256 rows), thr = v2*(1+2^-15)+2E
nq, nt = V.shape
hits_per_block = []
tot_hits = np.zeros(nq)
for blk in range(nt // 256):
    if blk == 0:
        thr = np.full(nq, np.inf)
    else:
        part = np.partition(V[:, :blk*256], 1, axis=1)[:, 1]
        thr = part * (1 + 2**-15) + 2 * E
    h = (V[:, blk*256:(blk+1)*256] <= thr[:, None])
    tot_hits += h.sum(1)
    # per 32-row step and slot: wave = 64 queries (2 sets x 32), slot = one row-of-16 per lane half... approximate: slot = 1 train row x 64 queries x2 halves -> 128 candidates = 2 rows x 64 queries
    hh = h.reshape(nq // 64, 64, 128, 2)   # waves, queries, slot, rows-in-slot (2 rows: the two half lanes)
    slot_hit = hh.any(axis=(1, 3))          # [wave, slot]
    hits_per_block.append(slot_hit.mean())
print("survivors per query (mean)", tot_hits.mean(), "excluding first window", (tot_hits - 256).mean())
print("fraction of slots with a hit per 256-row block:", np.round(hits_per_block, 3))
print("mean over sweep", np.mean(hits_per_block))

# ---- int8 coarse filter: round s8*v to int8 steps, exact integer distances dhat, |d - dhat| <= |e_a| + |e_b|
print("---- int8 coarse filter")
q8 = 255.0 / vmax          # steps per unit so that the largest value maps to 255 (a - 128 in int8)
ia, ib = np.rint(A * q8), np.rint(B * q8)
ea = np.linalg.norm(A * q8 - ia, axis=1); eb = np.linalg.norm(B * q8 - ib, axis=1)
print("quantisation step %.3f, |e| mean %.2f max %.2f" % (1 / q8, ea.mean(), ea.max()))
D8 = ((ib ** 2).sum(1)[:, None] + (ia ** 2).sum(1)[None, :] - 2 * ib @ ia.T)   # exact integer squared distances of the rounded rows
dh = np.sqrt(np.maximum(D8, 0))
# a row can be among the two nearest only if dhat - (ea_max + eb) <= (second smallest of dhat + ea_max + eb)
marg = ea.max() + eb
tot = np.zeros(nq); slot_frac = []
for blk in range(nt // 256):
    if blk == 0:
        thr = np.full(nq, np.inf)
    else:
        thr = np.partition(dh[:, :blk * 256], 1, axis=1)[:, 1] + 2 * marg
    hsel = dh[:, blk * 256:(blk + 1) * 256] <= thr[:, None]
    tot += hsel.sum(1)
    hh = hsel.reshape(nq // 64, 64, 128, 2)
    slot_frac.append(hh.any(axis=(1, 3)).mean())
final_thr = np.partition(dh, 1, axis=1)[:, 1] + 2 * marg
final_surv = (dh <= final_thr[:, None]).sum(1)
print("survivors filed per query over the sweep: mean %.1f (without the first window %.1f); rows inside the FINAL bound: mean %.1f median %.0f p90 %.0f max %d"
      % (tot.mean(), tot.mean() - 256, final_surv.mean(), np.median(final_surv), np.percentile(final_surv, 90), final_surv.max()))
print("fraction of slots with a hit per 256-row block:", np.round(slot_frac, 3), "mean", np.mean(slot_frac))
