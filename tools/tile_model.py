"""Offline model of roi_tile.hip's tiling on a dump of proposals (ROI5_DUMP=... python tools/roi5.py): visits per RoI,
RoIs per (tile, image), workgroup imbalance.   python tools/tile_model.py gpurun_out/props5.npz [WIN STEP]"""
import sys
import numpy as np

d = np.load(sys.argv[1])
WIN = int(sys.argv[2]) if len(sys.argv) > 2 else 48
STEP = int(sys.argv[3]) if len(sys.argv) > 3 else 32
props, nk = d["props"], d["nk"]
levels = [(100, 168, 1 / 8), (50, 84, 1 / 16), (25, 42, 1 / 32), (13, 21, 1 / 64)]
k_min, k_max = 3, 6


def tiles(n):
    return 1 if n <= WIN else (n - WIN + STEP - 1) // STEP + 1


def axis(v, n):
    ok = ~((v < -1) | (v > n))
    v = np.maximum(v, 0)
    lo = v.astype(np.int64)
    cl = lo >= n - 1
    lo = np.where(cl, n - 1, lo)
    hi = np.where(cl, n - 1, lo + 1)
    return lo, hi, ok


def owners(start, bsz, n, nw):
    lo0, _, ok0 = axis(start + 0.25 * bsz, n)
    base = np.minimum(np.where(ok0, lo0, 0) // STEP, nw - 1)
    out = []
    for p in range(7):
        l0, h0, k0 = axis(start + p * bsz + 0.25 * bsz, n)
        l1, h1, k1 = axis(start + p * bsz + 0.75 * bsz, n)
        first = np.where(k0, l0, l1)
        last = np.where(k1, h1, h0)
        o = np.where(last < base * STEP + WIN, base, np.minimum(first // STEP, nw - 1))
        o = np.where(~k0 & ~k1, base, o)
        out.append(o)
    return np.stack(out, 1)


tot_visits = tot_rois = 0
per_wg = []
for b in range(props.shape[0]):
    p = props[b, :nk[b]]
    area = (p[:, 2] - p[:, 0]) * (p[:, 3] - p[:, 1])
    lv = np.clip(np.floor(4 + np.log2(np.sqrt(area) / 224) + 1e-6), k_min, k_max).astype(int) - k_min
    for l, (H, W, s) in enumerate(levels):
        q = p[lv == l]
        nwy, nwx = tiles(H), tiles(W)
        sw, sh = q[:, 0] * s, q[:, 1] * s
        rw, rh = np.maximum(q[:, 2] * s - sw, 1), np.maximum(q[:, 3] * s - sh, 1)
        oy, ox = owners(sh, rh / 7, H, nwy), owners(sw, rw / 7, W, nwx)
        cnt = np.zeros((nwy, nwx), int)
        for i in range(len(q)):
            ys, xs = np.unique(oy[i]), np.unique(ox[i])
            for y in ys:
                for x in xs:
                    cnt[y, x] += 1
            tot_visits += len(ys) * len(xs)
        tot_rois += len(q)
        per_wg.append((l, cnt))
print("WIN %d STEP %d: RoIs %d, visits %d (%.3f per RoI)" % (WIN, STEP, tot_rois, tot_visits, tot_visits / tot_rois))
for l in range(4):
    c = np.stack([c for ll, c in per_wg if ll == l])
    print("level %d: tiles %s, RoI visits per (tile, image): mean %.1f max %d; per-tile mean over images:\n%s"
          % (l, c.shape[1:], c.mean(), c.max(), np.round(c.mean(0)).astype(int)))
