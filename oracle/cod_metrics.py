"""Oracle: the COD measures of engine/utils/metrics/metric.py::statistics.  TEST INFRASTRUCTURE ONLY.

A float64 numpy restatement of what one ``statistics.step`` appends per image and of ``get_result`` (metric.py:19-74): MAE
(:187-207), ACC (:139-159), IoU (:161-185), S-measure (:209-313), E-measure (:315-435), F-measure (:437-500) and the weighted
F-measure (:503-560).  Written from the measures' definitions as sums over pixels -- the same reduced quantities the HIP kernels
produce (csrc/cod_metrics.hip) -- not as a transcription of the class hierarchy.

The weighted F-measure needs scipy.ndimage.distance_transform_edt(return_indices=True), a compiled third-party routine
(scipy 1.15): exact Euclidean distance to the nearest foreground pixel and that pixel's index.  Its tie rule is not documented;
probing it (26 k tied pixels on random masks) shows it returns, among equidistant foreground pixels, the one with the smallest
column and then the smallest row.  ``nearest_foreground`` below restates that by brute force.

Pinned by tests/golden/g16_cod_metrics.npz (the real class on 18 seeded cases; tests/test_oracle_golden.py).
"""
import numpy as np

EPS = np.spacing(1)


def prepare(pred, gt):
    """_prepare_data (:125-133): gt min-max normalised then > 0.5; pred min-max normalised, or truncated to int when constant."""
    pred, gt = np.asarray(pred, np.float64), np.asarray(gt, np.float64)
    if gt.max() != gt.min():
        gt = (gt - gt.min()) / (gt.max() - gt.min())
    g = gt > 0.5
    if pred.max() != pred.min():
        p = (pred - pred.min()) / (pred.max() - pred.min())
    else:
        p = pred.astype(int).astype(np.float64)           # the reference continues with an INTEGER array here, see weighted_f
    return p, g


def is_constant(pred):
    pred = np.asarray(pred, np.float64)
    return bool(pred.max() == pred.min())


def _round_half_even(v):
    return float(np.round(v))


def s_measure(p, g, alpha=0.5):
    n = g.size
    y = g.mean()
    if y == 0:
        return 1.0 - p.mean()
    if y == 1:
        return p.mean()

    def s_object(vals):                                   # 2x / (x^2 + 1 + sigma + eps), sigma = sample std (ddof 1)
        x, sd = vals.mean(), vals.std(ddof=1)
        return 2 * x / (x * x + 1 + sd + EPS)

    obj = y * s_object(p[g]) + (1 - y) * s_object(1 - p[~g])
    h, w = g.shape
    rows, cols = np.nonzero(g)
    cx, cy = int(_round_half_even(cols.mean())) + 1, int(_round_half_even(rows.mean())) + 1
    gf = g.astype(np.float64)

    def ssim(a, b):
        m = a.size
        xa, xb = a.mean(), b.mean()
        va, vb, cab = ((a - xa) ** 2).sum() / (m - 1), ((b - xb) ** 2).sum() / (m - 1), ((a - xa) * (b - xb)).sum() / (m - 1)
        al, be = 4 * xa * xb * cab, (xa * xa + xb * xb) * (va + vb)
        return al / (be + EPS) if al != 0 else (1.0 if be == 0 else 0.0)

    quads = ((slice(0, cy), slice(0, cx)), (slice(0, cy), slice(cx, w)), (slice(cy, h), slice(0, cx)), (slice(cy, h), slice(cx, w)))
    w1, w2, w3 = cx * cy / n, cy * (w - cx) / n, (h - cy) * cx / n
    weights = (w1, w2, w3, 1 - w1 - w2 - w3)
    with np.errstate(all="ignore"):
        reg = sum(wt * ssim(p[q], gf[q]) for wt, q in zip(weights, quads))
    return max(0.0, alpha * obj + (1 - alpha) * reg)


def _enhanced_alignment(fg_fg, fg_bg, n_fg_gt, n):
    """E-measure value(s) from the counts of predicted-foreground pixels inside / outside the ground truth (:354-377,379-411)."""
    fg_fg, fg_bg = np.asarray(fg_fg, np.float64), np.asarray(fg_bg, np.float64)
    pred_fg = fg_fg + fg_bg
    pred_bg = n - pred_fg
    if n_fg_gt == 0:
        total = pred_bg
    elif n_fg_gt == n:
        total = pred_fg
    else:
        bg_fg = n_fg_gt - fg_fg
        bg_bg = pred_bg - bg_fg
        mp, mg = pred_fg / n, n_fg_gt / n
        total = 0.0
        for cnt, dp, dg in ((fg_fg, 1 - mp, 1 - mg), (fg_bg, 1 - mp, -mg), (bg_fg, -mp, 1 - mg), (bg_bg, -mp, -mg)):
            align = 2 * dp * dg / (dp * dp + dg * dg + EPS)
            total = total + (align + 1) ** 2 / 4 * cnt
    return total / (n - 1 + EPS)


def _histograms(p, g):
    u8 = (p * 255).astype(np.uint8)
    return np.bincount(u8[g], minlength=256), np.bincount(u8[~g], minlength=256)


def e_measure(p, g):
    n, n_fg = g.size, int(np.count_nonzero(g))
    thr = min(2 * p.mean(), 1.0)
    b = p >= thr
    adp = float(_enhanced_alignment(np.count_nonzero(b & g), np.count_nonzero(b & ~g), n_fg, n))
    fh, bh = _histograms(p, g)
    curve = _enhanced_alignment(np.cumsum(fh[::-1]), np.cumsum(bh[::-1]), n_fg, n)
    return adp, np.asarray(curve, np.float64) * np.ones(256)


def f_measure(p, g, beta=0.3):
    thr = min(2 * p.mean(), 1.0)
    b = p >= thr
    inter = np.count_nonzero(b & g)
    if inter == 0:
        adp = 0.0
    else:
        pre, rec = inter / np.count_nonzero(b), inter / np.count_nonzero(g)
        adp = (1 + beta) * pre * rec / (beta * pre + rec)
    fh, bh = _histograms(p, g)
    tp = np.cumsum(fh[::-1]).astype(np.float64)
    ps = tp + np.cumsum(bh[::-1])
    ps[ps == 0] = 1
    precision, recall = tp / ps, tp / max(np.count_nonzero(g), 1)
    num = (1 + beta) * precision * recall
    curve = num / np.where(num == 0, 1, beta * precision + recall)
    return adp, curve, precision, recall


def nearest_foreground(g):
    """(distance, row index, col index) of the nearest True pixel for every pixel; ties: smallest column, then smallest row
    (the observed behaviour of scipy.ndimage.distance_transform_edt(~g, return_indices=True))."""
    h, w = g.shape
    fr, fc = np.nonzero(g)
    order = np.lexsort((fr, fc))                          # candidates sorted by (col, row): argmin returns the first minimum
    fr, fc = fr[order], fc[order]
    yy, xx = np.mgrid[0:h, 0:w]
    d2 = (yy[..., None] - fr) ** 2 + (xx[..., None] - fc) ** 2
    k = d2.argmin(axis=-1)
    return np.sqrt(d2.min(axis=-1).astype(np.float64)), fr[k], fc[k]


def gauss7():
    ax = np.arange(-3, 4, dtype=np.float64)
    k = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / 50.0)     # fspecial('gaussian', 7, 5)
    k[k < np.finfo(np.float64).eps * k.max()] = 0
    return k / k.sum()


def weighted_f(p, g, beta=1.0, integer_pred=False):
    """integer_pred: the prediction was constant, so _prepare_data handed on an int array; |pred - gt| is then an int array and
    scipy.ndimage.convolve returns an INT array for it -- the smoothed error is truncated toward zero (:520-528)."""
    if not g.any():
        return 0.0
    dist, ir, ic = nearest_foreground(g)
    gf = g.astype(np.float64)
    e = np.abs(p - gf)
    et = np.where(g, e, e[ir, ic])
    h, w = g.shape
    pad = np.zeros((h + 6, w + 6))
    pad[3:-3, 3:-3] = et
    k = gauss7()
    ea = np.zeros((h, w))
    for dy in range(7):
        for dx in range(7):
            ea += k[dy, dx] * pad[dy:dy + h, dx:dx + w]   # raster order over the 49 taps, as scipy accumulates them
    if integer_pred:
        ea = np.trunc(ea)
    m = np.where(g & (ea < e), ea, e)
    ew = m * np.where(g, 1.0, 2 - np.exp(np.log(0.5) / 5 * dist))
    tpw, fpw = gf.sum() - ew[g].sum(), ew[~g].sum()
    r, pr = 1 - ew[g].mean(), tpw / (tpw + fpw + EPS)
    return (1 + beta) * r * pr / (r + beta * pr + EPS)


def image_measures(pred, gt):
    """Everything one ``statistics.step`` appends for one image."""
    p, g = prepare(pred, gt)
    gf = g.astype(np.float64)
    inter, union = np.count_nonzero((p != 0) & g), np.count_nonzero((p != 0) | g)
    adp_em, em_curve = e_measure(p, g)
    adp_fm, fm_curve, pc, rc = f_measure(p, g)
    return dict(mae=np.abs(p - gf).mean(), acc=np.count_nonzero(p == gf) / g.size, iou=(inter / union) if union else 1.0,
                sm=s_measure(p, g), wfm=weighted_f(p, g, integer_pred=is_constant(pred)), adp_em=adp_em, em_curve=em_curve, adp_fm=adp_fm, fm_curve=fm_curve,
                p_curve=pc, r_curve=rc)


def aggregate(per_image):
    """statistics.get_result (:59-74): means over images; E/F max and mean over the 256 thresholds of the mean curves."""
    mean = lambda k: float(np.mean([m[k] for m in per_image]))  # noqa: E731
    em = np.mean([m["em_curve"] for m in per_image], axis=0)
    fm = np.mean([m["fm_curve"] for m in per_image], axis=0)
    return {"ACC": mean("acc"), "mIOU": mean("iou"), "E_MAX": em.max(), "E_MEAN": em.mean(), "F_MAX": fm.max(), "F_MEAN": fm.mean(),
            "SMeasure": mean("sm"), "MAE": mean("mae"), "WFM": mean("wfm")}
