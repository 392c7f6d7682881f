"""Statistical comparison of two independent Monte-Carlo renders of the same image (test infrastructure).

Used for the RNG-policy tolerance of north_star ("match the reference CPU render on identical RNG seeds within a
stated per-channel tolerance"): the reference's serial streams and the device's counter RNG give two independent
estimates, so the tolerance is the Monte-Carlo error itself, estimated per pixel from the accumulators both sides keep
(sum, sum2, count: main.cpp:94-99).  The pixel value is the mean over CONTRIBUTING samples, so its
variance is D / n with D = sum2/n - (sum/n)^2, exactly the "dispersion" the reference computes (main.cpp:169-172).
"""
import numpy as np

GAMMA = np.float32(1) / np.float32(2.2)   # config.h:25


def pixel_stats(s, s2, c):
    n = c.astype(np.float64)
    ok = n > 0
    nn = np.where(ok, n, 1.0)[:, None]
    mean = np.where(ok[:, None], s.astype(np.float64) / nn, 0.0)
    var = np.where(ok[:, None], np.maximum(s2.astype(np.float64) / nn - mean * mean, 0.0), 0.0)
    return n, mean, var


def compare(acc_a, acc_b, min_count=12, pooled=True):
    """acc = (sum [n,3], sum2 [n,3], count [n]).  Returns per-channel statistics of the difference of the two renders.

    z            (mean_a - mean_b) / sqrt(D (1/n_a + 1/n_b)) over pixels with >= min_count contributing samples on both
                 sides and a non-zero predicted error: mean ~ 0 and rms ~ 1 if both sample the same distribution
                 (D = the pooled per-pixel variance; pooled=False uses D_a/n_a + D_b/n_b)
    rmse_mean    measured RMSE of the per-pixel means, next to the value predicted from the variances
    rmse_image   the same for the resolved float image pow(mean, gamma) * 255 (main.cpp:179-182), prediction by the delta
                 method; only pixels with mean > 0.02 enter (the gamma curve's slope is unbounded at 0)
    """
    na, ma, va = pixel_stats(*acc_a)
    nb, mb, vb = pixel_stats(*acc_b)
    if pooled:
        # Under the hypothesis being tested both sides sample ONE distribution per pixel, so its variance is best
        # estimated from both sets of accumulators together (less noisy than two separate estimates from few samples).
        _, _, vp = pixel_stats(acc_a[0].astype(np.float64) + acc_b[0], acc_a[1].astype(np.float64) + acc_b[1],
                               acc_a[2].astype(np.int64) + acc_b[2])
        va = vb = vp
    both = (na >= min_count) & (nb >= min_count)
    out = {"pixels": int(acc_a[2].size), "pixels_compared": int(both.sum()), "channels": []}
    for k in range(3):
        pred_var = va[:, k] / np.maximum(na, 1) + vb[:, k] / np.maximum(nb, 1)
        use = both & (pred_var > 1e-12)
        d = ma[use, k] - mb[use, k]
        z = d / np.sqrt(pred_var[use])
        bright = use & (ma[:, k] > 0.02) & (mb[:, k] > 0.02)
        ia, ib = ma[bright, k] ** float(GAMMA) * 255.0, mb[bright, k] ** float(GAMMA) * 255.0
        mid = 0.5 * (ma[bright, k] + mb[bright, k])
        slope = 255.0 * float(GAMMA) * mid ** (float(GAMMA) - 1.0)
        out["channels"].append({
            "n": int(use.sum()), "z_mean": float(z.mean()), "z_rms": float(np.sqrt((z * z).mean())),
            "z_abs_max": float(np.abs(z).max()),
            "rmse_mean": float(np.sqrt((d * d).mean())), "rmse_mean_predicted": float(np.sqrt(pred_var[use].mean())),
            "n_image": int(bright.sum()),
            "rmse_image": float(np.sqrt(((ia - ib) ** 2).mean())),
            "rmse_image_predicted": float(np.sqrt((slope * slope * pred_var[bright]).mean()))})
    # pixels without variance on either side (the directly visible light: every sample contributes the same value)
    flat = (na > 0) & (nb > 0) & (va.sum(1) < 1e-12) & (vb.sum(1) < 1e-12)
    out["flat_pixels"] = int(flat.sum())
    out["flat_pixels_max_abs_diff"] = float(np.abs(ma[flat] - mb[flat]).max()) if flat.any() else 0.0
    # contributing fraction: binomial counts of the whole frame
    ca, cb = float(acc_a[2].sum()), float(acc_b[2].sum())
    out["contributing"] = [ca, cb]
    out["contributing_z"] = (ca - cb) / np.sqrt(ca + cb)
    return out


def compare_blocks(acc_a, acc_b, width, height, block, pooled=True, min_pixel_fraction=1.0):
    """The same comparison on the image BINNED into block x block pixels: a bin's value is the mean of its pixels' means (the
    linear image box-filtered to 1 / block of its resolution), resolved like a pixel (pow(mean, gamma) * 255, main.cpp:179-182).
    A bin averages block^2 independent pixel estimates, so the Monte-Carlo error of the difference falls by the factor
    `block` and what remains is a number small enough to mean something to a reader: per channel the RMSE of the resolved
    float image over the bins, next to the RMSE predicted from the per-pixel variances (delta method), the z-scores of the bins,
    and the largest absolute difference of a bin.  A bin's mean runs over the pixels that have a contributing sample in BOTH
    renders; bins in which fewer than min_pixel_fraction of the pixels do (default: any pixel missing), or whose mean is below
    0.02 (the gamma curve's slope is unbounded at 0), do not enter.  `brightness_z` is the frame-wide
    bin: (sum of the pixel means of a - of b) / predicted error -- a bias between the policies shows there first."""
    na, ma, va = pixel_stats(*acc_a)
    nb, mb, vb = pixel_stats(*acc_b)
    if pooled:
        _, _, vp = pixel_stats(acc_a[0].astype(np.float64) + acc_b[0], acc_a[1].astype(np.float64) + acc_b[1],
                               acc_a[2].astype(np.int64) + acc_b[2])
        va = vb = vp
    ok = (na > 0) & (nb > 0)
    pv = np.where(ok[:, None], va / np.maximum(na, 1)[:, None] + vb / np.maximum(nb, 1)[:, None], 0.0)
    hb, wb = height // block, width // block

    def bins(x):   # [H * W, c] -> [bins, block^2, c]; rows / columns beyond a whole number of bins are left out
        x = x.reshape(height, width, -1)[:hb * block, :wb * block]
        return x.reshape(hb, block, wb, block, -1).transpose(0, 2, 1, 3, 4).reshape(hb * wb, block * block, -1)

    w = bins(ok[:, None].astype(np.float64))                  # [bins, block^2, 1]
    cnt = np.maximum(w.sum(1), 1.0)                           # [bins, 1]
    ma_b, mb_b, pv_b = (bins(ma) * w).sum(1) / cnt, (bins(mb) * w).sum(1) / cnt, (bins(pv) * w).sum(1) / (cnt * cnt)
    whole = w.sum(1)[:, 0] >= min_pixel_fraction * block * block
    out = {"block": block, "bins": hb * wb, "channels": []}
    g = float(GAMMA)
    for k in range(3):
        use = whole & (pv_b[:, k] > 1e-14) & (ma_b[:, k] > 0.02) & (mb_b[:, k] > 0.02)
        d = ma_b[use, k] - mb_b[use, k]
        z = d / np.sqrt(pv_b[use, k])
        ia, ib = ma_b[use, k] ** g * 255.0, mb_b[use, k] ** g * 255.0
        slope = 255.0 * g * (0.5 * (ma_b[use, k] + mb_b[use, k])) ** (g - 1.0)
        tot = ok & (pv[:, k] > 0)
        out["channels"].append({
            "n": int(use.sum()), "z_mean": float(z.mean()), "z_rms": float(np.sqrt((z * z).mean())),
            "rmse_image": float(np.sqrt(((ia - ib) ** 2).mean())),
            "rmse_image_predicted": float(np.sqrt((slope * slope * pv_b[use, k]).mean())),
            "max_abs_diff_image": float(np.abs(ia - ib).max()),
            "brightness_z": float((ma[tot, k].sum() - mb[tot, k].sum()) / np.sqrt(pv[tot, k].sum()))})
    return out


# The stated tolerance: what "the two policies render the same image" means quantitatively.
TOLERANCE = {"z_rms": (0.93, 1.08), "z_mean_abs": 0.05, "rmse_ratio": 1.08, "contributing_z_abs": 5.0,
             # the image binned 8 x 8 and 16 x 16 (compare_blocks): RMSE of the resolved float image, per channel, in units of
             # 1/255 -- at most 1.1 x (1.15 x: a quarter of the bins) what the variances predict AND at most the absolute figure
             # (64-seed fixture, 32 768 samples per pixel: measured 0.98 / 0.99 / 1.10 and 0.51 / 0.48 / 0.61 on the MI355X)
             "blocks": {8: {"rmse_ratio": 1.10, "rmse_image_max": 1.5}, 16: {"rmse_ratio": 1.15, "rmse_image_max": 0.8}},
             # the same per PIXEL (compare: rmse_image), in units of 1/255: measured 8.2 / 8.2 / 9.2
             "rmse_image_pixel_max": 11.0,
             "brightness_z_abs": 3.5}


def assert_same_image(r, tol=TOLERANCE):
    for k, ch in enumerate(r["channels"]):
        assert ch["n"] > 0.5 * r["pixels"], (k, ch)
        assert tol["z_rms"][0] < ch["z_rms"] < tol["z_rms"][1], (k, ch)
        assert abs(ch["z_mean"]) < tol["z_mean_abs"], (k, ch)
        assert ch["rmse_mean"] < tol["rmse_ratio"] * ch["rmse_mean_predicted"], (k, ch)
        assert ch["rmse_image"] < tol["rmse_ratio"] * ch["rmse_image_predicted"], (k, ch)
        assert ch["rmse_image"] < tol["rmse_image_pixel_max"], (k, ch)
    assert abs(r["contributing_z"]) < tol["contributing_z_abs"], r["contributing"]
    assert r["flat_pixels_max_abs_diff"] < 1e-6, r


def assert_same_binned_image(rb, tol=TOLERANCE):
    """rb = compare_blocks(...) for a block size named in tol["blocks"]."""
    t = tol["blocks"][rb["block"]]
    for k, ch in enumerate(rb["channels"]):
        assert ch["n"] > 0.6 * rb["bins"], (k, ch)
        assert ch["rmse_image"] < t["rmse_ratio"] * ch["rmse_image_predicted"], (rb["block"], k, ch)
        assert ch["rmse_image"] < t["rmse_image_max"], (rb["block"], k, ch)
        assert abs(ch["brightness_z"]) < tol["brightness_z_abs"], (rb["block"], k, ch)
