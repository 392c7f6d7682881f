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


# The stated tolerance: what "the two policies render the same image" means quantitatively.
TOLERANCE = {"z_rms": (0.93, 1.08), "z_mean_abs": 0.05, "rmse_ratio": 1.08, "contributing_z_abs": 5.0}


def assert_same_image(r, tol=TOLERANCE):
    for k, ch in enumerate(r["channels"]):
        assert ch["n"] > 0.5 * r["pixels"], (k, ch)
        assert tol["z_rms"][0] < ch["z_rms"] < tol["z_rms"][1], (k, ch)
        assert abs(ch["z_mean"]) < tol["z_mean_abs"], (k, ch)
        assert ch["rmse_mean"] < tol["rmse_ratio"] * ch["rmse_mean_predicted"], (k, ch)
        assert ch["rmse_image"] < tol["rmse_ratio"] * ch["rmse_image_predicted"], (k, ch)
    assert abs(r["contributing_z"]) < tol["contributing_z_abs"], r["contributing"]
    assert r["flat_pixels_max_abs_diff"] < 1e-6, r
