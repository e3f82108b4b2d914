"""TEST INFRASTRUCTURE ONLY — CPU restatement of the RAM-Net losses and metrics.

Reference (relative to /root/reference/RAM_Net):
  * scale_invariant_loss ....... model/loss.py:6-9     (pinned by golden vectors)
  * scale_invariant_log_loss ... model/loss.py:12-15   (pinned: tests/golden/loss_extra.npz)
  * mse_loss ................... model/loss.py:18-19    (pinned)
  * trainer loss assembly with the half-resolution mse term ... trainer/lstm_trainer.py:152-226 (pinned: loss_extra.npz)
  * MultiScaleGradient ......... model/loss.py:22-70   (PARITY UNPINNED, see below)
  * abs_rel_diff & friends ..... model/metric.py:8-33  (pinned)
  * prepare_depth_data ......... evaluation.py:74-96   (pinned)
"""
import numpy as np
import torch
import torch.nn.functional as F


def scale_invariant_loss(y_input, y_target, weight=1.0, n_lambda=1.0):
    """w * (mean(d^2) - lambda * mean(d)^2) over the non-NaN pixels of d = input - target."""
    d = y_input - y_target
    d = d[~torch.isnan(d)]
    return weight * ((d * d).mean() - n_lambda * d.mean() ** 2)


def si_loss_grad(y_input, y_target, weight=1.0, n_lambda=1.0):
    """Closed-form d loss / d y_input (what the HIP backward kernel implements)."""
    d = y_input - y_target
    valid = ~torch.isnan(d)
    n = valid.sum().to(d.dtype)
    mean = d[valid].sum() / n
    g = weight * (2.0 * d / n - 2.0 * n_lambda * mean / n)
    return torch.where(valid, g, torch.zeros_like(g))


def mse_loss(y_input, y_target):
    m = ~torch.isnan(y_target)
    return F.mse_loss(y_input[m], y_target[m])


def scale_invariant_log_loss(y_input, y_target, n_lambda=1.0):
    """model/loss.py:12-15: the scale-invariant statistic on d = log(input) - log(target), NaN entries of d dropped."""
    d = torch.log(y_input) - torch.log(y_target)
    d = d[~torch.isnan(d)]
    return (d * d).mean() - n_lambda * d.mean() ** 2


def mse_loss_downsampled(y_input, y_target, downsampling_factor=0.5):
    """The trainer's mse term (lstm_trainer.py:169-185): both maps through F.interpolate(bilinear, align_corners=False,
    recompute_scale_factor=False) when the factor is not 1, then mse_loss over the non-NaN target entries."""
    if downsampling_factor != 1.0:
        y_target = F.interpolate(y_target, scale_factor=downsampling_factor, mode="bilinear", align_corners=False, recompute_scale_factor=False)
        y_input = F.interpolate(y_input, scale_factor=downsampling_factor, mode="bilinear", align_corners=False, recompute_scale_factor=False)
    return mse_loss(y_input, y_target)


def total_batch_loss(preds, targets, weights, L, loss="scale_invariant_loss", loss_params=None, mse=None):
    """calculate_losses + calculate_total_batch_loss (lstm_trainer.py:152-226) for ONE supervised key: `preds` / `targets` / `weights`
    list the L*terms supervised maps; loss = sum(w * nominal) / L [+ mse_weight * sum(w * mse_ds) / L]."""
    fn = {"scale_invariant_loss": scale_invariant_loss, "scale_invariant_log_loss": scale_invariant_log_loss, "mse_loss": mse_loss}[loss]
    total = sum(w * fn(p, t, **(loss_params or {})) for p, t, w in zip(preds, targets, weights)) / float(L)
    if mse is not None:
        total = total + mse.get("weight", 1.0) * sum(w * mse_loss_downsampled(p, t, mse.get("downsampling_factor", 0.5))
                                                     for p, t, w in zip(preds, targets, weights)) / float(L)
    return total


def spatial_gradient(x):
    """kornia==0.4.0 ``spatial_gradient(x, mode='sobel', order=1, normalized=True)``.

    PARITY UNPINNED: kornia 0.4.0 (requirements.txt:32) is not under /root/reference and is not
    installed in this image.  Published algorithm restated here: per-channel cross-correlation with
    Sobel-x [[-1,0,1],[-2,0,2],[-1,0,1]] / 8 and its transpose, replicate padding of 1 pixel,
    output B x C x 2 x H x W (index 0 = d/dx, index 1 = d/dy).
    """
    b, c, h, w = x.shape
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]], dtype=x.dtype) / 8.0
    k = torch.stack([kx, kx.t()])[:, None]                      # 2 x 1 x 3 x 3
    xp = F.pad(x.reshape(b * c, 1, h, w), (1, 1, 1, 1), mode="replicate")
    return F.conv2d(xp, k).reshape(b, c, 2, h, w)


def multi_scale_grad_loss(prediction, target, start_scale=1, num_scales=4):
    """MultiScaleGradient.forward (loss.py:34-62), non-preview branch.  PARITY UNPINNED (kornia)."""
    diff = prediction - target
    total = 0
    for s in range(num_scales):
        k = start_scale * (2 ** s)
        g = spatial_gradient(F.avg_pool2d(diff, k, k))
        ok = ~torch.isnan(g)
        total = total + torch.abs(g[ok]).sum() / ok.sum() * target.shape[0] * 2
    return total / num_scales


# --------------------------------------------------------------------------- numpy metrics
def abs_rel_diff(y_input, y_target, eps=1e-6):
    a = np.abs(y_target - y_input)
    return (a[~np.isnan(a)] / (y_target[~np.isnan(y_target)] + eps)).mean()


def squ_rel_diff(y_input, y_target, eps=1e-6):
    a = np.abs(y_target - y_input)
    ok = ~np.isnan(a)
    return (a[ok] ** 2 / (y_target[ok] ** 2 + eps)).mean()


def rms_linear(y_input, y_target):
    a = np.abs(y_target - y_input)
    return np.sqrt((a[~np.isnan(a)] ** 2).mean())


def scale_invariant_error(y_input, y_target):
    a = np.abs(y_target - y_input)
    a = a[~np.isnan(a)]
    return (a ** 2).mean() - a.mean() ** 2


def mean_error(y_input, y_target):
    a = np.abs(y_target - y_input)
    return a[~np.isnan(a)].mean()


def median_error(y_input, y_target):
    a = np.abs(y_target - y_input)
    return np.median(a[~np.isnan(a)])


def prepare_depth_data(target, prediction, clip_distance, reg_factor):
    """Normalised log depth -> metric depth (evaluation.py:74-96, down_scale_factor == 1)."""
    prediction = np.exp(reg_factor * (prediction - np.float32(1.0))).astype(np.float32)
    target = np.exp(reg_factor * (target - np.float32(1.0))).astype(np.float32)
    target = target * clip_distance
    prediction = prediction * clip_distance
    prediction = np.clip(prediction, np.exp(-1 * reg_factor) * clip_distance, clip_distance)
    return target, prediction


def evaluation_table(target, prediction, mask, eps=1e-5):
    """The ten rows one frame adds to the evaluation table (evaluation.py:201-241: `add_to_metrics` on metric depths).  The delta
    thresholds, RMS_log and median_diff are plain numpy means / medians over the mask (a NaN target inside makes the last two NaN,
    and counts as a miss of every threshold); the other six are the NaN-skipping functions of model/metric.py."""
    t, p = target[mask], prediction[mask]
    with np.errstate(invalid="ignore"):
        ratio = np.maximum(t / (p + eps), p / (t + eps))
        out = {"threshold_delta_1.25": np.mean(ratio <= 1.25), "threshold_delta_1.25^2": np.mean(ratio <= 1.25 ** 2),
               "threshold_delta_1.25^3": np.mean(ratio <= 1.25 ** 3)}
        lt, lp = np.log(t + eps), np.log(p + eps)
        out.update({"abs_rel_diff": abs_rel_diff(p, t), "squ_rel_diff": squ_rel_diff(p, t), "RMS_linear": rms_linear(p, t),
                    "RMS_log": np.sqrt(((lt - lp) ** 2).mean()), "SILog": scale_invariant_error(lp, lt), "mean_depth_error": mean_error(p, t),
                    "median_diff": np.abs(np.median(t) - np.median(p))})
    return out


def depth_to_normalised_log(depth, clip_distance, reg_factor):
    """Target normalisation of data_loader/dataset.py:296-305."""
    f = np.clip(depth, 0.0, clip_distance) / clip_distance
    with np.errstate(divide="ignore"):
        f = 1.0 + np.log(f) / reg_factor
    return f.clip(0, 1.0)
