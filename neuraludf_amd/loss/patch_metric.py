"""SSIM patch error with a full-patch Gaussian window (reference loss/patch_metric.py:9-41, 69-84):
one value per ray = sum over the 3 channels of (1 - ssim) / 2.  Runs in the `nudf_ssim_patch` HIP
kernel (forward + backward w.r.t. the predicted patch)."""
import math

import torch

from .._lib import call, ptr


def gaussian_window(h_patch_size, std=1.5):
    ws = 2 * h_patch_size + 1
    g = torch.tensor([math.exp(-(x - ws // 2) ** 2 / float(2 * std ** 2)) for x in range(ws)])
    g = g / g.sum()
    return (g[:, None] @ g[None, :]).reshape(-1).contiguous()


class _SSIMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, window):
        pred = pred.detach().contiguous()
        gt = gt.detach().contiguous()
        n, npx, _ = pred.shape
        out = torch.empty(n, device=pred.device)
        call("nudf_ssim_patch", ptr(pred), ptr(gt), ptr(window), n, npx, ptr(out), None, None)
        ctx.save_for_backward(pred, gt, window)
        return out

    @staticmethod
    def backward(ctx, d_out):
        pred, gt, window = ctx.saved_tensors
        n, npx, _ = pred.shape
        d_pred = torch.empty_like(pred)
        out = torch.empty(n, device=pred.device)
        call("nudf_ssim_patch", ptr(pred), ptr(gt), ptr(window), n, npx, ptr(out), ptr(d_out.contiguous()), ptr(d_pred))
        return d_pred, None, None


_win_cache = {}


def ssim_patch_error(pred, gt, h_patch_size):
    """pred, gt: [N, Npx, 3] -> [N]."""
    key = (h_patch_size, str(pred.device))
    if key not in _win_cache:
        _win_cache[key] = gaussian_window(h_patch_size).to(pred.device)
    return _SSIMFn.apply(pred, gt, _win_cache[key])
