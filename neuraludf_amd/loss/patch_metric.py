"""Per-ray patch errors of ColorPatchLoss (reference loss/loss.py:66-73, loss/patch_metric.py:9-67): 'ssim' (sum over
the 3 channels of (1 - ssim) / 2, full-patch Gaussian window), 'l1', 'ssd' and 'ncc' (1 - mean-channel NCC).  All run
in the `nudf_patch_metric` HIP kernels (forward + backward w.r.t. the predicted patch)."""
import math

import torch

from .._lib import call, ptr


def gaussian_window(h_patch_size, std=1.5):
    ws = 2 * h_patch_size + 1
    g = torch.tensor([math.exp(-(x - ws // 2) ** 2 / float(2 * std ** 2)) for x in range(ws)])
    g = g / g.sum()
    return (g[:, None] @ g[None, :]).reshape(-1).contiguous()


PATCH_TYPES = {"ssim": 0, "l1": 1, "ssd": 2, "ncc": 3}


class _PatchMetricFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, window, kind):
        pred = pred.detach().contiguous()
        gt = gt.detach().contiguous()
        n, npx, _ = pred.shape
        out = torch.empty(n, device=pred.device)
        call("nudf_patch_metric", kind, ptr(pred), ptr(gt), ptr(window), n, npx, ptr(out), None, None)
        ctx.save_for_backward(pred, gt, window)
        ctx.kind = kind
        return out

    @staticmethod
    def backward(ctx, d_out):
        pred, gt, window = ctx.saved_tensors
        n, npx, _ = pred.shape
        d_pred = torch.empty_like(pred)
        out = torch.empty(n, device=pred.device)
        call("nudf_patch_metric", ctx.kind, ptr(pred), ptr(gt), ptr(window), n, npx, ptr(out), ptr(d_out.contiguous()),
             ptr(d_pred))
        return d_pred, None, None, None


_win_cache = {}


def patch_error(pred, gt, h_patch_size, kind="ssim"):
    """pred, gt: [N, Npx, 3] -> [N]; kind in PATCH_TYPES."""
    key = (h_patch_size, str(pred.device))
    if key not in _win_cache:
        _win_cache[key] = gaussian_window(h_patch_size).to(pred.device)
    return _PatchMetricFn.apply(pred, gt, _win_cache[key], PATCH_TYPES[kind])


def ssim_patch_error(pred, gt, h_patch_size):
    return patch_error(pred, gt, h_patch_size, "ssim")
