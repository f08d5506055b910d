"""Validation metrics of the shipped option files on the device (csrc/metrics.hip through the C ABI):
psnr / ssim (basicsr.metrics, esrgan_s2naip_urban.yml:153-162) and cpsnr (/root/reference/ssr/metrics/cpsnr.py:8-59).

Images are uint8 [B, H, W, C] device tensors as `tensor2img_u8` produces them (BasicSR's tensor2img quantisation: clamp to
[0, 1], * 255, round half to even; kept RGB — the three metrics are invariant to the channel order when test_y_channel is false)."""
from __future__ import annotations

import math

import torch

from . import hip


def tensor2img_u8(x: torch.Tensor, truncate: bool = False) -> torch.Tensor:
    """float32 NCHW on the device -> uint8 NHWC on the device.  truncate=True is the astype(uint8) of infer_grid.py:60-64."""
    x = x.contiguous().float()
    assert x.is_cuda and x.dim() == 4
    n, c, h, w = x.shape
    out = torch.empty(n, h, w, c, dtype=torch.uint8, device=x.device)
    hip.check(hip.lib().ssr_quantize_u8(x.data_ptr(), out.data_ptr(), n, c, h, w, 1 if truncate else 0, hip.stream_ptr()),
              "ssr_quantize_u8")
    return out


def _pair(a, b):
    assert a.shape == b.shape and a.dtype == torch.uint8 and b.dtype == torch.uint8 and a.is_cuda and b.is_cuda, \
        f"Image shapes are different: {tuple(a.shape)}, {tuple(b.shape)}."
    if a.dim() == 4:
        assert a.shape[0] == 1, "one image per call (the reference validates with batch size 1)"
        a, b = a[0], b[0]
    return a.contiguous(), b.contiguous()


def _no_y(test_y_channel):
    if test_y_channel:
        raise NotImplementedError("metrics with test_y_channel: true (no shipped option file uses it)")


def _shift_sums(a, b, crop, m):
    h, w, c = a.shape
    out = torch.empty((m + 1) ** 2, c, 2, dtype=torch.int64, device=a.device)
    hip.check(hip.lib().ssr_metric_shift_sums(a.data_ptr(), b.data_ptr(), h, w, c, crop, m, out.data_ptr(), hip.stream_ptr()),
              "ssr_metric_shift_sums")
    return out.cpu(), (h - 2 * crop - m) * (w - 2 * crop - m)


def calculate_psnr(img, img2, crop_border, input_order="HWC", test_y_channel=False, **kw) -> float:
    _no_y(test_y_channel)
    a, b = _pair(img, img2)
    s, n = _shift_sums(a, b, crop_border, 0)
    mse = int(s[0, :, 1].sum()) / (n * a.shape[2])
    return float("inf") if mse == 0 else 10.0 * math.log10(255.0 * 255.0 / mse)


def calculate_cpsnr(img, img2, crop_border, input_order="HWC", test_y_channel=False, **kw) -> float:
    """cpsnr.py:36-55: for each of the 81 relative offsets the per-channel mean difference is removed before the MSE:
    mean((d - mean d)^2) = (S2 - S1^2 / n) / n per channel, averaged over the channels."""
    _no_y(test_y_channel)
    a, b = _pair(img, img2)
    s, n = _shift_sums(a, b, crop_border, 8)
    s = s.double()
    mse = ((s[:, :, 1] - s[:, :, 0] ** 2 / n) / n).mean(dim=1)
    best = float(mse.min())
    return float("inf") if best == 0 else 10.0 * math.log10(255.0 * 255.0 / best)


def calculate_ssim(img, img2, crop_border, input_order="HWC", test_y_channel=False, **kw) -> float:
    _no_y(test_y_channel)
    a, b = _pair(img, img2)
    h, w, c = a.shape
    out = torch.empty(c, dtype=torch.float64, device=a.device)
    hip.check(hip.lib().ssr_metric_ssim_sums(a.data_ptr(), b.data_ptr(), h, w, c, crop_border, out.data_ptr(), hip.stream_ptr()),
              "ssr_metric_ssim_sums")
    n = (h - 2 * crop_border - 10) * (w - 2 * crop_border - 10)
    return float((out.cpu() / n).mean())


METRICS = {"calculate_psnr": calculate_psnr, "calculate_ssim": calculate_ssim, "calculate_cpsnr": calculate_cpsnr}


def imwrite_rgb(img_hwc_u8, path: str):
    """basicsr imwrite (cv2.imwrite of a BGR array) writes the same PNG pixels as saving the RGB array directly."""
    from PIL import Image
    Image.fromarray(img_hwc_u8).save(path)
