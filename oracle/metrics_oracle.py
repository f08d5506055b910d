"""TEST INFRASTRUCTURE — CPU (numpy, float64) restatement of the validation metrics the shipped option files name
(esrgan_s2naip_urban.yml:150-186).  Only tests/ may import this module.

  calculate_cpsnr   follows /root/reference/ssr/metrics/cpsnr.py:8-59 statement by statement.
  calculate_psnr / calculate_ssim   live in basicsr==1.4.2 (basicsr/metrics/psnr_ssim.py; /root/reference/requirements.txt:1), which is
      not on disk: restated from its published source — parity unpinned by the reference — and anchored on the option files
      (`crop_border: 4`, `test_y_channel: false`).  cv2 is absent too: cv2.getGaussianKernel(11, 1.5) and the 'valid' part of
      cv2.filter2D are written out.
Inputs: uint8 HWC images (what tensor2img hands to calculate_metric, ssr_esrgan_model.py:302-308)."""
import numpy as np


def _crop(img, crop_border):
    return img[crop_border:-crop_border, crop_border:-crop_border, ...] if crop_border != 0 else img


def calculate_psnr(img, img2, crop_border, **kw):
    a, b = _crop(img, crop_border).astype(np.float64), _crop(img2, crop_border).astype(np.float64)
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)


def _gauss(ksize=11, sigma=1.5):
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2
    k = np.exp(-(i * i) / (2 * sigma * sigma))
    return k / k.sum()


def _filter_valid(img, window):
    k = window.shape[0]
    h, w = img.shape
    out = np.zeros((h - k + 1, w - k + 1))
    for i in range(k):
        for j in range(k):
            out += window[i, j] * img[i:i + h - k + 1, j:j + w - k + 1]
    return out


def _ssim(img, img2):
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    g = _gauss()
    window = np.outer(g, g)
    mu1, mu2 = _filter_valid(img, window), _filter_valid(img2, window)     # filter2D(...)[5:-5, 5:-5]
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1 = _filter_valid(img ** 2, window) - mu1_sq
    s2 = _filter_valid(img2 ** 2, window) - mu2_sq
    s12 = _filter_valid(img * img2, window) - mu1_mu2
    return (((2 * mu1_mu2 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))).mean()


def calculate_ssim(img, img2, crop_border, **kw):
    a, b = _crop(img, crop_border).astype(np.float64), _crop(img2, crop_border).astype(np.float64)
    return float(np.array([_ssim(a[..., i], b[..., i]) for i in range(a.shape[2])]).mean())


def calculate_cpsnr(img, img2, crop_border, **kw):
    """ssr/metrics/cpsnr.py:8-59 (PROBA-V cPSNR): best PSNR over 9x9 relative translations with a per-channel brightness bias."""
    img1 = _crop(img, crop_border).astype(np.float64)
    img2 = _crop(img2, crop_border).astype(np.float64)
    max_offset = 8
    height, width = img1.shape[0], img1.shape[1]
    ch, cw = height - max_offset, width - max_offset
    best = None
    for ro in range(max_offset + 1):
        for co in range(max_offset + 1):
            c1 = img1[ro:, co:][0:ch, 0:cw].copy()
            c2 = img2[(max_offset - ro):, (max_offset - co):][0:ch, 0:cw].copy()
            for c in range(img1.shape[2]):
                c2[:, :, c] += np.mean(c1[:, :, c] - c2[:, :, c])
            mse = np.mean(np.square(c1 - c2))
            if best is None or mse < best:
                best = mse
    return float("inf") if best == 0 else 10.0 * np.log10(255.0 * 255.0 / best)
