"""Mirror of ``utils/ssim_psnr.py``: calculate_psnr / SSIM backed by the fused HIP metric kernel."""
from .. import ops


def calculate_psnr(img1, img2):
    return ops.psnr_ssim(img1, img2)[0]


class SSIM:
    def __init__(self, window_size=11, size_average=True):
        if window_size != 11 or not size_average:
            raise NotImplementedError("dpmn_amd SSIM: 11x11 window, size_average=True (utils/ssim_psnr.py:55)")

    def __call__(self, img1, img2):
        return ops.psnr_ssim(img1, img2)[1]
