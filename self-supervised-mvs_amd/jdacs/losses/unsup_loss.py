"""Drop-in for jdacs/losses/unsup_loss.py: ``UnSupLoss()(imgs, cams, depth)``.

Same call signature, same value and the same attributes after a call (``reconstr_loss``, ``ssim_loss``, ``smooth_loss``,
``unsup_loss``) as the reference class (unsup_loss.py:19-83); the ~60 indexing / elementwise launches per view of the
reference's ``inverse_warping`` + ``compute_reconstr_loss`` + ``SSIM`` + ``depth_smoothness`` + top-3 selection are three
HIP launches forward and two backward (csrc/unsup_loss.hip).  ``smooth_lambda`` is the reference's ``args.smooth_lambda``
(config.py:46, default 1.0) as a constructor argument instead of a module-level argparse global."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops


class UnSupLoss(nn.Module):
    def __init__(self, smooth_lambda: float = 1.0):
        super().__init__()
        self.smooth_lambda = float(smooth_lambda)

    def forward(self, imgs, cams, depth):
        """imgs [B,N,3,H,W], cams [B,N,2,4,4] (extrinsic, intrinsic at quarter resolution), depth [B,H/4,W/4]."""
        if imgs.dim() != 5 or cams.dim() != 5 or imgs.shape[1] != cams.shape[1]:
            raise ValueError("Different number of images and projection matrices: imgs %s cams %s"
                             % (tuple(imgs.shape), tuple(cams.shape)))
        b, n = imgs.shape[:2]
        if n < 4:
            raise ValueError("UnSupLoss selects the 3 best of the N-1 source views (unsup_loss.py:76): needs N >= 4, got %d" % n)
        with torch.no_grad():
            # F.interpolate(scale_factor=0.25, bilinear) of every view in one call, then NHWC (unsup_loss.py:36-37,53-54)
            q = F.interpolate(imgs.reshape(b * n, *imgs.shape[2:]), scale_factor=0.25, mode="bilinear")
            q = q.permute(0, 2, 3, 1).reshape(b, n, q.shape[2], q.shape[3], 3)
            kinv, proj = ops.unsup_view_transforms(cams.float())
        if tuple(depth.shape) != (b, q.shape[2], q.shape[3]):
            raise ValueError("depth must be [B,H/4,W/4] = %s, got %s" % ((b, q.shape[2], q.shape[3]), tuple(depth.shape)))
        total, reconstr, ssim, smooth = ops.unsup_loss(depth, q[:, 0], [q[:, v] for v in range(1, n)], kinv, proj,
                                                       self.smooth_lambda)
        self.reconstr_loss, self.ssim_loss, self.smooth_loss = reconstr, ssim, smooth
        self.unsup_loss = total
        return total
