"""Stand-in for the ONE cv2 function the reference's training step uses (training/loss.py:156): TEST INFRASTRUCTURE, lets the
unmodified `training.loss.StyleGAN2Loss` import and run offline (opencv is absent)."""
import numpy as np


def boundingRect(mask):
    """(x, y, w, h) of the non-zero pixels of a 2-D array; (0, 0, 0, 0) when there are none."""
    ys, xs = np.nonzero(np.asarray(mask))
    if ys.size == 0:
        return 0, 0, 0, 0
    return int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1)
