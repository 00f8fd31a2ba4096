"""Stand-in for pytorch3d.ops.knn.knn_points (K=1 only), used to run the unmodified reference.
pytorch3d is un-vendored and unpinned (reference README.md:48); its K-NN returns SQUARED L2
distances and int64 indices. Distances here are evaluated as ((dx*dx)+(dy*dy))+(dz*dz) in fp32
(no FMA) and ties go to the lowest index -- tie-breaking in the real kernel is implementation
defined ("parity unpinned on ties")."""
import torch


def knn_points(p1, p2, K=1, **kw):
    assert K == 1
    B, N, _ = p1.shape
    d_out = torch.empty(B, N, 1, dtype=p1.dtype)
    i_out = torch.empty(B, N, 1, dtype=torch.long)
    chunk = 4096
    for b in range(B):
        q = p2[b]
        for s in range(0, N, chunk):
            x = p1[b, s:s + chunk]
            dx = x[:, None, 0] - q[None, :, 0]
            dy = x[:, None, 1] - q[None, :, 1]
            dz = x[:, None, 2] - q[None, :, 2]
            d2 = (dx * dx + dy * dy) + dz * dz
            m, i = torch.min(d2, dim=1)      # torch.min returns the first minimal index on CPU
            d_out[b, s:s + chunk, 0] = m
            i_out[b, s:s + chunk, 0] = i
    return d_out, i_out, None
