"""SparseConvTensor stand-in (see spconv/pytorch/__init__.py in this shim for the semantics)."""
import torch


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, **kw):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)

    def replace_feature(self, f):
        return SparseConvTensor(f, self.indices, self.spatial_shape, self.batch_size)

    def dense(self, channels_first=True):
        D, H, W = self.spatial_shape
        C = self.features.shape[1]
        out = torch.zeros(self.batch_size, D, H, W, C, dtype=self.features.dtype)
        i = self.indices.long()
        out[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = self.features
        return out.permute(0, 4, 1, 2, 3).contiguous() if channels_first else out
