"""On-device counterpart of datasets/utils/dataset_generator.py:26-39.

``SlicedDatasetGenerator.generate_item(points, target_partition_points)`` keeps the reference's signature for one
cloud; ``generate_batch`` slices a whole (B,N,3) batch in one launch.  Same law as the reference (the first accepted
plane of an i.i.d. sequence of planes through three uniform points, the reference's own plane formula); by default
the draws are Philox draws on the device instead of numpy's global generator.  ``planes=`` takes the candidate
(params, bias) rows from the caller — with numpy's sequence the split is the reference's, bit for bit (float64
classification as HyperPlane.check_point, tests/golden/slicer.npz).
"""
import torch

from ...ops import slice_clouds


class SlicedDatasetGenerator(object):

    @staticmethod
    def generate_batch(points, target_partition_points=1024, seed=0, planes=None):
        part, rest, _ = slice_clouds(points, target_partition_points, seed, planes=planes)
        return part, rest

    @staticmethod
    def generate_item(points, target_partition_points=1024, seed=0, planes=None):
        pts = torch.as_tensor(points, dtype=torch.float32)
        if not pts.is_cuda:
            pts = pts.cuda()
        part, rest = SlicedDatasetGenerator.generate_batch(pts.unsqueeze(0), target_partition_points, seed, planes=planes)
        return part[0], rest[0]
