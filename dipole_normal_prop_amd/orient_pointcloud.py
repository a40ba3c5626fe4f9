"""Entry point kept under the reference's name (orient_pointcloud.py there): patch-wise orientation.
`run(opts)` = pipeline.orient_patches; flags in options.py.

    python -m dipole_normal_prop_amd.orient_pointcloud --pc cloud.xyz --export_dir out --number_parts 30 \
        --minimum_points_per_patch 100 --diffuse
"""
import torch

from . import options, pipeline

torch.manual_seed(1)
run = pipeline.orient_patches

if __name__ == "__main__":
    options.main(run)
