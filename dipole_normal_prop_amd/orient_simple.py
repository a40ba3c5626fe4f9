"""Entry point kept under the reference's name (orient_simple.py there): per-point greedy propagation.
`run(opts)` = pipeline.orient_points; flags in options.py."""
import torch

from . import options, pipeline

torch.manual_seed(1)
run = pipeline.orient_points

if __name__ == "__main__":
    options.main(run)
