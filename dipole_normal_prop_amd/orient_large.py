"""Entry point kept under the reference's name (orient_large.py there): orientation of large clouds through
<= 500 representatives per patch.  `run(opts)` = pipeline.orient_representatives; flags in options.py."""
import torch

from . import options, pipeline

torch.manual_seed(1)
run = pipeline.orient_representatives

if __name__ == "__main__":
    options.main(run)
