"""Entry point kept under the reference's name (reference_orientation.py there): orientation transfer from an
oriented reference cloud.  `run(opts)` = pipeline.transfer_reference."""
import argparse
from pathlib import Path

from . import pipeline

run = pipeline.transfer_reference


def get_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser("orientation transfer")
    for flag, helptext in (("input", "cloud to orient (3 or 6 columns)"), ("reference", "oriented cloud (6 columns)"),
                           ("output", "where to write the result")):
        parser.add_argument("--" + flag, type=Path, required=True, help=helptext)
    parser.add_argument("--n", type=int, default=30, help="size of knn for normal estimation")
    parser.add_argument("--estimate_normals", action="store_true",
                        help="estimate PCA normals for a 3-column input instead of taking the field direction")
    return parser


if __name__ == "__main__":
    run(get_parser().parse_args())
