"""Command-line surface of the orient_* entry points: the reference's flag names and defaults
(options.py:5-20 there) from one table, plus the opts.txt dump its scripts write (options.py:23-32)."""
import argparse
from pathlib import Path

# (flag, kwargs) - order as in `--help` of the reference
_FLAGS = (
    ("export_dir", dict(type=Path, required=True, help="export directory")),
    ("propagation_iters", dict(type=int, default=10, help="run a propagation every this many voting rounds")),
    ("number_parts", dict(type=int, default=15, help="voxel grid resolution per axis")),
    ("minimum_points_per_patch", dict(type=int, default=21, help="merge voxels with fewer points")),
    ("curvature_threshold", dict(type=float, default=0.0, help="flatness ratio below which a patch is not voted on")),
    ("pc", dict(type=Path, required=True, help="pc to read")),
    ("models", dict(type=Path, nargs="+", default=[],
                    help="PointCNN checkpoints of the reference; the network step is outside this package, "
                         "a non-empty list is rejected")),
    ("iters", dict(type=int, default=100, help="voting rounds (they only schedule propagations here)")),
    ("diffuse", dict(action="store_true", help="let every oriented patch act on all points, not only the pending ones")),
    ("weighted_prop", dict(action="store_true", help="scale dipoles by the vote confidence")),
    ("estimate_normals", dict(action="store_true", help="estimate unoriented PCA normals first")),
    ("n", dict(type=int, default=30, help="size of knn for normal estimation")),
)


def get_parser(name="Base Options") -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(name)
    for flag, kwargs in _FLAGS:
        parser.add_argument("--" + flag, **kwargs)
    return parser


def export_options(opts) -> None:
    """`key: value` per line, no trailing newline, into <export_dir>/opts.txt."""
    lines = [f"{key}: {value}" for key, value in vars(opts).items()]
    (Path(opts.export_dir) / "opts.txt").write_text("\n".join(lines))


def reject_models(opts) -> None:
    if getattr(opts, "models", None):
        raise SystemExit("--models: the PointCNN voting step (torch_geometric) is not part of this package; "
                         "run without --models to orient with the dipole propagation alone")


def main(flow, parser=None):
    """Shared `__main__` of the entry modules: parse, create the export directory, dump opts.txt, run."""
    opts = (parser or get_parser()).parse_args()
    Path(opts.export_dir).mkdir(exist_ok=True, parents=True)
    export_options(opts)
    return flow(opts)
