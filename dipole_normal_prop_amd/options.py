"""Command-line flags of the orient_* entry scripts - the same flag names, defaults and opts.txt
dump as the reference's options.py:5-32."""
import argparse
from pathlib import Path


def get_parser(name='Base Options') -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(name)
    p.add_argument('--export_dir', type=Path, required=True, help='export directory')
    p.add_argument('--propagation_iters', default=10, type=int, help='test epochs')
    p.add_argument('--number_parts', type=int, default=15)
    p.add_argument('--minimum_points_per_patch', type=int, default=21)
    p.add_argument('--curvature_threshold', default=0.0, type=float)
    p.add_argument('--pc', type=Path, required=True, help='pc to read')
    p.add_argument('--models', nargs='+', type=Path, default=[],
                   help='PointCNN checkpoints of the reference; the network step is outside this '
                        'package, a non-empty list is rejected')
    p.add_argument('--iters', default=100, type=int, help='iters to optimize')
    p.add_argument('--diffuse', action='store_true')
    p.add_argument('--weighted_prop', action='store_true')
    p.add_argument('--estimate_normals', action='store_true')
    p.add_argument('--n', type=int, default=30, help='size of knn for normal estimation')
    return p


def export_options(opts):
    text = '\n'.join(f'{k}: {v}' for k, v in opts.__dict__.items())
    with open(opts.export_dir / 'opts.txt', 'w+') as fh:
        fh.write(text)


def reject_models(opts):
    if getattr(opts, 'models', None):
        raise SystemExit('--models: the PointCNN voting step (torch_geometric) is not part of this package; '
                         'run without --models to orient with the dipole propagation alone')
