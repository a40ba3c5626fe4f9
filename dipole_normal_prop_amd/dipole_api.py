"""Importable API of the reference's dipole_api.py: orient_large(opts) (dipole_api.py:14-87) with its
own parser (dipole_api.py:101-132; the reference hard-codes Windows paths as defaults, here --pc and
--export_dir are required instead)."""
from . import options
from .orient_large import run as _run_large


def orient_large(opts):
    """Patch partition, <=500 representatives per patch, dipole propagation on the representatives,
    global flip by the mean potential, export to opts.export_dir/final_result.xyz."""
    return _run_large(opts)


def get_parser():
    p = options.get_parser('dipole api')
    p.set_defaults(number_parts=10, minimum_points_per_patch=100, iters=5, diffuse=True, weighted_prop=True)
    return p


if __name__ == '__main__':
    opts = get_parser().parse_args()
    opts.export_dir.mkdir(exist_ok=True, parents=True)
    options.export_options(opts)
    orient_large(opts)
