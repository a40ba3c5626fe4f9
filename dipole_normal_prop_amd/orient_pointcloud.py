"""Counterpart of the reference's orient_pointcloud.py (run(opts), orient_pointcloud.py:11-76):
patch partition -> flatness filter -> orient_center per patch -> greedy dipole propagation over
the patches -> global flip by the mean potential -> export.  The PointCNN voting iterations of the
reference (lines 42-54) are outside this package; with no models they are no-ops there as well,
except that the propagation also runs every `propagation_iters` iterations - reproduced here.

    python -m dipole_normal_prop_amd.orient_pointcloud --pc cloud.xyz --export_dir out --number_parts 30 \
        --minimum_points_per_patch 100 --diffuse
"""
import torch

from . import field_utils, options, util
from .options import get_parser

torch.manual_seed(1)


def run(opts):
    options.reject_models(opts)
    MyTimer = util.timer_factory()
    device = torch.device(torch.cuda.current_device() if torch.cuda.is_available() else 'cpu')
    pc = util.load_xyz(opts.pc).to(device)
    if opts.estimate_normals:
        with MyTimer('estimating normals'):
            pc = util.estimate_normals(pc, max_nn=opts.n)
    pc, transform = util.Transform.trans(pc)
    input_pc = pc.clone()

    with MyTimer('divide patches'):
        patch_indices = util.divide_pc(input_pc[:, :3], opts.number_parts, min_patch=opts.minimum_points_per_patch)
        all_patches_indices = [x.clone() for x in patch_indices]
    with MyTimer('filter patches'):
        patch_indices = util.fix_n_filter(input_pc, patch_indices, opts.curvature_threshold)
    print(f'number of patches {len(patch_indices)}')
    with MyTimer('orient center'):
        util.orient_center_patches(input_pc, [p for _, p in patch_indices])
    pc_probs = torch.ones_like(input_pc[:, 0])

    def propagate():
        with torch.no_grad(), MyTimer('propagation'):
            field_utils.strongest_field_propagation(input_pc, patch_indices, all_patches_indices,
                                                    diffuse=opts.diffuse,
                                                    weights=pc_probs if opts.weighted_prop else None)

    for it in range(opts.iters):
        if it % opts.propagation_iters == 0 and (it != 0 or opts.propagation_iters == 1):
            propagate()
    propagate()

    with MyTimer('fix global orientation'):
        if field_utils.measure_mean_potential(input_pc) < 0:
            input_pc[:, 3:] *= -1
    MyTimer.print_total_time()
    with MyTimer('exporting result', count=False):
        util.export_pc(transform.inverse(input_pc).transpose(0, 1), opts.export_dir / 'final_result.xyz')
    return input_pc


if __name__ == '__main__':
    opts = get_parser().parse_args()
    opts.export_dir.mkdir(exist_ok=True, parents=True)
    options.export_options(opts)
    run(opts)
