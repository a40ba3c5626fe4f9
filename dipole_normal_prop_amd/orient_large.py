"""Counterpart of the reference's orient_large.py (run(opts), orient_large.py:10-82): as
orient_pointcloud, but every patch is represented by at most 500 random representatives
(torch.randperm under torch.manual_seed(1)); the propagation runs on the representatives and the
rest of each patch follows (field_utils.strongest_field_propagation_reps, always diffuse)."""
from pathlib import Path

import torch

from . import field_utils, options, util

torch.manual_seed(1)
MAX_PATCH_SIZE = 500


def run(opts):
    options.reject_models(opts)
    export_path: Path = opts.export_dir
    export_path.mkdir(exist_ok=True)
    device = torch.device(torch.cuda.current_device() if torch.cuda.is_available() else 'cpu')
    MyTimer = util.timer_factory()
    with MyTimer('load pc', count=False):
        input_pc = util.load_xyz(opts.pc, append_normals=False).to(device)
    input_pc, transform = util.Transform.trans(input_pc)
    if opts.estimate_normals:
        with MyTimer('estimating normals'):
            input_pc = util.estimate_normals(input_pc, max_nn=opts.n)
    if input_pc.shape[1] != 6:
        raise SystemExit('the cloud has no normals: pass --estimate_normals')

    with MyTimer('divide patches'):
        patch_indices = util.divide_pc(input_pc[:, :3], opts.number_parts, min_patch=opts.minimum_points_per_patch)
        all_patches_indices = [x.clone() for x in patch_indices]
    with MyTimer('filter patches'):
        patch_indices = util.fix_n_filter(input_pc, patch_indices, opts.curvature_threshold)
    print(f'number of patches {len(patch_indices)}/{len(all_patches_indices)}')
    with MyTimer('orient center'):
        util.orient_center_patches(input_pc, [p for _, p in patch_indices])
    with MyTimer('find reps'):
        represent = []
        for p in all_patches_indices:
            perm = torch.randperm(p.shape[0]).to(p.device)
            represent.append((p[perm[:MAX_PATCH_SIZE]], p[perm[MAX_PATCH_SIZE:]]))
    with MyTimer('propagating field'):
        field_utils.strongest_field_propagation_reps(input_pc, represent, diffuse=True)
    with MyTimer('fix global orientation'):
        if field_utils.measure_mean_potential(input_pc) < 0:
            input_pc[:, 3:] *= -1
    with MyTimer('exporting result', count=False):
        util.export_pc(transform.inverse(input_pc).transpose(0, 1), export_path / 'final_result.xyz')
    MyTimer.print_total_time()
    return input_pc


if __name__ == '__main__':
    opts = options.get_parser().parse_args()
    opts.export_dir.mkdir(exist_ok=True, parents=True)
    options.export_options(opts)
    run(opts)
