"""Orchestration of the orientation pipelines around the dipole kernels.

One module holds the stages; the reference's entry scripts (orient_pointcloud.py:11-76, orient_large.py:10-82,
orient_simple.py:8-34, reference_orientation.py:8-28) survive only as thin modules that pick a flow, so that
`python -m dipole_normal_prop_amd.orient_large ...` and `orient_large.run(opts)` keep working.  Behavioural
contract per flow (what the reference's scripts do, minus the PointCNN voting that is outside this package):

    points           Transform -> per-point greedy propagation from point 0 -> global flip -> export
    patches          Transform -> voxel partition + merge -> flatness filter -> orient_center per kept patch ->
                     greedy patch propagation (repeated on the reference's iteration schedule) -> global flip -> export
    representatives  as `patches`, but every patch is represented by <= 500 random points (torch.randperm
                     under torch.manual_seed(1)) and the rest of the patch follows its representatives
    transfer         one field evaluation from an oriented reference cloud onto the input cloud
"""
from pathlib import Path

import torch

from . import field_utils, options, util

REPRESENTATIVES_PER_PATCH = 500


class _Stages:
    """Wall-clock log of the stages of one run (the reference prints one line per stage as well)."""

    def __init__(self):
        self._timer = util.timer_factory()

    def __call__(self, label, counted=True):
        return self._timer(label, count=counted)

    def total(self):
        self._timer.print_total_time()


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def _read_cloud(path, stages, keep_three_columns=False):
    with stages("load pc", counted=False):
        return util.load_xyz(path, append_normals=not keep_three_columns).to(_device())


def _maybe_estimate(cloud, opts, stages, max_nn):
    if getattr(opts, "estimate_normals", False):
        with stages("estimating normals"):
            cloud = util.estimate_normals(cloud, max_nn=max_nn)
    return cloud


def _partition(cloud, opts, stages):
    """(kept, every) patch lists: `every` takes part in the greedy ordering, `kept` = [(index, rows)] passed the
    flatness filter and receives orient_center / the diffuse sign pass."""
    with stages("divide patches"):
        every = util.divide_pc(cloud[:, :3], opts.number_parts, min_patch=opts.minimum_points_per_patch)
    with stages("filter patches"):
        kept = util.fix_n_filter(cloud, every, opts.curvature_threshold)
    print(f"number of patches {len(kept)}/{len(every)}")
    with stages("orient center"):
        util.orient_center_patches(cloud, [rows for _, rows in kept])
    return kept, every


def _global_flip(cloud, stages):
    with stages("fix global orientation"):
        if field_utils.measure_mean_potential(cloud) < 0:
            cloud[:, 3:] *= -1


def _write(cloud, transform, export_dir, stages):
    # the reference prints its "warning: %d inf/nan in field_grad" lines inside the offending call; here they are
    # deferred (_staging._WarnState) - print whatever is pending before the result is written, not at exit
    field_utils.flush_warnings()
    with stages("exporting result", counted=False):
        util.export_pc(transform.inverse(cloud).transpose(0, 1), Path(export_dir) / "final_result.xyz")


def orient_points(opts):
    stages = _Stages()
    Path(opts.export_dir).mkdir(exist_ok=True)
    cloud = _maybe_estimate(_read_cloud(opts.pc, stages), opts, stages, max_nn=30)
    cloud, transform = util.Transform.trans(cloud)
    with stages("propagating field"):
        field_utils.strongest_field_propagation_points(cloud, diffuse=opts.diffuse, starting_point=0)
    _global_flip(cloud, stages)
    _write(cloud, transform, opts.export_dir, stages)
    stages.total()
    return cloud


def orient_patches(opts):
    options.reject_models(opts)
    stages = _Stages()
    cloud = _maybe_estimate(_read_cloud(opts.pc, stages), opts, stages, max_nn=opts.n)
    cloud, transform = util.Transform.trans(cloud)
    kept, every = _partition(cloud, opts, stages)
    confidence = torch.ones_like(cloud[:, 0])        # the network's per-point vote confidence; 1 without models

    def propagate():
        with torch.no_grad(), stages("propagation"):
            field_utils.strongest_field_propagation(cloud, kept, every, diffuse=opts.diffuse,
                                                    weights=confidence if opts.weighted_prop else None)

    # the reference interleaves a propagation into its voting iterations every `propagation_iters` rounds
    # (never at round 0 unless propagation_iters == 1) and runs one more at the end
    for rnd in range(opts.iters):
        if rnd % opts.propagation_iters == 0 and (rnd != 0 or opts.propagation_iters == 1):
            propagate()
    propagate()
    _global_flip(cloud, stages)
    stages.total()
    _write(cloud, transform, opts.export_dir, stages)
    return cloud


def _representatives(every):
    """<= REPRESENTATIVES_PER_PATCH random points per patch and the rest (orient_large.py:48-52): one
    torch.randperm per patch on the CPU generator, in patch order - the reference's random stream under
    torch.manual_seed(1) - assembled into two index vectors on the host and gathered on the device ONCE."""
    if not isinstance(every, util.PatchList):
        every = util.PatchList(torch.cat(list(every)), [int(p.shape[0]) for p in every])
    rep_pos, rest_pos, rep_sizes, rest_sizes, start = [], [], [], [], 0
    for n in every.sizes:
        shuffle = torch.randperm(n)
        rep_pos.append(shuffle[:REPRESENTATIVES_PER_PATCH] + start)
        rest_pos.append(shuffle[REPRESENTATIVES_PER_PATCH:] + start)
        rep_sizes.append(min(n, REPRESENTATIVES_PER_PATCH))
        rest_sizes.append(max(n - REPRESENTATIVES_PER_PATCH, 0))
        start += n
    flat = every.flat
    rep_flat = flat[torch.cat(rep_pos).to(flat.device)] if rep_pos else flat[:0]
    rest_flat = flat[torch.cat(rest_pos).to(flat.device)] if rest_pos else flat[:0]
    return util.RepLists(util.PatchList(rep_flat, rep_sizes, disjoint=every.disjoint),
                         util.PatchList(rest_flat, rest_sizes, disjoint=every.disjoint))


def orient_representatives(opts):
    options.reject_models(opts)
    stages = _Stages()
    Path(opts.export_dir).mkdir(exist_ok=True)
    cloud, transform = util.Transform.trans(_read_cloud(opts.pc, stages, keep_three_columns=True))
    cloud = _maybe_estimate(cloud, opts, stages, max_nn=opts.n)
    if cloud.shape[1] != 6:
        raise SystemExit("the cloud has no normals: pass --estimate_normals")
    kept, every = _partition(cloud, opts, stages)
    with stages("find reps"):
        reps = _representatives(every)
    with stages("propagating field"):
        field_utils.strongest_field_propagation_reps(cloud, reps, diffuse=True)
    _global_flip(cloud, stages)
    _write(cloud, transform, opts.export_dir, stages)
    stages.total()
    return cloud


def transfer_reference(opts):
    stages = _Stages()
    with stages("load input pc", counted=False):
        cloud = util.load_xyz(opts.input, append_normals=False).to(_device())
    with stages("load reference pc", counted=False):
        oriented = util.load_xyz(opts.reference).to(_device())
    if cloud.shape[-1] == 3:
        cloud = _maybe_estimate(cloud, opts, stages, max_nn=opts.n)
    with stages("calculating field"):
        cloud = field_utils.reference_field(oriented, cloud)
    field_utils.flush_warnings()
    with stages("export referenced normals", counted=False):
        util.export_pc(cloud.transpose(1, 0), opts.output)
    stages.total()
    return cloud
