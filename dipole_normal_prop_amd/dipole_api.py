"""Importable API of the reference's dipole_api.py: orient_large(opts) (dipole_api.py:14-87) with its
own parser (dipole_api.py:101-132; the reference hard-codes Windows paths as defaults, here --pc and
--export_dir are required instead)."""
from . import options
from .pipeline import orient_representatives as _run_large


def orient_large(opts):
    """Patch partition, <=500 representatives per patch, dipole propagation on the representatives,
    global flip by the mean potential, export to opts.export_dir/final_result.xyz."""
    return _run_large(opts)


def simple_estimate(xyz_data, config):
    """The request handler behind the reference's socket servers (socket_server.py:18-27): numpy xyz in
    (float64 on the wire) -> PCA normals -> unit-box transform -> per-point dipole propagation -> global flip
    by the mean potential -> numpy [N,6] out in the input's frame and dtype.  A float64 cloud (what arrives on
    the wire, util.py:71-77) stays float64 throughout: the per-point propagation runs in the fp64 kernel
    (dnp_point_greedy_f64), as the reference's does.  The wire framing around this handler (JSON header + raw
    float64, socket_server_para.py:142-195) is dipole_normal_prop_amd.wire; the TCP server itself is not built."""
    import torch
    from . import field_utils, util
    dev = torch.device("cuda", torch.cuda.current_device())
    input_pc = util.npxyz2tensor(xyz_data).to(dev)
    input_pc = util.estimate_normals(input_pc, max_nn=30)
    input_pc, transform = util.Transform.trans(input_pc)
    field_utils.strongest_field_propagation_points(input_pc, diffuse=config["diffuse"], starting_point=0)
    if field_utils.measure_mean_potential(input_pc) < 0:
        input_pc[:, 3:] *= -1
    return transform.inverse(input_pc).cpu().numpy()


def get_parser():
    p = options.get_parser('dipole api')
    p.set_defaults(number_parts=10, minimum_points_per_patch=100, iters=5, diffuse=True, weighted_prop=True)
    return p


if __name__ == '__main__':
    options.main(orient_large, get_parser())
