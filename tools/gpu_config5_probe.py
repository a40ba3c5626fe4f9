"""field_grad / reference_field at S = T = 100 000 on the patch-sorted sphere with 6-column, 3-column and jittered targets
(BASELINE config 5): where the time of a reference_field call goes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.workloads import sphere_cloud, fibonacci_patches
from dipole_normal_prop_amd import field_utils as fu, util
dev = torch.device("cuda:0")
pc = sphere_cloud(); patches = fibonacci_patches(pc)
off, idx, _ = util.patch_csr(patches, dev)
pts = pc.to(dev)[idx].contiguous()
g = torch.Generator().manual_seed(3)
tgt = (pts[:, :3].cpu() + 1e-3 * torch.randn(pts.shape[0], 3, generator=g)).to(dev)
def t(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("field_grad(pts, pts)        ", round(t(lambda: fu.field_grad(pts, pts)), 3))
print("field_grad(pts, pts[:, :3]) ", round(t(lambda: fu.field_grad(pts, pts[:, :3].contiguous())), 3))
print("field_grad(pts, tgt)        ", round(t(lambda: fu.field_grad(pts, tgt)), 3))
print("reference_field(pts, tgt)   ", round(t(lambda: fu.reference_field(pts, tgt)), 3))
t6 = torch.cat([tgt, pts[:, 3:]], 1)
print("reference_field(pts, tgt6)  ", round(t(lambda: fu.reference_field(pts, t6.clone())), 3))
