"""Wall-time breakdown of N gloo ranks sharing one GPU: python tools/gpu_dist_probe.py <world>"""
import os, socket, sys, time
T0 = time.time()
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def worker(rank, world, port, q, t_spawn):
    t = [("spawned+imports", time.time() - t_spawn)]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    a = time.time(); dist.init_process_group("gloo", rank=rank, world_size=world); t.append(("init_pg", time.time() - a))
    a = time.time()
    from dipole_normal_prop_amd import field_utils as fu, parallel
    from tools.workloads import headline_workload
    t.append(("import pkg", time.time() - a))
    dev = torch.device("cuda:0")
    a = time.time(); pc, patches, _ = headline_workload(); t.append(("workload", time.time() - a))
    a = time.time(); allp = [p.to(dev) for p in patches]; pts = pc.clone().to(dev); torch.cuda.synchronize(); t.append(("to dev", time.time() - a))
    a = time.time(); parallel.sharded_patch_propagation(pts, list(enumerate(allp)), allp, diffuse=True); torch.cuda.synchronize(); t.append(("propagation", time.time() - a))
    a = time.time(); parallel.sharded_patch_propagation(pts, list(enumerate(allp)), allp, diffuse=True); torch.cuda.synchronize(); t.append(("propagation again", time.time() - a))
    q.put((rank, t))
    dist.destroy_process_group()

if __name__ == "__main__":
    world = int(sys.argv[1])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    t_spawn = time.time()
    procs = [ctx.Process(target=worker, args=(r, world, port, q, t_spawn)) for r in range(world)]
    [p.start() for p in procs]
    for _ in range(world):
        r, t = q.get(timeout=600)
        print(r, " | ".join(f"{k} {v:.1f}s" for k, v in t), flush=True)
    [p.join() for p in procs]
    print("total", time.time() - T0)
