"""BASELINE.json configs[3]: a batch of 64 jittered ~30k-tet boxes, 45 eigenpairs each, dealt over
the ranks by the LPT rule, no data-path collective, one gather of fixed-size records at the end.

    python tools/batch_bench.py [--meshes 64] [--n 17]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/batch_bench.py
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesheditor_amd import api, meshes, sharding


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--meshes", type=int, default=64)
    ap.add_argument("--n", type=int, default=17, help="grid cells per edge (17 -> 29 478 tets)")
    ap.add_argument("--threads", type=int, default=3, help="host threads (contexts) per GPU: concurrent solves")
    a = ap.parse_args()
    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    import torch
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    ctxs = [api.Context(local_rank) for _ in range(max(1, a.threads))]
    ctx = ctxs[0]
    batch = []
    for i in range(a.meshes):
        p, t = meshes.jittered_box(a.n, 1000 + i)
        batch.append((p, t, meshes.MATERIALS[meshes.MATERIAL_ORDER[i % 7]], {"num_modes": 30, "num_fem_modes": 45}))

    def solve(i, m, worker=0):
        ctx = ctxs[worker]
        p, t, mat, kw = m
        cfg = api.default_config(num_modes=kw["num_modes"], num_fem_modes=kw["num_fem_modes"])
        ex = p[:: max(1, len(p) // 10)][:10].astype(np.float32)
        return api.mesh2modes(ctx, p, t, api.material(*mat), ex, config=cfg)
    solve(0, batch[0])  # warm-up (library code objects)
    ctx.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    recs = sharding.solve_batch(batch, solve, 45, dist, "cuda" if dist is not None else "cpu", threads=a.threads)
    [c.synchronize() for c in ctxs]
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        pairs = sum(len(r["eigenvalues"]) for r in recs)
        print(json.dumps({"workload": "batch of %d jittered boxes, %d tets each, 45 eigenpairs" % (a.meshes, len(batch[0][1])), "n_gpus": world, "threads_per_gpu": a.threads, "seconds": dt,
                          "eigenpairs_per_s": pairs / dt, "meshes_per_s": a.meshes / dt, "iterations_mean": float(np.mean([r["iterations"] for r in recs]))}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
