"""world_size-2 gloo test (CPU) of the N > 1 path of bench.py: replicas, max-over-ranks time, summed work.
The hot path has no data-path collective, so this is all the distributed logic there is."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    elapsed_local = 0.5 + 0.25 * rank                # rank 1 is the slow one
    flops_local = bench.flops(bench.WORKLOAD)        # every rank runs the same per-GPU workload
    dist.barrier()
    t, f = bench.aggregate_over_ranks(elapsed_local, flops_local, dist, "cpu")
    out[rank] = (t, f)
    dist.barrier()
    dist.destroy_process_group()


def test_replica_aggregation_world2_gloo():
    import bench
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    one = bench.flops(bench.WORKLOAD)
    for r in range(world):
        t, f = res[r]
        assert abs(t - 0.75) < 1e-12            # MAX over ranks
        assert abs(f - 2 * one) < 1e-3 * one    # SUM over ranks (weak scaling: per-GPU work fixed)


def test_single_process_passthrough_and_flop_convention():
    import bench
    assert bench.aggregate_over_ranks(1.5, 10.0) == (1.5, 10.0)
    w = bench.WORKLOAD
    # C3 fwd+bwd causal = 240.58 GFLOP
    assert abs(bench.flops(w) / 1e9 - 240.58) < 0.01
    assert abs(bench.causal_fraction(4096, 4096) - 4097 / 8192) < 1e-12
