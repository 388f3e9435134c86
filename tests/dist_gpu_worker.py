"""Child process of tests/test_e2e_gpu.py: one rank of rupphash_amd.dist.hash_and_group_device with the real Engine.
Started as a fresh interpreter (never forked from a process that holds the GPU).  Ranks may share one GPU: the backend is
then gloo (collectives staged through host memory; RCCL refuses two ranks on one device); on a multi-GPU node pass nccl.

usage: dist_gpu_worker.py RANK WORLD PORT N_TOTAL FIRST_IMAGE SIMILARITY VARIANTS(0/1) BACKEND OUT_JSON
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    rank, world, port, n_total, first, sim, variants = (int(x) for x in sys.argv[1:8])
    backend, out = sys.argv[8], sys.argv[9]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from rupphash_amd import Engine
    from rupphash_amd import dist as D

    ndev = max(torch.cuda.device_count(), 1)
    local = rank % ndev
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = Engine(local)
        dev = torch.device("cuda", local)
        lo, hi = D.shard_range(n_total, rank, world)
        imgs = torch.empty((hi - lo, 512 * 512 * 3), dtype=torch.uint8, device=dev)
        eng.synth_images_dev(imgs.data_ptr(), first + lo, hi - lo, 512, 512, stream=torch.cuda.current_stream().cuda_stream)
        groups, info = D.hash_and_group_device(eng, imgs, n_total, sim, dist, variants=bool(variants))
        if rank == 0:
            with open(out, "w") as f:
                json.dump({"groups": groups, "info": info, "backend": str(dist.get_backend()), "world": dist.get_world_size()}, f)
        dist.barrier()
        eng.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
