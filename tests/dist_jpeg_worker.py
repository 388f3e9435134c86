"""Child process of tests/test_e2e_gpu.py: one rank of rupphash_amd.dist.scan_jpeg_files_and_group with the real Engine (JPEG files ->
decode + hash on the GPU -> all-gather -> sweep -> groups).  Ranks may share one GPU (gloo), as in dist_gpu_worker.py.

usage: dist_jpeg_worker.py RANK WORLD PORT N_TOTAL SIMILARITY BACKEND OUT_JSON
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def make_files(eng, first, count):
    """file k of the sequence: synthetic image k // 2 as a JPEG -- even k baseline quality 90, odd k quality 60 (every fourth of those
    progressive): each pair must come out as a group; file 7 is cut off in its header (unreadable)"""
    import io

    from PIL import Image

    files = []
    for k in range(first, first + count):
        img = eng.synth_images(1000 + k // 2, 1)[0]
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", quality=90 if k % 2 == 0 else 60, subsampling=2, progressive=(k % 8 == 3))
        data = buf.getvalue()
        files.append(data[:100] if k == 7 else data)
    return files


def main():
    rank, world, port, n_total, sim = (int(x) for x in sys.argv[1:6])
    backend, out = sys.argv[6], sys.argv[7]
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
        lo, hi = D.shard_range(n_total, rank, world)
        groups, info = D.scan_jpeg_files_and_group(eng, make_files(eng, lo, hi - lo), n_total, sim, dist, threads=4)
        if rank == 0:
            with open(out, "w") as f:
                json.dump({"groups": groups, "info": info, "backend": str(dist.get_backend()), "world": dist.get_world_size()}, f)
        dist.barrier()
        eng.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
