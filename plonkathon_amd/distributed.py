"""Proof-level data parallelism over the GPUs of one node.

Proofs are independent, so the path shards by proof index with NO data-path collective: rank r of W
proves indices r, r+W, r+2W, ... on its own GPU (circuit polynomials, SRS window table and twiddles
are replicated per GPU).  The only exchange is one all_gather of the finished proofs — 768 bytes each
(9 affine G1 + 6 Fr) — over RCCL/xGMI (`backend="nccl"` on ROCm) or gloo in the CPU tests.
"""
import os

PROOF_BYTES = 768


def shard_indices(total: int, rank: int, world: int):
    """Indices of the proofs rank `rank` of `world` is responsible for (round-robin)."""
    return list(range(rank, total, world))


def init_from_env(backend=None):
    """torch.distributed initialisation from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return None
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist


def gather_proofs(local_blob: bytes, total: int, dist=None):
    """all_gather the per-rank proof blobs and return all `total` proofs in global index order.

    `local_blob` holds this rank's proofs (768 B each) in the order of `shard_indices`.  Ranks may
    own different counts (total % world != 0), so blobs are padded to the largest shard."""
    if dist is None:
        assert len(local_blob) == PROOF_BYTES * total
        return [local_blob[PROOF_BYTES * i : PROOF_BYTES * (i + 1)] for i in range(total)]
    import torch

    rank, world = dist.get_rank(), dist.get_world_size()
    per = (total + world - 1) // world
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    buf = bytearray(local_blob) + bytearray(PROOF_BYTES * per - len(local_blob))
    mine = torch.frombuffer(buf, dtype=torch.uint8).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = [None] * total
    for r in range(world):
        raw = bytes(parts[r].cpu().numpy().tobytes())
        for j, idx in enumerate(shard_indices(total, r, world)):
            out[idx] = raw[PROOF_BYTES * j : PROOF_BYTES * (j + 1)]
    return out


def max_over_ranks(value: float, dist=None) -> float:
    if dist is None:
        return value
    import torch

    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
