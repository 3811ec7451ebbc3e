"""Multi-GPU decomposition of one aggregate verification.

Signer i's Miller value is independent of every other signer's, so the batch is cut into
contiguous ranges, one per rank (= per GPU); the only exchange is ONE all-gather of the per-rank
partial products (384 B alt-bn128 / 576 B BLS12-381 each), after which every rank multiplies the
partials in rank order, applies the single final exponentiation and holds the same verdict.
RCCL's reduction operators do not know the Fp12 product, hence gather-then-multiply
(`ncclProd` would multiply limbs as integers)."""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous [lo, hi) of rank `rank` among `world` ranks."""
    return n * rank // world, n * (rank + 1) // world


def all_gather_bytes(part, world):
    """part: 1-D uint8 tensor (on the device of the default process group's backend).
    Returns a [world, len] uint8 tensor holding every rank's bytes in rank order."""
    if world == 1:
        return part.reshape(1, -1).clone()
    out = torch.empty(world * part.numel(), dtype=torch.uint8, device=part.device)
    dist.all_gather_into_tensor(out, part.contiguous().reshape(-1))
    return out.reshape(world, part.numel())
