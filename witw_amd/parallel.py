"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend name 'nccl').

The encoder has no BatchNorm, so samples are independent and the minibatch shards across ranks
(SURVEY.md §8e). The loss couples the GLOBAL batch (nn.DataParallel semantics of the reference,
model/cvig_baseline.py:339-343; normaliser 2B(B-1) with B = global batch, model/cvig_fov.py:380):
embeddings are all-gathered (2 MiB per rank per side at B=128) and every rank evaluates the
match + loss over the global batch.
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def all_gather_embeddings(surface_embed, overhead_embed):
    """[b,16,4,We],[b,16,4,64] per rank -> global [B,...] tensors in rank order (one fused
    collective for both sides when their shapes agree, two otherwise)."""
    n = world()
    if n == 1:
        return surface_embed, overhead_embed
    su = surface_embed.contiguous()
    ov = overhead_embed.contiguous()
    su_all = torch.empty((n * su.shape[0],) + tuple(su.shape[1:]), dtype=su.dtype, device=su.device)
    ov_all = torch.empty((n * ov.shape[0],) + tuple(ov.shape[1:]), dtype=ov.dtype, device=ov.device)
    h1 = dist.all_gather_into_tensor(su_all, su, async_op=True)
    h2 = dist.all_gather_into_tensor(ov_all, ov, async_op=True)
    h1.wait()
    h2.wait()
    return su_all, ov_all
