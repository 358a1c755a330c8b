"""Image sharding across the GPUs of a node.

Images are independent on this path (every loop of the reference is per image: RPN/rpn.py:63,74,80,105;
ROIHead/roi_head.py:99,178), so the batch is split by image, one process per GPU, and the hot path
itself needs no collective.  The only exchange is the final gather of the per-image proposals
(zero-padded to num_post_nms rows) and their counts: two all_gathers over RCCL ("nccl" backend) on
GPU tensors, or gloo on CPU tensors in the tests.  RoI features (~200 MB per image) stay on the GPU
that produced them.
"""
import torch


def shard_range(rank, world, n_images):
    """Contiguous, balanced slice of the global batch owned by `rank`: [lo, hi)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, rem = divmod(n_images, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_proposals(props, counts, dist=None, group=None):
    """All-gather the zero-padded proposals (b_local, P, 4) and counts (b_local,) of every rank and
    return them in global image order as (B, P, 4), (B,).  Ranks may own different numbers of
    images (shard_range): shards are padded to the largest one for the collective.
    With dist=None (single process) the inputs are returned unchanged."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return props, counts
    world = dist.get_world_size(group)
    n_local = torch.tensor([props.shape[0]], dtype=torch.int64, device=props.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad_p = torch.zeros((m,) + tuple(props.shape[1:]), dtype=props.dtype, device=props.device)
    pad_c = torch.zeros((m,), dtype=counts.dtype, device=counts.device)
    pad_p[:props.shape[0]] = props
    pad_c[:counts.shape[0]] = counts
    all_p = [torch.empty_like(pad_p) for _ in range(world)]
    all_c = [torch.empty_like(pad_c) for _ in range(world)]
    dist.all_gather(all_p, pad_p, group=group)
    dist.all_gather(all_c, pad_c, group=group)
    return (torch.cat([p[:s] for p, s in zip(all_p, sizes)], dim=0),
            torch.cat([c[:s] for c, s in zip(all_c, sizes)], dim=0))
