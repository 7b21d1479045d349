"""Multi-GPU split of the sample loop: one process per GPU, each renders a contiguous range of the
samples of every pixel; the float4 accumulation buffers are summed onto one rank by the library's own RCCL
reduce (mirt_comm_reduce, on the render stream). torch.distributed is the control plane only: it carries the
128-byte communicator id from rank 0 to the others and provides the barriers; its backend can be gloo.
The Philox key holds the GLOBAL sample index, so the ranks' streams are disjoint and the reduced image
equals a single-GPU render of all samples up to fp32 reassociation."""


def sample_range(total_spp, world_size, rank):
    """Strong split: contiguous, balanced (sizes differ by at most one) partition of [0, total_spp)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(int(total_spp), int(world_size))
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


def weak_range(spp_per_rank, rank):
    """Weak split: every rank renders spp_per_rank samples, rank r owning [r*spp, (r+1)*spp)."""
    return rank * int(spp_per_rank), int(spp_per_rank)


def reduce_accum(accum, dst=0, group=None):
    """Sum the per-rank accumulation buffers (torch tensors of width*height*4 floats: sum r, g, b and the
    sample count) onto rank `dst`. The only exchange step of the path."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return accum


def join_comm(ctx, rank, world_size, group=None):
    """Make `ctx` rank `rank` of an RCCL communicator of `world_size` contexts, one per process: rank 0 creates
    the id, the process group broadcasts its bytes, every rank joins (collective: all ranks must call it)."""
    import torch.distributed as dist
    import ray_tracing_amd as rt
    ids = [rt.comm_unique_id() if rank == 0 else None]
    if world_size > 1:
        dist.broadcast_object_list(ids, src=0, group=group)
    ctx.comm_init_rank(world_size, rank, ids[0])
