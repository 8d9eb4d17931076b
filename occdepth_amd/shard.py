"""Batch-data-parallel sharding of forward frames: frames are independent in eval mode,
so rank r of W processes frames {r, r+W, ...}; no data-path collective exists.  The only communication is
the benchmark's barrier and the MAX-reduction of the per-rank wall time."""
import torch


def frames_for_rank(n_frames, rank, world):
    return list(range(rank, n_frames, world))


def max_over_ranks(seconds, dist=None, device="cpu"):
    """Largest per-rank elapsed time (what bounds whole-job throughput)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def fence(dist=None):
    """barrier + device sync on both sides of a timed region."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def gather_frames(local_results, n_frames, dist=None):
    """Reassemble {frame_index: tensor} dicts from all ranks on every rank (tests / offline inference)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_results[i] for i in range(n_frames)]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local_results)
    merged = {}
    for p in parts:
        merged.update(p)
    return [merged[i] for i in range(n_frames)]
