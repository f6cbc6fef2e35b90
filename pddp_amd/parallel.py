"""Multi-GPU: one process per GPU, trajectories sharded contiguously, ONE
collective per solve - an all-gather of each rank's best rollout.

The reference has no distributed code; the trajectory batch is a
new axis and its trajectories are independent, so the data path needs no
collective at all.  The only exchange is the reduction the north star names:
every rank contributes the fixed-size record {J_best, global index,
Z_best[(N+1) n], U_best[N m]} to one `all_gather_into_tensor` (RCCL over xGMI
with backend "nccl", gloo on CPU for tests) and takes the global argmin
redundantly.  At double cartpole size the record is ~5 KB: latency-bound.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world_size):
    """Contiguous [lo, hi) slice of `total` trajectories owned by `rank`;
    sizes differ by at most one."""
    base, rem = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_best(J, Z, U, offset=0):
    """Fused buffer [J, index, Z.flatten(), U.flatten()] of this rank's best
    (lowest finite cost) trajectory.  J [B], Z [B, N+1, n], U [B, N, m]."""
    Jc = torch.where(torch.isfinite(J), J, torch.full_like(J, float("inf")))
    idx = torch.argmin(Jc)
    head = torch.stack([Jc[idx].to(torch.float64),
                        (idx + offset).to(torch.float64)])
    return torch.cat([head, Z[idx].reshape(-1).to(torch.float64),
                      U[idx].reshape(-1).to(torch.float64)])


def gather_best_rollout(J, Z, U, offset=0, group=None):
    """All-gathers every rank's best rollout and returns the global best as
    (J_best, global_index, Z_best [N+1, n], U_best [N, m]); identical on every
    rank.  Works without an initialised process group (world size 1)."""
    mine = pack_best(J, Z, U, offset)
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(group)
        out = torch.empty(world * mine.numel(), dtype=mine.dtype,
                          device=mine.device)
        dist.all_gather_into_tensor(out, mine.contiguous(), group=group)
        out = out.view(world, -1)
    else:
        out = mine.unsqueeze(0)
    best = torch.argmin(out[:, 0])
    row = out[best]
    nz = Z[0].numel()
    Zb = row[2:2 + nz].reshape(Z.shape[1:]).to(Z.dtype)
    Ub = row[2 + nz:].reshape(U.shape[1:]).to(U.dtype)
    return row[0].to(J.dtype), int(row[1].item()), Zb, Ub
