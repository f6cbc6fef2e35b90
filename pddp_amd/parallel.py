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
    (lowest finite cost) trajectory, in the run's own dtype (the index is
    exact in float32 up to 2^24 trajectories).  J [B], Z [B, N+1, n],
    U [B, N, m].  Device ops only - no host synchronisation."""
    Jc = torch.where(torch.isfinite(J), J, torch.full_like(J, float("inf")))
    idx = torch.argmin(Jc)
    head = torch.stack([Jc[idx], (idx + offset).to(J.dtype)])
    return torch.cat([head, Z[idx].reshape(-1).to(J.dtype),
                      U[idx].reshape(-1).to(J.dtype)])


def gather_best_rollout(J, Z, U, offset=0, group=None, sync=True):
    """All-gathers every rank's best rollout and returns the global best as
    (J_best, global_index, Z_best [N+1, n], U_best [N, m]); identical on every
    rank.  Works without an initialised process group (world size 1).
    `sync=False`: the index comes back as a 0-dim device tensor and nothing
    waits for the host - the form for one exchange PER ITERATION (SURVEY
    8(e)) inside a loop of asynchronous launches."""
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
    index = row[1].round().to(torch.int64)
    return row[0].to(J.dtype), (int(index.item()) if sync else index), Zb, Ub
