"""Multi-GPU: one process per GPU, trajectories sharded contiguously, ONE
collective per iteration - an all-gather of each rank's best rollout.

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
    idx = torch.argmin(Jc).reshape(1)
    pick = lambda t: torch.index_select(t, 0, idx).reshape(-1).to(J.dtype)
    return torch.cat([pick(Jc), (idx + offset).to(J.dtype), pick(Z), pick(U)])


def _best_row(out):
    """Row of lowest cost of the gathered records [world, rec] as a NEW tensor
    (`out[argmin]` with a device index is a host round trip - `.item()` - and a
    view into a receive buffer that a later post overwrites; index_select is
    neither)."""
    best = torch.argmin(out[:, 0]).reshape(1)
    return torch.index_select(out, 0, best)[0]


def _index_fits(dtype, n_total):
    """The record carries the global index in the run's dtype: exact up to
    2^24 trajectories in float32 (2^53 in float64)."""
    limit = {torch.float32: 1 << 24, torch.float64: 1 << 53}.get(dtype)
    if limit is None or n_total > limit:
        raise ValueError(
            "best-rollout record: global trajectory index %d is not exact in "
            "%s" % (n_total, dtype))


class BestRolloutExchange(object):
    """The exchange of the best rollout, once PER ITERATION inside a loop of
    asynchronous launches - built so that a round of 83 us pays
    a few microseconds for it:

    * `post()` packs this rank's record with ONE launch (`pddp_pack_best_*`;
      the torch form of the same selection is eight small kernels, half a
      round) on the caller's stream, and issues the all-gather on a SIDE
      stream behind an event: the collective (latency-bound: ~2 KB per rank)
      overlaps with the next round's kernels, nothing waits for the host.
    * the send / receive buffers rotate (`depth` of them); a buffer is packed
      again only after the gather that read it has finished (stream waits on
      events - again no host wait).
    * `result()` makes the caller's stream wait for the latest gather and
      returns (J_best, global index [0-dim tensor], Z_best, U_best),
      identical on every rank - fresh tensors (the receive buffers rotate
      under later posts), picked without a host round trip."""

    def __init__(self, J, Z, U, group=None, depth=4):
        self.group = group
        self.world = (dist.get_world_size(group)
                      if dist.is_available() and dist.is_initialized() else 1)
        self.zshape, self.ushape = tuple(Z.shape[1:]), tuple(U.shape[1:])
        self.nz = int(Z[0].numel())
        self.nu = int(U[0].numel())
        self.dtype, self.device = J.dtype, J.device
        rec = 2 + self.nz + self.nu
        opts = dict(dtype=J.dtype, device=J.device)
        self.recv = [torch.zeros(self.world, rec, **opts) for _ in range(depth)]
        # (a world of one: the record is packed straight into the result)
        self.send = [torch.zeros(rec, **opts) if self.world > 1 else r[0]
                     for r in self.recv]
        # events, made once: slot k packed / its gather finished
        self.packed = [torch.cuda.Event() for _ in range(depth)]
        self.done = [torch.cuda.Event() for _ in range(depth)]
        self.used = [False] * depth
        self.side = torch.cuda.Stream(device=J.device)
        self.k = -1

    def post(self, J, Z, U, offset=0):
        from . import _native
        _index_fits(self.dtype, int(offset) + int(J.numel()))
        k = self.k = (self.k + 1) % len(self.send)
        main = torch.cuda.current_stream(self.device)
        if self.used[k] and self.world > 1:
            main.wait_event(self.done[k])  # slot k: its last gather has read it
        self.used[k] = True
        p = _native.ptr
        _native.call("pddp_pack_best", self.dtype, int(J.numel()), self.nz,
                     self.nu, p(J), p(Z), p(U), int(offset), p(self.send[k]),
                     _native.stream_handle(self.device))
        if self.world > 1:
            self.packed[k].record(main)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.packed[k])
                dist.all_gather_into_tensor(self.recv[k].view(-1),
                                            self.send[k], group=self.group)
                self.done[k].record(self.side)
        return self

    def result(self):
        k = self.k
        if self.world > 1:
            torch.cuda.current_stream(self.device).wait_event(self.done[k])
        row = _best_row(self.recv[k])
        Zb = row[2:2 + self.nz].reshape(self.zshape)
        Ub = row[2 + self.nz:].reshape(self.ushape)
        return row[0], row[1].round().to(torch.int64), Zb, Ub


_exchanges = {}


def post_best_rollout(J, Z, U, offset=0, group=None):
    """`BestRolloutExchange.post` on an exchange cached per (device, dtype,
    shapes, group): what a fit loop calls after every round; `.result()` of the
    returned object when somebody wants the global best."""
    key = (J.device, J.dtype, tuple(Z.shape), tuple(U.shape), id(group))
    ex = _exchanges.get(key)
    if ex is None:
        if len(_exchanges) >= 4:
            _exchanges.clear()
        ex = _exchanges[key] = BestRolloutExchange(J, Z, U, group)
    return ex.post(J, Z, U, offset)


def gather_best_rollout(J, Z, U, offset=0, group=None, sync=True):
    """All-gathers every rank's best rollout and returns the global best as
    (J_best, global_index, Z_best [N+1, n], U_best [N, m]); identical on every
    rank.  Works without an initialised process group (world size 1).
    `sync=False`: the index comes back as a 0-dim device tensor and nothing
    waits for the host - the form for one exchange PER ITERATION (SURVEY
    8(e)) inside a loop of asynchronous launches."""
    if J.is_cuda and Z.is_contiguous() and U.is_contiguous() and \
            J.is_contiguous() and Z.dtype == J.dtype and U.dtype == J.dtype:
        Jb, index, Zb, Ub = post_best_rollout(J, Z, U, offset, group).result()
        return Jb, (int(index.item()) if sync else index), Zb, Ub
    _index_fits(J.dtype, int(offset) + int(J.numel()))
    mine = pack_best(J, Z, U, offset)
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(group)
        out = torch.empty(world * mine.numel(), dtype=mine.dtype,
                          device=mine.device)
        dist.all_gather_into_tensor(out, mine.contiguous(), group=group)
        out = out.view(world, -1)
    else:
        out = mine.unsqueeze(0)
    row = _best_row(out)
    nz = Z[0].numel()
    Zb = row[2:2 + nz].reshape(Z.shape[1:]).to(Z.dtype)
    Ub = row[2 + nz:].reshape(U.shape[1:]).to(U.dtype)
    index = row[1].round().to(torch.int64)
    return row[0].to(J.dtype), (int(index.item()) if sync else index), Zb, Ub
