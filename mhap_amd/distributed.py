"""One-process-per-GPU plumbing for the sharded self-overlap.

Reads are dealt round-robin: global read r (0-based FASTA order) lives on rank r % world at local slot r // world.
Each rank sketches its shard into local tables; the tables are exchanged with ONE all-gather per table
(RCCL over xGMI when the backend is "nccl"; gloo on CPU in the tests).  bench.py keeps the gathered tables rank after
rank (gather_rank_major: no re-layout copy; every rank searches the entry range of its own reads); gather_global_order
re-lays them out in global read order so that entry 2r / 2r+1 is the forward / reverse-complement strand of read r —
ids are then monotonic in entry order, which the brute-force candidate kernel's triangular tile skipping relies on.  Every rank then searches the queries
whose read ordinal is congruent to its rank (mhap_find_matches_self_shard); no collective is needed after that,
records are concatenated by the host (their order is unspecified in the reference too).
"""
import numpy as np
import torch


# ---- the N > 1 exchange step: entry-sharded index, forward query sketches exchanged ------------------------------------------
# Every rank builds the inverted index of its OWN reads only (1/N of the inserts) and keeps its MinHash / ordered tables
# resident; what is exchanged are the forward-strand QUERY sketches (half of a rank's table bytes).  Default: one all-gather
# per table (gather_forward: MinHash rows and meta first, the ordered rows asynchronously while the candidates are computed)
# and ONE search call per rank.  When HBM is short (or MHAP_BENCH_RING=1) the bundles travel round a ring instead (rank r
# sends to r+1, receives from r-1: 2 bundles in memory), and each rank searches the visiting bundle against its shard.  A pair's hit count only involves the
# postings of its stored entry, which all live on that entry's rank, so counts are complete locally; toSelf's id rule
# (J/impl/MinHashSearch.java:215-219) reports every unordered pair exactly once.  Per rank: 1/N of the index build, N probe
# passes over small tables, 2 query bundles + its own shard in memory (C5: ~70 GB of the 288), (N-1)/N of the forward
# tables received point-to-point over xGMI while the previous bundle is being searched.

def gather_forward(rows, world, dist, async_op=False):
    """All-gather the ranks' forward query rows [n_pad, ...] into [world*n_pad, ...], rank after rank (row o*n_pad + j = forward
    strand of read j*world + o).  Returns (tensor, work) — work is None unless async_op on the RCCL path."""
    if world == 1:
        return rows, None
    if rows.is_cuda and dist.get_backend() == "nccl":      # RCCL over xGMI
        out = torch.empty((world * rows.shape[0],) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        work = dist.all_gather_into_tensor(out.view(world, -1), rows.contiguous().view(1, -1), async_op=async_op)
        return out, (work if async_op else None)
    src = rows.contiguous().cpu()                           # gloo (CPU tests; functional multi-rank runs on one GPU)
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src)
    return torch.cat(parts, 0).to(rows.device), None


def all_bundle_ids(n_total, world):
    """ids of the gathered forward rows (gather_forward's row order)."""
    return np.concatenate([bundle_ids(n_total, world, o) for o in range(world)])


def forward_rows(table):
    """[2*n_pad, ...] per-rank table (entry 2j = forward strand of local read j) -> contiguous [n_pad, ...] forward rows."""
    return table.view((table.shape[0] // 2, 2) + tuple(table.shape[1:]))[:, 0].contiguous()


def bundle_ids(n_total, world, origin):
    """ids of the forward query rows of rank `origin`'s bundle (local read slot j is global read j*world + origin, 1-based ids)."""
    n_pad = shard_size(n_total, world)
    return np.arange(n_pad, dtype=np.int64) * world + origin + 1


def local_entry_ids(n_total, world, rank):
    """(ids, is_fwd) of rank `rank`'s own index entries (both strands of its reads)."""
    ids = np.repeat(bundle_ids(n_total, world, rank), 2)
    fwd = np.tile(np.array([1, 0], dtype=np.uint8), shard_size(n_total, world))
    return ids, fwd


def ring_post(bundle, world, rank, dist):
    """Start passing `bundle` (a tuple of tensors) to rank+1 and receiving the next one from rank-1.
    Returns (requests, received tensors); call ring_wait before touching the received tensors."""
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    if bundle[0].is_cuda and dist.get_backend() == "nccl":       # RCCL point-to-point over xGMI
        recv = tuple(torch.empty_like(t) for t in bundle)
        ops = [dist.P2POp(dist.isend, t, nxt) for t in bundle] + [dist.P2POp(dist.irecv, t, prv) for t in recv]
        return dist.batch_isend_irecv(ops), recv, None
    send_cpu = tuple(t.cpu().contiguous() for t in bundle)       # gloo (CPU tests; functional multi-rank runs on one GPU)
    recv_cpu = tuple(torch.empty_like(t) for t in send_cpu)
    reqs = [dist.isend(t, nxt) for t in send_cpu] + [dist.irecv(t, prv) for t in recv_cpu]
    return reqs, recv_cpu, bundle[0].device


def ring_wait(reqs, recv, device):
    for r in reqs:
        r.wait()
    return recv if device is None else tuple(t.to(device) for t in recv)



def shard_size(n_total, world):
    """Equal shard size (reads per rank) — shards are padded with zero-length reads, which sketch to status 2."""
    return (n_total + world - 1) // world


def pad_shard(fasta, n_total, world):
    """Pad a rank's FastaData (reads rank, rank+world, ...) to shard_size() with zero-length placeholder reads."""
    from .api import FastaData
    n_pad = shard_size(n_total, world)
    pad = n_pad - len(fasta)
    if pad <= 0:
        return fasta
    return FastaData(fasta.bases, np.concatenate([fasta.offsets, np.zeros(pad, np.int64)]),
                     np.concatenate([fasta.lengths, np.zeros(pad, np.int32)]),
                     np.concatenate([fasta.ids, np.zeros(pad, dtype=np.int64)]))


def global_entry_ids(n_total, world):
    """(ids, is_fwd) host arrays of the gathered index in global read order (padding reads get ids past n_total)."""
    n_pad = shard_size(n_total, world)
    ids = np.repeat(np.arange(1, n_pad * world + 1, dtype=np.int64), 2)
    fwd = np.tile(np.array([1, 0], dtype=np.uint8), n_pad * world)
    return ids, fwd


def gather_global_order(local, world, dist=None):
    """All-gather a per-rank table [2*n_pad, ...] and return it as [2*n_pad*world, ...] in global read order."""
    if world == 1:
        return local
    n_pad = local.shape[0] // 2
    tail = tuple(local.shape[1:])
    if local.is_cuda and dist.get_backend() == "nccl":      # RCCL over xGMI
        gathered = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(gathered.view(world, -1), local.contiguous().view(1, -1))
    else:                                                    # gloo (CPU tests; functional multi-rank runs on one GPU)
        src = local.contiguous().cpu()
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src)
        gathered = torch.stack(parts, 0).to(local.device)
    g = gathered.view(world, n_pad, 2, -1).permute(1, 0, 2, 3).contiguous()    # [slot][rank][strand] = read slot*world+rank
    return g.view((n_pad * world * 2,) + tail)


def rank_major_entry_ids(n_total, world):
    """(ids, is_fwd) of the gathered index in RANK-MAJOR order: entry rank*2*n_pad + 2*slot + strand is read slot*world+rank.
    This is the layout all_gather_into_tensor produces by itself, so no re-layout copy of the tables is needed; ids are not
    monotonic in entry order (only the brute-force candidate kernel's tile skipping cares, the inverted index does not)."""
    n_pad = shard_size(n_total, world)
    slot = np.arange(n_pad, dtype=np.int64)
    ids = np.concatenate([np.repeat(slot * world + r + 1, 2) for r in range(world)])
    fwd = np.tile(np.array([1, 0], dtype=np.uint8), n_pad * world)
    return ids, fwd


def gather_rank_major(local, world, dist=None):
    """All-gather a per-rank table [2*n_pad, ...] into [world*2*n_pad, ...], rank after rank (no re-layout)."""
    if world == 1:
        return local
    if local.is_cuda and dist.get_backend() == "nccl":      # RCCL over xGMI
        gathered = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(gathered.view(world, -1), local.contiguous().view(1, -1))
        return gathered
    src = local.contiguous().cpu()                           # gloo (CPU tests; functional multi-rank runs on one GPU)
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src)
    return torch.cat(parts, 0).to(local.device)


def rank_major_query_range(n_total, world, rank):
    """(first entry, entry count) of rank `rank`'s own reads in the rank-major index = the queries it searches."""
    n_pad = shard_size(n_total, world)
    return rank * 2 * n_pad, 2 * n_pad


def shard_query_reads(n_total, world, rank):
    """0-based global read indices this rank searches (round-robin balances the triangular id rule)."""
    return np.arange(rank, shard_size(n_total, world) * world, world, dtype=np.int64)
