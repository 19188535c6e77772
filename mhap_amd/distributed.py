"""One process per GPU: what a host does AROUND the library's multi-GPU entry points.

The exchange itself lives in libmhaphip.so (mhap_dist_*: pack the forward query rows, RCCL all-gather over xGMI, search against the
rank's own index shard).  A host only has to (1) deal the reads — read i of the data set goes to rank i % world, which balances the
id < id rule of J/impl/MinHashSearch.java:215-219 — and (2) carry rank 0's 128-byte communicator id to the other ranks over
whatever channel it has.  bench.py uses torch.distributed for (2) (and for its barrier / max-over-ranks timing); these helpers are
what it calls, and tests/test_distributed_cpu.py runs them over gloo.
"""
import numpy as np


def shard_indices(n_total, world, rank):
    """0-based indices of the reads rank `rank` holds: rank, rank + world, ... (ids are index + 1, FastaData.java:180-181)."""
    return np.arange(rank, n_total, world, dtype=np.int64)


def shard_of(fasta, world, rank):
    """Rank `rank`'s share of a FastaData holding the whole data set (global ids kept)."""
    return fasta.subset(shard_indices(len(fasta), world, rank)) if world > 1 else fasta


def broadcast_unique_id(dist, rank, make_id):
    """Rank 0 creates the communicator id (make_id(): MinHashSearch.dist_unique_id) and every rank returns the same 128 bytes.
    Works on any torch.distributed backend (a uint8 tensor broadcast; NCCL/RCCL groups need it on the device)."""
    import torch
    buf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = torch.tensor(list(make_id()), dtype=torch.uint8)
    if dist is None:
        return bytes(buf.tolist())
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = buf.to(dev)
    dist.broadcast(t, src=0)
    return bytes(t.cpu().tolist())


def pair_owner(id_a, id_b, world):
    """The rank that reports the overlap of reads id_a and id_b (1-based ids): the one that stores the lower id
    (the query must have the larger id, J/impl/MinHashSearch.java:215-219, and a stored read lives on exactly one rank)."""
    return (min(id_a, id_b) - 1) % world
