"""The N > 1 path on CPU (world_size 2 and 3, gloo): the host-side helpers bench.py uses around the library's multi-GPU entry
points (mhap_amd/distributed.py), and the sharded search itself restated with the oracle's primitives — every rank keeps the
sketches of the reads it was dealt, the forward query rows of all ranks are all-gathered, every rank scores all queries against
its own shard under the toSelf id rules — whose union of records must equal the single-process oracle run.  (The GPU
implementation of the same scheme, mhap_dist_* / mhap_group_*, is checked on the device in tests/test_cli_gpu.py and
tests/test_gpu_parity.py.)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_TOTAL, LEN, H, S, K2 = 23, 1500, 64, 256, 12   # odd read count: unequal shards


def _sketch(fa, i):
    """(minhash fwd, minhash rc, ordered fwd (padded to S), ordered rc, sizes, seqlens) of read i, from the oracle."""
    import oracle_lib as O
    s = fa.sequence(i)
    out = []
    for strand in (s, O.rc(s)):
        _, mh = O.minhash(strand, 16, H)
        _, od, seqlen = O.ordered(strand, K2, S)
        pad = np.zeros((S, 2), dtype=np.int32)
        pad[:len(od)] = od
        out.append((mh, pad, len(od), seqlen))
    return out


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mhap_amd
    import oracle_lib as O
    from mhap_amd import distributed as md
    full = mhap_amd.synth_reads(N_TOTAL, LEN, seed=11, error_rate=0.08)
    shard = mhap_amd.synth_reads(N_TOTAL, LEN, seed=11, error_rate=0.08, shard=rank, nshards=world)
    # the deal: generator shards and shard_of agree, ids stay global
    mine = md.shard_of(full, world, rank)
    assert shard.ids.tolist() == mine.ids.tolist() == (md.shard_indices(N_TOTAL, world, rank) + 1).tolist()
    assert np.array_equal(shard.bases, mine.bases)
    # rank 0's communicator id reaches every rank unchanged
    uid = md.broadcast_unique_id(dist, rank, mhap_amd.MinHashSearch.dist_unique_id)
    assert len(uid) == 128
    uids = [None] * world
    dist.all_gather_object(uids, uid)
    assert all(u == uids[0] for u in uids)
    # this rank's tables (both strands of its reads)
    n_loc = len(shard)
    sk = [_sketch(shard, i) for i in range(n_loc)]
    # the exchange: forward rows padded to the largest shard, all-gathered (what mhap_dist_find_matches_self does over RCCL)
    n_pad = (N_TOTAL + world - 1) // world
    send_mh = torch.zeros((n_pad, H), dtype=torch.int32)
    send_od = torch.zeros((n_pad, S, 2), dtype=torch.int32)
    send_meta = torch.full((n_pad, 3), -1, dtype=torch.int64)          # id, ordered size, ordered seqlen; -1 = padding row
    for j in range(n_loc):
        mh, od, size, seqlen = sk[j][0]
        send_mh[j] = torch.from_numpy(mh); send_od[j] = torch.from_numpy(od)
        send_meta[j] = torch.tensor([int(shard.ids[j]), size, seqlen])
    g_mh = [torch.zeros_like(send_mh) for _ in range(world)]; dist.all_gather(g_mh, send_mh)
    g_od = [torch.zeros_like(send_od) for _ in range(world)]; dist.all_gather(g_od, send_od)
    g_meta = [torch.zeros_like(send_meta) for _ in range(world)]; dist.all_gather(g_meta, send_meta)
    q_mh = torch.cat(g_mh).numpy(); q_od = torch.cat(g_od).numpy(); q_meta = torch.cat(g_meta).numpy()
    # the search: every gathered forward query against this rank's stored strands, toSelf rules (MinHashSearch.java:200-225)
    lines = []
    for q in range(q_mh.shape[0]):
        qid, qsize, qlen = (int(v) for v in q_meta[q])
        if qid < 0:
            continue
        for j in range(n_loc):
            mid = int(shard.ids[j])
            if mid >= qid:                      # same read, or the pair belongs to the rank that stores the lower id
                continue
            assert md.pair_owner(qid, mid, world) == rank
            for strand in (0, 1):
                mh, od, size, seqlen = sk[j][strand]
                if int((mh == q_mh[q]).sum()) < 3:
                    continue
                ov = O.overlap(q_od[q][:qsize], qlen, od[:size], seqlen, K2, 0.2)
                if ov["empty"] or ov["score"] < 0.78:
                    continue
                b1, b2 = ov["b1"], ov["b2"]
                if strand:
                    b1, b2 = LEN - ov["b2"] - 1, LEN - ov["b1"] - 1
                rec = {"from_id": qid, "to_id": mid, "score": ov["score"], "raw": ov["raw"], "a1": ov["a1"], "a2": ov["a2"], "alen": LEN,
                       "b1": b1, "b2": b2, "blen": LEN, "to_rc": strand}
                lines.append(O.format_record(rec))
    gathered = [None] * world
    dist.all_gather_object(gathered, lines)
    if rank == 0:
        torch.save({"lines": gathered}, out)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, tmp_path, port_base):
    import mhap_amd
    import oracle_lib as O
    out = str(tmp_path / f"w{world}.pt")
    port = port_base + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=False)["lines"]
    full = mhap_amd.synth_reads(N_TOTAL, LEN, seed=11, error_rate=0.08)
    want = O.record_lines(O.run_self(full, H=H, S=S, nthreads=2)["records"])
    assert len(want) > 10
    union = sorted(ln for part in got for ln in part)
    assert union == want                                   # every pair reported exactly once, by exactly one rank
    assert sum(1 for part in got if part) == world         # and every rank had work


def test_sharded_search_two_ranks_equals_single_process(tmp_path):
    _run(2, tmp_path, 29500)


def test_sharded_search_three_ranks_equals_single_process(tmp_path):
    _run(3, tmp_path, 31500)
