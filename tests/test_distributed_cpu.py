"""N>1 path on CPU: two gloo processes deal reads round-robin, exchange their per-rank sketch tables with the same
all-gather + global-order re-layout the GPU path uses (mhap_amd/distributed.py), and the result must equal the
single-process table; the per-rank query shards must partition the reads."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_TOTAL, LEN, H, WORLD = 11, 400, 16, 2   # odd read count: exercises the padded shard


def _oracle_rows(fa):
    import oracle_lib as O
    rows = np.zeros((2 * len(fa), H), dtype=np.int32)
    for i in range(len(fa)):
        if fa.lengths[i] == 0:
            continue                                   # padding read: status 2, row stays zero
        s = fa.sequence(i)
        rows[2 * i] = O.minhash(s, 16, H)[1]
        rows[2 * i + 1] = O.minhash(O.rc(s), 16, H)[1]
    return rows


def _worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    import mhap_amd
    from mhap_amd import distributed as md
    shard = mhap_amd.synth_reads(N_TOTAL, LEN, seed=9, shard=rank, nshards=WORLD)
    assert shard.ids.tolist() == [r + 1 for r in range(rank, N_TOTAL, WORLD)]
    shard = md.pad_shard(shard, N_TOTAL, WORLD)
    assert len(shard) == md.shard_size(N_TOTAL, WORLD)
    local = torch.from_numpy(_oracle_rows(shard))
    g = md.gather_global_order(local, WORLD, dist)
    g_rm = md.gather_rank_major(local, WORLD, dist)
    mine = md.shard_query_reads(N_TOTAL, WORLD, rank)
    allq = [torch.zeros(len(mine), dtype=torch.int64) for _ in range(WORLD)]
    dist.all_gather(allq, torch.from_numpy(mine))
    if rank == 0:
        torch.save({"table": g, "table_rm": g_rm, "queries": torch.stack(allq)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process(tmp_path):
    import mhap_amd
    from mhap_amd import distributed as md
    out = str(tmp_path / "g.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(port, out), nprocs=WORLD, join=True)
    got = torch.load(out)
    full = mhap_amd.synth_reads(N_TOTAL, LEN, seed=9)
    want = _oracle_rows(full)
    n_pad = md.shard_size(N_TOTAL, WORLD)
    table = got["table"].numpy()
    assert table.shape == (2 * n_pad * WORLD, H)
    assert np.array_equal(table[:2 * N_TOTAL], want)          # global read order, fwd/rc interleaved
    assert not table[2 * N_TOTAL:].any()                       # padding entries
    ids, fwd = md.global_entry_ids(N_TOTAL, WORLD)
    assert ids[:2 * N_TOTAL].tolist() == np.repeat(full.ids, 2).tolist() and fwd[:4].tolist() == [1, 0, 1, 0]
    q = got["queries"].numpy()
    assert sorted(q.ravel().tolist()) == list(range(n_pad * WORLD))   # query shards partition the reads
    assert all((q[r] % WORLD == r).all() for r in range(WORLD))
    # rank-major layout (what bench.py uses: no re-layout copy): entry rank*2*n_pad + 2*slot + strand = read slot*WORLD + rank
    rm = got["table_rm"].numpy()
    ids_rm, fwd_rm = md.rank_major_entry_ids(N_TOTAL, WORLD)
    assert rm.shape == table.shape and len(ids_rm) == rm.shape[0]
    for e in range(rm.shape[0]):
        r = int(ids_rm[e]) - 1
        if r < N_TOTAL:
            assert np.array_equal(rm[e], want[2 * r + (1 - int(fwd_rm[e]))]), e
        else:
            assert not rm[e].any()
    firsts = [md.rank_major_query_range(N_TOTAL, WORLD, r) for r in range(WORLD)]
    assert firsts == [(r * 2 * n_pad, 2 * n_pad) for r in range(WORLD)]
    assert sorted(set(int(i) for i in ids_rm if i <= N_TOTAL)) == list(range(1, N_TOTAL + 1))


# ---- query rotation (what bench.py's N > 1 step does): ring helpers over gloo, 3 ranks --------------------------------------
RING_WORLD = 3


def _ring_worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=RING_WORLD)
    import mhap_amd
    from mhap_amd import distributed as md
    shard = md.pad_shard(mhap_amd.synth_reads(N_TOTAL, LEN, seed=9, shard=rank, nshards=RING_WORLD), N_TOTAL, RING_WORLD)
    local = torch.from_numpy(_oracle_rows(shard))                    # [2*n_pad, H], fwd/rc interleaved
    cur = (md.forward_rows(local), torch.full((md.shard_size(N_TOTAL, RING_WORLD), 4), rank, dtype=torch.int32))
    seen = []
    for t in range(RING_WORLD):
        pending = md.ring_post(cur, RING_WORLD, rank, dist) if t + 1 < RING_WORLD else None
        origin = (rank - t) % RING_WORLD
        assert int(cur[1][0, 0]) == origin                             # the bundle visiting at step t comes from rank - t
        seen.append((origin, cur[0].clone(), md.bundle_ids(N_TOTAL, RING_WORLD, origin)))
        if pending is not None:
            cur = md.ring_wait(*pending)
    ids, fwd = md.local_entry_ids(N_TOTAL, RING_WORLD, rank)
    torch.save({"seen": seen, "ids": ids, "fwd": fwd}, out + str(rank))
    dist.barrier()
    dist.destroy_process_group()


def test_query_bundles_visit_every_rank(tmp_path):
    import mhap_amd
    from mhap_amd import distributed as md
    out = str(tmp_path / "ring")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_ring_worker, args=(port, out), nprocs=RING_WORLD, join=True)
    full = mhap_amd.synth_reads(N_TOTAL, LEN, seed=9)
    want = _oracle_rows(full)
    all_ids = []
    for rank in range(RING_WORLD):
        got = torch.load(out + str(rank), weights_only=False)
        assert sorted(o for o, _, _ in got["seen"]) == list(range(RING_WORLD))      # every rank's queries came by exactly once
        for origin, rows, qids in got["seen"]:
            for j, qid in enumerate(qids):
                if qid <= N_TOTAL:
                    assert np.array_equal(rows[j].numpy(), want[2 * (int(qid) - 1)]), (rank, origin, j)   # forward row of read qid
                else:
                    assert not rows[j].any()
        assert got["fwd"].tolist() == [1, 0] * md.shard_size(N_TOTAL, RING_WORLD)
        all_ids += [int(i) for i in got["ids"][::2] if i <= N_TOTAL]
    assert sorted(all_ids) == list(range(1, N_TOTAL + 1))                            # the per-rank indexes partition the reads
