#!/usr/bin/env python3
"""bench.py — MHAP hot path on MI355X: self-overlap of synthetic PacBio-style reads (BASELINE.json configs[1]).

A "step" is ONE full pass of the hot path over the whole data set: sketch every read (both strands: k-mer
murmur hashes, tf weights, weighted MinHash, ordered bottom-S sketch), build the index tables in HBM, all-pairs
candidate count, second-stage overlap scoring, accepted records delivered to the host.  The packed reads are
resident in HBM before the timed region starts (mhap_stage_reads); record text formatting is outside it.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1 (strong scaling, same 100k reads): rank r sketches reads r, r+N, ...; the per-rank sketch tables are
all-gathered with RCCL over xGMI; every rank searches its round-robin query shard against the full index.
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import mhap_amd  # noqa: E402
from mhap_amd import MhapParams, MinHashSearch  # noqa: E402
from mhap_amd import distributed as mdist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# MinHash instruction ceilings measured on MI355X (profiles/r01_valu_microbench.txt):
XORSHIFT_CEILING_PER_CHAIN = 4.79e12   # tools/valu_peak.hip: one chain per lane-register pair (2 v_lshlrev_b64 + 6 ops per step)
VALU_FULL_RATE = 6.5e13               # lane-ops/s of a full-rate 32-bit VALU op chip-wide (v_fma_f32 / v_xor_b32, tools/valu_ops.hip)
BITSLICED_OPS_PER_32_STEPS = 107 + 15  # 43 v_xor + 64 v_xor/v_bitop3 per step of 32 chains + ~15 ops of candidate filter
XORSHIFT_CEILING_BITSLICED = 32 * VALU_FULL_RATE / BITSLICED_OPS_PER_32_STEPS   # 1.70e13 steps/s


def sketch_bytes_per_read(L, H, S, k2):
    """packed read once + both MinHash rows + both ordered rows."""
    sp = min(S, L - k2 + 1)
    return (L + 3) // 4 + 2 * 4 * H + 2 * 8 * sp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--length", type=int, default=10000)
    ap.add_argument("--hashes", type=int, default=512)
    ap.add_argument("--error-rate", type=float, default=0.15)
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="0 = auto (about 10-30 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # MHAP_BENCH_BACKEND=gloo lets several ranks share one GPU (functional test of the N>1 path without RCCL)
    backend = os.environ.get("MHAP_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or (os.environ.get("MHAP_BENCH_FORCE_DIST") and "RANK" in os.environ):   # (1 rank under torchrun: RCCL path on one GPU)
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    n_total, L, H, S, k, k2 = args.reads, args.length, args.hashes, 1536, 16, 12
    p = MhapParams(kmer_size=k, num_hashes=H, ordered_kmer_size=k2, ordered_sketch_size=S, device=local_rank)
    seed = 0x4D484150 ^ 2
    t_gen = time.time()
    fa = mhap_amd.synth_reads(n_total, L, seed=seed, error_rate=args.error_rate, shard=rank, nshards=world)
    n_local = len(fa)
    n_pad = mdist.shard_size(n_total, world)    # equal shard size for the all-gather (pad = zero-length reads)
    fa = mdist.pad_shard(fa, n_total, world)
    t_gen = time.time() - t_gen

    ms = MinHashSearch(p)
    t_stage = time.perf_counter()
    ms.stage(fa)                                  # 2-bit pack on the host + H2D: packed reads now resident in HBM
    t_stage = time.perf_counter() - t_stage
    dev = torch.device("cuda", local_rank)
    force_dist = world > 1 or bool(os.environ.get("MHAP_BENCH_FORCE_DIST"))   # 1 rank through the N>1 code path (RCCL on one GPU)
    if force_dist:
        loc_mh = torch.zeros((2 * n_pad, H), dtype=torch.int32, device=dev)
        loc_od = torch.zeros((2 * n_pad, S, 2), dtype=torch.int32, device=dev)
        loc_mt = torch.zeros((2 * n_pad, 4), dtype=torch.int32, device=dev)
        gids, gfwd = mdist.rank_major_entry_ids(n_total, world)
        q_first, q_count = mdist.rank_major_query_range(n_total, world, rank)
    async_gather = dist is not None and backend == "nccl" and os.environ.get("MHAP_BENCH_ASYNC_GATHER", "1") != "0"

    def step():
        ms.clear()
        if not force_dist:
            ms.add_staged()
            recs = ms.find_matches()
        else:
            ms.sketch_staged_device(loc_mh.data_ptr(), loc_od.data_ptr(), loc_mt.data_ptr())
            g_mh = mdist.gather_rank_major(loc_mh, world, dist)      # RCCL all-gather, tables stay rank after rank
            g_mt = mdist.gather_rank_major(loc_mt, world, dist)
            if async_gather:
                # the ordered-sketch table (85 % of the bytes) is only read by the second stage: its all-gather runs while this
                # rank builds the inverted index from the MinHash table
                torch.cuda.current_stream().synchronize()
                g_od = torch.empty((world * loc_od.shape[0],) + tuple(loc_od.shape[1:]), dtype=loc_od.dtype, device=dev)
                work = dist.all_gather_into_tensor(g_od.view(world, -1), loc_od.view(1, -1), async_op=True)
                ms.set_device_index(gids, gfwd, g_mh.data_ptr(), g_od.data_ptr(), g_mt.data_ptr())
                ms.prepare_index()
                work.wait()
            else:
                g_od = mdist.gather_rank_major(loc_od, world, dist)
            torch.cuda.synchronize()
            if not async_gather:
                ms.set_device_index(gids, gfwd, g_mh.data_ptr(), g_od.data_ptr(), g_mt.data_ptr())
            recs = ms.find_matches(q_first, q_count)                  # this rank's own reads against the whole index
            step.keep = (g_mh, g_od, g_mt)
        return recs

    def fence():
        ms.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ms.reset_kernel_times()
    fence()
    t0 = time.perf_counter()
    nrec = 0
    for _ in range(args.steps):
        recs = step()
        nrec = len(recs)
    fence()
    elapsed = time.perf_counter() - t0
    st = ms.stats()
    kt = ms.kernel_times()
    # output fingerprint of the last step (outside the timed region): SHA-256 of the sorted record lines on one GPU, and an
    # order- and shard-independent checksum (sum of the first 8 digest bytes of every line, mod 2^64) that is comparable across N
    lines = sorted(mhap_amd.records_to_lines(recs))
    sha = hashlib.sha256("\n".join(lines).encode()).hexdigest() if world == 1 else None
    csum = 0
    for ln in lines:
        csum = (csum + int.from_bytes(hashlib.sha256(ln.encode()).digest()[:8], "little")) & ((1 << 64) - 1)

    rdev = dev if backend == "nccl" else torch.device("cpu")
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
    tot_rec = torch.tensor([nrec], dtype=torch.int64, device=rdev)
    csum_t = torch.tensor([csum & 0xFFFF, (csum >> 16) & 0xFFFF, (csum >> 32) & 0xFFFF, csum >> 48], dtype=torch.int64, device=rdev)   # 16-bit limbs
    kms = torch.tensor([kt[kname]["ms"] for kname in mhap_amd.KERNEL_NAMES], dtype=torch.float64, device=rdev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot_rec, op=dist.ReduceOp.SUM)
        dist.all_reduce(kms, op=dist.ReduceOp.MAX)
        dist.all_reduce(csum_t, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    total_records = int(tot_rec.item())
    limbs = [int(v) for v in csum_t.tolist()]
    records_checksum = (limbs[0] + (limbs[1] << 16) + (limbs[2] << 32) + (limbs[3] << 48)) & ((1 << 64) - 1)
    sec_per_step = elapsed / max(args.steps, 1)

    if rank == 0:
        K = max(args.steps, 1)
        kernel_ms_per_step = {kname: float(kms[i].item()) / K for i, kname in enumerate(mhap_amd.KERNEL_NAMES)}
        sketch_ms = sum(kernel_ms_per_step[x] for x in ("hash_kmers", "kmer_weight", "minhash", "ordered"))
        search_ms = sum(kernel_ms_per_step[x] for x in ("candidate", "overlap", "index_build", "index_query"))
        # dominant kernel roofline
        dom = max(kernel_ms_per_step, key=kernel_ms_per_step.get)
        launches = max(kt[dom]["launches"], 1)
        avg_launch_s = kt[dom]["ms"] / launches / 1e3
        reads_per_launch = n_local * K / launches if dom in ("hash_kmers", "kmer_weight", "minhash", "ordered") else None
        if dom == "candidate":
            alg_bytes = (n_total + 2 * n_total) * 4 * H / world * K / launches      # §8(d): every sketch row read once per pass
        elif dom == "overlap":
            alg_bytes = st["candidates_compared"] * K / launches * 8 * S * 2
        elif dom in ("index_build", "index_query"):
            alg_bytes = (n_total + 2 * n_total) * 4 * H / world * K / launches
        else:
            alg_bytes = sketch_bytes_per_read(L, H, S, k2) * reads_per_launch
        achieved = alg_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        traffic, traffic_src = pmc_traffic(dom + "_kernel", n_total, L, world)
        roofline = {"bound": "hbm", "kernel": dom + "_kernel", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                    "alg_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(avg_launch_s * 1e3, 4), "launches": int(launches)}
        # measured HBM rate of every kernel (PMC bytes per launch x launches / its summed time): which ones are memory-side
        hbm_by_kernel = {}
        for kname in mhap_amd.KERNEL_NAMES:
            tb, _ = pmc_traffic(("overlap_join" if kname == "overlap" else kname) + "_kernel", n_total, L, world)
            if tb and kt[kname]["ms"] > 0:
                hbm_by_kernel[kname] = {"GB_per_step": round(tb * kt[kname]["launches"] / K / 1e9, 2),
                                        "GB_per_s": round(tb * kt[kname]["launches"] / (kt[kname]["ms"] / 1e3) / 1e9, 1)}
        # integer-VALU view of the MinHash kernel (the path is integer min-reduction work, not HBM-bound: SURVEY F12)
        steps_per_read = 2 * (L - k + 1) * H
        mh_s = kernel_ms_per_step["minhash"] / 1e3
        xs_rate = steps_per_read * n_local / mh_s if mh_s > 0 else 0.0
        valu = {"kernel": "minhash_kernel", "xorshift_steps_per_s": round(xs_rate, 1),
                "ceiling_steps_per_s": XORSHIFT_CEILING_BITSLICED, "frac_of_ceiling": round(xs_rate / XORSHIFT_CEILING_BITSLICED, 4),
                "ceiling_note": "bit-sliced rows (32 chains per lane as 64 bit-planes): one step of 32 chains is 107 full-rate ops "
                                "(43 v_xor_b32 + 64 v_xor_b32/v_bitop3_b32) + ~15 ops of candidate filter; ceiling = 32 x 6.5e13 "
                                "lane-ops/s / 122 with every issue slot used.  The per-chain formulation (2 v_lshlrev_b64 + 6 ops "
                                "per step) tops out at 4.79e12 steps/s (tools/valu_peak.hip)",
                "vs_per_chain_ceiling": round(xs_rate / XORSHIFT_CEILING_PER_CHAIN, 4)}
        if kernel_ms_per_step["candidate"] > 0 and st["slot_compares"] > 0:   # stats are per step (the index is cleared every step)
            valu["candidate_slot_compares_per_s"] = round(st["slot_compares"] / (kernel_ms_per_step["candidate"] / 1e3), 1)

        out = {
            "metric": "overlaps/sec", "value": round(total_records / sec_per_step, 2), "unit": "overlaps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sec_per_step * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": f"{n_total} synthetic PacBio-style reads x {L} bp (30x, {args.error_rate:.0%} error), k={k}, "
                                   f"--num-hashes {H}, ordered sketch k2={k2} S={S}, self-overlap (BASELINE configs[1])",
                       "parallelism": f"reads round-robin over {world} GPU(s); RCCL all-gather of sketch tables" if world > 1 else "1 GPU"},
            "records_per_step": total_records,
            "sketches_per_sec": round(2 * n_total / (sketch_ms / 1e3), 1) if sketch_ms > 0 else None,
            "sketches_per_sec_note": "strands / summed sketch-kernel time (rank max)",
            "overlaps_per_sec_search_only": round(total_records / (search_ms / 1e3), 1) if search_ms > 0 else None,
            "kernel_ms_per_step": {kk: round(v, 3) for kk, v in kernel_ms_per_step.items()},
            "candidates_per_step": int(st["candidates_compared"]),
            "index_elements_per_step": int(st["table_elements"]),
            "overlap_slow_pairs_per_step": int(st["slow_pairs"]),
            "records_sha256_sorted_lines": sha, "records_checksum": "%016x" % records_checksum,
            "hbm_traffic_by_kernel": hbm_by_kernel,
            "roofline": roofline, "valu": valu,
            "input_gen_s": round(t_gen, 2),
            "staging_ms_untimed": round(t_stage * 1e3, 1),
            "value_incl_host_pack_and_pcie": round(total_records / (sec_per_step + t_stage), 2),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, L, H, S, k, k2)
        print(json.dumps(out), flush=True)
    ms.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(kernel, n_total, L, world):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (tools/pmc_summary.py).  The counters
    were collected on the default workload (100k x 10kb, 1 GPU); for any other shape the field stays null."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not (os.path.exists(path) and n_total == 100000 and L == 10000 and world == 1):
        return None, None
    try:
        d = json.load(open(path))["kernels"].get(kernel)
        return (int(d["hbm_bytes_per_launch"]), "profiles/r01_pmc_traffic.json (" + d["fetch_rule"] + ")") if d else (None, None)
    except Exception:
        return None, None


def cpu_baseline(args, L, H, S, k, k2):
    """The CPU oracle (a C++ restatement of the reference's Java, kind "port") timed on this box's host cores on a
    bounded sample of the same workload: same read length, coverage, error model and flags, fewer reads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = os.cpu_count() or 1
    n = args.cpu_sample_reads
    if n <= 0:
        # ~2*(L-k+1)*H xorshift steps per read at ~1.5e9 steps/s/core; aim at ~15 s
        per_read_s = 2.0 * (L - k + 1) * H / 1.5e9
        n = int(max(200, min(20000, 15.0 * cores / per_read_s)))
    fa = mhap_amd.synth_reads(n, L, seed=(0x4D484150 ^ 2) + 1, error_rate=args.error_rate)
    t = time.perf_counter()
    res = O.run_self(fa, k=k, H=H, k2=k2, S=S, nthreads=cores, cap=1 << 24)
    wall = time.perf_counter() - t
    nrec = len(res["records"])
    busy = res["sketch_s"] + res["search_s"]
    return {"value": round(nrec / busy, 2) if busy > 0 else None, "unit": "overlaps/s", "cores": cores, "kind": "port",
            "sample": f"{n} reads x {L} bp at the same 30x coverage/error model (own genome), same flags; "
                      f"{nrec} records in {busy:.2f} s ({res['sketch_s']:.2f} s sketch + {res['search_s']:.2f} s search)",
            "sketches_per_sec": round(2 * n / res["sketch_s"], 1) if res["sketch_s"] > 0 else None,
            "reads_per_sec": round(n / busy, 2) if busy > 0 else None, "wall_s": round(wall, 2),
            "note": "C++ restatement of MHAP's Java path (no JVM on this box), std::thread on all host cores"}


if __name__ == "__main__":
    main()
