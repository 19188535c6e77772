#!/usr/bin/env python3
"""bench.py — MHAP hot path on MI355X: self-overlap of synthetic PacBio-style reads (default: BASELINE.json configs[1]).

A "step" is ONE full pass of the hot path over the whole data set: sketch every read (both strands: k-mer
murmur hashes, tf / tf-idf weights, weighted MinHash, ordered bottom-S sketch), build the index tables in HBM, inverted-index
candidate count, second-stage overlap scoring, accepted records delivered to the host.  The packed reads are
resident in HBM before the timed region starts (mhap_stage_reads); record text formatting is outside it.

  python bench.py --gpus 1 --steps K --warmup W [--config c1|c2|c3|c4|c4slice|c5slice]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1 (strong scaling, same data set): rank r sketches reads r, r+N, ... and indexes them — only them; the exchange is inside
the library (mhap_dist_find_matches_self: the forward query sketches of all ranks are all-gathered over an RCCL communicator the
library creates — the ordered rows asynchronously, under the candidate stage — and every rank searches them against its own index).
torch.distributed only carries rank 0's communicator id to the other ranks and does the barrier / max-over-ranks timing.
Prints ONE JSON line on rank 0.  At N = 1 the line also carries
  parity_check : the GPU path and the CPU oracle run on the SAME sample reads, sorted-record SHA-256 compared;
  cpu_baseline : the oracle timed on this box's host cores on that sample (threads = the cgroup CPU quota);
  end_to_end   : the native driver (mhap-hip) from FASTA open to the last record flushed, on the bench data set.
"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import mhap_amd  # noqa: E402
from mhap_amd import MhapParams, MinHashSearch  # noqa: E402
from mhap_amd import distributed as mdist  # noqa: E402
from mhap_amd import workloads as W  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# Integer ceilings of the MinHash kernel (it is VALU-bound, not HBM-bound: SURVEY F12).
VALU_SPEC = 256 * 4 * 32 * 2.4e9       # lane-ops/s at spec: 256 CUs x 4 SIMD-32 x 2.4 GHz = 7.86e13
VALU_MEASURED = 6.5e13                 # tools/valu_ops.hip on MI355X (profiles/r01_valu_microbench.txt): sustained clock under an all-VALU load
XORSHIFT_CEILING_PER_CHAIN = 4.79e12   # tools/valu_peak.hip: one chain per lane-register pair (2 v_lshlrev_b64 + 6 ops per step)
BITSLICED_OPS_PER_32_STEPS = 92 + 9 + 4   # 28 v_xor + 64 v_xor/v_bitop3 per step of 32 chains (tools/gen_bs_step92.py) + 9 ops of depth filter + compare/loop


def sketch_bytes_per_read(L, H, S, k2):
    """packed read once + both MinHash rows + both ordered rows."""
    sp = min(S, L - k2 + 1)
    return (L + 3) // 4 + 2 * 4 * H + 2 * 8 * sp


def host_cpus():
    """(affinity, cgroup quota in cores or None, threads to use).  `nproc` / os.cpu_count() report the machine; the container's
    cgroup cpu.max is what the process can actually burn."""
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    use = aff if quota is None else max(1, min(aff, int(round(quota))))
    return aff, quota, use


def sorted_sha(lines):
    return hashlib.sha256("\n".join(sorted(lines)).encode()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(W.CONFIGS) + ["c3"])
    ap.add_argument("--reads", type=int, default=0, help="override the configuration's read count")
    ap.add_argument("--length", type=int, default=0, help="override the configuration's read length")
    ap.add_argument("--hashes", type=int, default=0)
    ap.add_argument("--error-rate", type=float, default=0.15)
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="0 = auto (about 15-20 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip parity_check / cpu_baseline / end_to_end (kernel A/B runs)")
    ap.add_argument("--soak-seconds", type=float, default=-1.0,
                    help="after the timed region: repeat the step, untimed for `value`, for about this long and report the per-step spread "
                         "(settled clocks; also keeps the GPU busy long enough for a 5 s utilisation sampler to see it).  Default 8, "
                         "0 with --no-cpu-baseline")
    ap.add_argument("--smi-trace", default="",
                    help="write a clock / power trace of THIS rank's card (tools/smi_trace.py, picked by its PCI address) over the soak leg to "
                         "this file and put its summary into the line's `soak.smi`; a capture without clock or power samples is reported as failed")
    ap.add_argument("--preflight", action="store_true",
                    help="check what an N-GPU run needs (devices, RCCL entry points, peer access, rendezvous variables) and print why "
                         "it cannot run instead of hanging in a collective; exit code 0 = ready")
    ap.add_argument("--exchange-only", action="store_true",
                    help="the exchange of the chosen configuration by itself (VERDICT r05 item 6): every rank's real row volumes are all-gathered "
                         "(a) with NO compute beside them (mhap_dist_selftest with the bytes of the ordered rows / of the MinHash + meta + id rows) and "
                         "(b) by the eager add, UNDER the MinHash kernel (events on the exchange stream: mhap_dist_exchange_timing), with the part of "
                         "(b) the search still had to wait for; one JSON line with every rank's numbers: fabric time and the interaction with a "
                         "power-bound kernel come apart in the first record of a real N-GPU run.  Launch like the timed run")
    ap.add_argument("--dry-collective", action="store_true",
                    help="form the N-rank communicator exactly as the timed run does (torch only carries the unique id), all-gather 1 MB "
                         "per rank through the library's transport, check every rank's block and peer access, print one JSON line with "
                         "every rank's view (ncclCommCount / ncclCommUserRank / device PCI id) and exit: a fabric or rendezvous problem "
                         "shows here, not as a hang of the compute step.  Launch like the timed run (torch.distributed.run for N > 1)")
    args = ap.parse_args()
    if args.preflight:
        raise SystemExit(preflight(args.gpus))
    if args.dry_collective:
        raise SystemExit(dry_collective(args.gpus))
    if args.exchange_only:
        raise SystemExit(exchange_only(args))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if world > 1 and torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible — one rank per GPU "
                         "(python bench.py --gpus N --preflight says what is missing)")
    backend = "nccl"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or (os.environ.get("MHAP_BENCH_FORCE_DIST") and "RANK" in os.environ):   # (1 rank under torchrun: RCCL path on one GPU)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    S, k, k2 = 1536, 16, 12
    flt = None
    filter_path = None
    tmpdir = tempfile.mkdtemp(prefix="mhap_bench_")
    t_gen = time.time()
    if args.config == "c3":
        # BASELINE configs[2]: real E. coli reads are in neither tree; they are supplied on the box through MHAP_C3_FASTA
        path = W.c3_fasta_path()
        if path is None:
            if rank == 0:
                print(json.dumps({"metric": "overlaps/sec", "value": None, "unit": "overlaps/s", "n_gpus": world,
                                  "config": {"workload": "E. coli PacBio P6-C4 reads (BASELINE configs[2])"},
                                  "c3": "C3 skipped: MHAP_C3_FASTA is not set / not readable"}), flush=True)
            return
        whole = mhap_amd.FastaData.from_file(path)
        n_total, H = len(whole), args.hashes or 512
        L = int(np.median(whole.lengths))
        idx = np.arange(rank, n_total, world)
        fa = whole.subset(idx) if world > 1 else whole
        cfg_label = f"{n_total} reads of {os.path.basename(path)} (median {L} bp), default flags (BASELINE configs[2])"
        cfg = dict(repeats=None, filter=False, seed=0)
    else:
        cfg = W.CONFIGS[args.config]
        n_total, L, H = args.reads or cfg["reads"], args.length or cfg["length"], args.hashes or cfg["hashes"]
        fa = W.config_reads(args.config, shard=rank, nshards=world, reads=n_total, length=L, error_rate=args.error_rate)
        cfg_label = cfg["label"] if (n_total, L, H) == (cfg["reads"], cfg["length"], cfg["hashes"]) else \
            f"{n_total} synthetic reads x {L} bp, --num-hashes {H} ({args.config} generator)"
        if cfg["filter"]:
            # the -f file of configs[4]: k-mer counts of a sample of the reads (every rank derives the same file)
            stride = max(1, n_total // 2000)      # reads 0, stride, 2 stride, ... of the same data set
            head = fa if world == 1 else W.config_reads(args.config, shard=0, nshards=stride, reads=n_total, length=L, error_rate=args.error_rate)
            filter_path = os.path.join(tmpdir, "kmers.txt")
            W.write_filter_file(head, filter_path, max_reads=2000)
            flt = mhap_amd.FrequencyCounts.from_file(filter_path, filter_cutoff=1e-5, repeat_weight=0.9)
    p = MhapParams(kmer_size=k, num_hashes=H, ordered_kmer_size=k2, ordered_sketch_size=S, device=local_rank)
    n_local = len(fa)
    fa_bench = fa
    t_gen = time.time() - t_gen

    ms = MinHashSearch(p, kmer_filter=flt)
    t_stage = time.perf_counter()
    ms.stage(fa)                                  # 2-bit pack on the host + H2D: packed reads now resident in HBM
    t_stage = time.perf_counter() - t_stage
    dev = torch.device("cuda", local_rank)
    force_dist = world > 1 or bool(os.environ.get("MHAP_BENCH_FORCE_DIST"))   # 1 rank through the N>1 code path (RCCL on one GPU)
    eager_note = None
    if force_dist:
        os.environ.setdefault("MHAP_DIST_TIMEOUT_S", "600")   # (a rank that never arrives makes the others give up, not hang: mhap_dist.hip)
        # the ranks form their communicator inside the library (ncclCommInitRank); the host only hands rank 0's id round
        ms.dist_init(rank, world, mdist.broadcast_unique_id(dist, rank, MinHashSearch.dist_unique_id))
    rank_views = None
    if force_dist:
        rank_views = gather_rank_views(ms, dist, rank, world, local_rank)
    phase = {"sketch": 0.0, "exchange": 0.0, "search": 0.0}

    def step(timed_phases=False):
        ms.clear()
        t0 = time.perf_counter()
        if not force_dist:
            ms.add_staged()
            if timed_phases:
                ms.synchronize()
                phase["sketch"] += time.perf_counter() - t0
                t0 = time.perf_counter()
            recs = ms.find_matches()
            if timed_phases:
                phase["search"] += time.perf_counter() - t0
            return recs
        # N > 1: this rank's reads are sketched into its own tables and indexed here and nowhere else (1/N of the inverted-index
        # build, filled while the reads are sketched); the collective search gathers every rank's forward query rows inside the library
        ms.add_staged()
        if timed_phases:
            ms.synchronize()
            phase["sketch"] += time.perf_counter() - t0
            t0 = time.perf_counter()
        recs = ms.dist_find_matches()
        if timed_phases:
            tm = ms.dist_last_timing()
            exch = (tm["gather_small_ms"] + tm["wait_ordered_ms"]) / 1e3    # what of the gather the compute did not hide
            phase["exchange"] += exch
            phase["search"] += time.perf_counter() - t0 - exch
        return recs

    def fence():
        ms.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def job_checksum(recs):
        """order- and shard-independent fingerprint of a step's records over all ranks (binary fields; 16-bit limbs through all_reduce)"""
        m64 = np.zeros(len(recs), dtype=np.uint64)
        with np.errstate(over="ignore"):
            for f in ("from_id", "to_id", "a1", "a2", "b1", "b2", "to_rc"):
                m64 = (m64 ^ recs[f].astype(np.int64).view(np.uint64)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x632BE59BD9B4E019)
                m64 ^= m64 >> np.uint64(29)
            c = int(m64.sum(dtype=np.uint64)) if len(recs) else 0
        t = torch.tensor([len(recs), c & 0xFFFF, (c >> 16) & 0xFFFF, (c >> 32) & 0xFFFF, c >> 48], dtype=torch.int64, device=dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return tuple(int(x) for x in t.tolist())

    if force_dist and not os.environ.get("MHAP_BENCH_NO_EAGER"):
        # Eager exchange (mhap_dist_set_eager): the add gathers the rank's rows while it computes.  It is switched on only after one
        # step each way has produced the same records over all ranks; otherwise the run goes on with the exchange at search time.
        ref = job_checksum(step())
        ms.dist_set_eager(True)
        got = job_checksum(step())
        if got == ref:
            eager_note = "on (records equal to the exchange-at-search-time step: %d)" % ref[0]
        else:
            ms.dist_set_eager(False)
            eager_note = "off: the eager step's records differed (%s vs %s)" % (got, ref)
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
    # one more, untimed, step with a fence between the phases: wall time of the sketch phase (packed reads in HBM -> both tables in
    # HBM, host bookkeeping included), of the table exchange (N > 1) and of the search phase (tables in HBM -> records on the host)
    step(timed_phases=True)
    fence()
    # output fingerprint of the last step (outside the timed region): SHA-256 of the sorted record lines on one GPU, and an
    # order- and shard-independent checksum (sum of the first 8 digest bytes of every line, mod 2^64) that is comparable across N
    # (beyond a few million records the text of every line is not formed in Python: `records_checksum` is then an order-independent
    #  mix of the binary record fields — comparable across N and across runs of the same build, not with the text-based one)
    big_output = len(recs) > 4_000_000
    if big_output:
        sha = None
        m64 = np.zeros(len(recs), dtype=np.uint64)
        with np.errstate(over="ignore"):
            for f in ("from_id", "to_id", "a1", "a2", "alen", "b1", "b2", "blen", "to_rc"):
                m64 = (m64 ^ recs[f].astype(np.int64).view(np.uint64)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x632BE59BD9B4E019)
                m64 ^= m64 >> np.uint64(29)
            for f in ("score", "raw"):
                m64 = (m64 ^ np.ascontiguousarray(recs[f], dtype=np.float64).view(np.uint64)) * np.uint64(0xBF58476D1CE4E5B9) + np.uint64(1)
                m64 ^= m64 >> np.uint64(31)
            csum = int(m64.sum(dtype=np.uint64))
        lines = None
    else:
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
    pht = torch.tensor([phase["sketch"], phase["exchange"], phase["search"]], dtype=torch.float64, device=rdev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot_rec, op=dist.ReduceOp.SUM)
        dist.all_reduce(kms, op=dist.ReduceOp.MAX)
        dist.all_reduce(pht, op=dist.ReduceOp.MAX)
        dist.all_reduce(csum_t, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    total_records = int(tot_rec.item())
    limbs = [int(v) for v in csum_t.tolist()]
    records_checksum = (limbs[0] + (limbs[1] << 16) + (limbs[2] << 32) + (limbs[3] << 48)) & ((1 << 64) - 1)
    sec_per_step = elapsed / max(args.steps, 1)

    # soak leg (outside the timed region, before any CPU-side leg): the same step repeated for ~8 s.  The step count derives from the
    # rank-max step time, so every rank of an N > 1 run takes the same number of collective searches.
    soak_s = args.soak_seconds if args.soak_seconds >= 0 else (0.0 if args.no_cpu_baseline else 8.0)
    soak = None
    if soak_s > 0 and sec_per_step > 0:
        n_soak = int(max(1, min(2000, round(soak_s / sec_per_step))))
        per = []
        fence()
        smi_proc = None
        if args.smi_trace and rank == 0:
            import subprocess as _sp
            pr = torch.cuda.get_device_properties(dev)
            try:
                bdf = "%04x:%02x:%02x.0" % (int(pr.pci_domain_id), int(pr.pci_bus_id), int(pr.pci_device_id))
            except AttributeError:
                bdf = None
            cmd = [sys.executable, os.path.join(ROOT, "tools", "smi_trace.py"), args.smi_trace, str(soak_s + 30.0)] + (["--bdf", bdf] if bdf else [])
            smi_proc = _sp.Popen(cmd, stdout=_sp.PIPE, stderr=_sp.PIPE, text=True)
            time.sleep(0.3)
        for _ in range(n_soak):
            ts = time.perf_counter()
            step()
            ms.synchronize()
            per.append(time.perf_counter() - ts)
        fence()
        per_ms = np.array(per) * 1e3
        soak = {"steps": n_soak, "mean_ms_per_step": round(float(per_ms.mean()), 3), "median_ms_per_step": round(float(np.median(per_ms)), 3),
                "min_ms": round(float(per_ms.min()), 3), "max_ms": round(float(per_ms.max()), 3),
                "what": "the timed step repeated back to back after the timed region (this rank's wall time per step, no barrier in between)"}
        if smi_proc is not None:
            smi_proc.terminate()
            try:
                so, se = smi_proc.communicate(timeout=40)
            except Exception:   # noqa: BLE001
                smi_proc.kill(); so, se = "", "smi_trace did not stop"
            try:
                soak["smi"] = json.loads([l for l in so.splitlines() if l.startswith("{")][-1])
            except Exception:   # noqa: BLE001
                soak["smi"] = None
            soak["smi_ok"] = smi_proc.returncode == 0
            soak["smi_file"] = args.smi_trace
            if smi_proc.returncode != 0:
                print("bench: " + (se.strip().splitlines()[-1] if se.strip() else "smi_trace failed"), file=sys.stderr, flush=True)

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
        traffic, traffic_src = pmc_traffic(dom + "_kernel", args.config, n_total, L, world)
        # the name rocprofv3 prints for the dominant kernel (profiles/rNN_rocprofv3_kernel_stats.csv): weight-1 strands (every strand
        # without -f) run minhash_w1_kernel, weighted ones minhash_kernel<4,true,true>; the PMC summary files both under "minhash_kernel"
        rocprof_name = {"minhash": "minhash_kernel<4,true,true> (+ minhash_w1_kernel)" if cfg.get("filter") else "minhash_w1_kernel",
                        "overlap": "overlap_join_kernel", "index_build": "index_tile_kernel + index_bins_kernel"}.get(dom, dom + "_kernel")
        # `bound` names what really bounds the dominant kernel: the sketch kernels are integer-VALU work, the index and
        # second-stage kernels memory-side.  achieved / peak / unit / frac stay the SURVEY §8(d) recipe (algorithmic HBM bytes per launch
        # over the launch time against 8 TB/s) whatever the bound, and are repeated under `hbm`; the VALU fraction is filled in below
        real_bound = "valu" if dom in ("hash_kmers", "kmer_weight", "minhash", "ordered", "candidate") else "hbm"
        roofline = {"bound": real_bound, "kernel": rocprof_name, "pmc_summary_key": dom + "_kernel", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                    "hbm": {"achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6)},
                    "alg_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(avg_launch_s * 1e3, 4), "launches": int(launches),
                    "note": "achieved / peak / frac: the SURVEY §8(d) recipe (algorithmic HBM bytes / launch time vs 8 TB/s), tiny by construction "
                            "for an integer-VALU kernel; `bound` names the real bound and `valu` (same object, and the line's `valu`) its fraction"}
        # measured HBM rate of every kernel (PMC bytes per launch x launches / its summed time): which ones are memory-side
        hbm_by_kernel = {}
        for kname in mhap_amd.KERNEL_NAMES:
            tb, _ = pmc_traffic(("overlap_join" if kname == "overlap" else kname) + "_kernel", args.config, n_total, L, world)
            if tb and kt[kname]["ms"] > 0:
                hbm_by_kernel[kname] = {"GB_per_step": round(tb * kt[kname]["launches"] / K / 1e9, 2),
                                        "GB_per_s": round(tb * kt[kname]["launches"] / (kt[kname]["ms"] / 1e3) / 1e9, 1)}
        # integer-VALU view of the MinHash kernel: xorshift steps actually taken (weight x H per distinct k-mer; ~3x under -f)
        wfac = 3.0 if (cfg.get("filter")) else 1.0
        steps_per_read = 2 * (L - k + 1) * H * wfac
        mh_s = kernel_ms_per_step["minhash"] / 1e3
        xs_rate = steps_per_read * n_local / mh_s if mh_s > 0 else 0.0
        ceil_spec = 32 * VALU_SPEC / BITSLICED_OPS_PER_32_STEPS
        ceil_meas = 32 * VALU_MEASURED / BITSLICED_OPS_PER_32_STEPS
        valu = {"bound": "valu", "kernel": "minhash_kernel<4,true,true> (+ minhash_w1_kernel)" if cfg.get("filter") else "minhash_w1_kernel", "xorshift_steps_per_s": round(xs_rate, 1),
                "ceiling_spec_steps_per_s": round(ceil_spec, 1), "frac_of_spec_ceiling": round(xs_rate / ceil_spec, 4),
                "ceiling_measured_clock_steps_per_s": round(ceil_meas, 1), "frac_of_measured_ceiling": round(xs_rate / ceil_meas, 4),
                "ceiling_note": "bit-sliced rows (32 chains per lane as 64 bit-planes): one step of 32 chains is 92 vector ops "
                                "(28 v_xor_b32 + 64 v_xor_b32/v_bitop3_b32, 61 of the 92 three-input; 107 until the late-round-5 step) + 10 v_bitop3_b32 of depth filter + 3 of compare/loop = 105.  Spec "
                                "ceiling = 32 x (256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 7.86e13 lane-ops/s) / 105; the measured one uses the "
                                "6.5e13 lane-ops/s an all-VALU probe of TWO-input ops sustains (tools/valu_ops.hip) — under the power cap the three-input mix of this step holds a lower clock "
                                "(1.97 GHz against 2.2 with the 107-op step: MHAP_MINHASH_PROF), so the fraction fell while the kernel got faster.  Per-chain formulation: 4.79e12 steps/s",
                "vs_per_chain_ceiling": round(xs_rate / XORSHIFT_CEILING_PER_CHAIN, 4),
                "weight_factor_assumed": wfac}
        if dom == "minhash":
            roofline["valu"] = {"achieved": round(xs_rate, 1), "peak": round(ceil_spec, 1), "unit": "xorshift steps/s", "frac": round(xs_rate / ceil_spec, 4),
                                "frac_of_measured_clock_ceiling": round(xs_rate / ceil_meas, 4)}

        # second stage: the stage whose honest bound IS HBM/L2 bandwidth — 8 S' bytes of the stored row per candidate
        # pair + 8 S' of the query row once per query; the time is the `overlap` slot (join kernel + the per-lane kernel of the
        # pairs the join hands over).  The candidate stage beside it: random lookups into the inverted index, 2 lines per lookup.
        Sp = min(S, L - k2 + 1)
        ov_s, iq_s = kernel_ms_per_step["overlap"] / 1e3, kernel_ms_per_step["index_query"] / 1e3
        alg2 = (int(st["candidates_compared"]) + n_total) * 8 * Sp
        tr2, tr2_src = pmc_traffic("overlap_join_kernel", args.config, n_total, L, world)
        ach2 = alg2 / ov_s / 1e9 if ov_s > 0 else 0.0
        roofline_stage2 = {"bound": "hbm", "kernel": "overlap_join_kernel (+ overlap_kernel for the pairs it hands over)",
                           "achieved": round(ach2, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach2 / HBM_PEAK_GBS, 4),
                           "traffic": tr2, "traffic_source": tr2_src, "alg_bytes_per_step": int(alg2), "ms_per_step": round(ov_s * 1e3, 3),
                           "pairs_per_s": round(int(st["candidates_compared"]) / ov_s, 1) if ov_s > 0 else None,
                           "candidate_stage": {"kernel": "index_query_kernel (+ index_query_dense_kernel)", "ms_per_step": round(iq_s * 1e3, 3),
                                               "lookups_per_s": round(n_total * H / iq_s, 1) if iq_s > 0 else None,
                                               "postings_per_s": round(int(st["table_elements"]) / iq_s, 1) if iq_s > 0 else None,
                                               "index_build_ms_per_step": round(kernel_ms_per_step["index_build"], 3)},
                           "note": "algorithmic bytes = (candidate pairs + queries) x 8 S' (SURVEY §8(d): the stored row per pair, the query "
                                   "row once per query) over the summed time of the second-stage kernels"}

        strands = 2 * n_total
        out = {
            "metric": "overlaps/sec", "value": round(total_records / sec_per_step, 2), "unit": "overlaps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sec_per_step * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64", "data": "synthetic" if args.config != "c3" else "real",
            "config": {"workload": f"{cfg_label}, k={k}, --num-hashes {H}, ordered sketch k2={k2} S={S}, self-overlap"
                                   + (", -f k-mer filter, --filter-threshold 1e-5" if cfg.get("filter") else ""),
                       "name": args.config, "error_rate": args.error_rate,
                       "parallelism": f"reads round-robin over {world} GPU(s); per-rank index of own reads; RCCL all-gather of the forward query sketches inside libmhaphip (mhap_dist_*)" if world > 1 else "1 GPU"},
            "records_per_step": total_records,
            "sketches_per_sec": round(strands / float(pht[0].item()), 1) if float(pht[0].item()) > 0 else None,
            "sketches_per_sec_note": "2N strands / wall time of the sketch phase (packed reads in HBM -> MinHash + ordered tables in HBM, host "
                                     "bookkeeping of the add included; rank max), SURVEY §8(d)",
            "sketches_per_sec_kernels_only": round(strands / (sketch_ms / 1e3), 1) if sketch_ms > 0 else None,
            "overlaps_per_sec_search_phase": round(total_records / float(pht[2].item()), 1) if float(pht[2].item()) > 0 else None,
            "phase_wall_ms": {"sketch": round(float(pht[0].item()) * 1e3, 3), "exchange": round(float(pht[1].item()) * 1e3, 3),
                              "search": round(float(pht[2].item()) * 1e3, 3)},
            "kernel_ms_per_step": {kk: round(v, 3) for kk, v in kernel_ms_per_step.items()},
            "candidates_per_step": int(st["candidates_compared"]),
            "index_elements_per_step": int(st["table_elements"]),
            "overlap_slow_pairs_per_step": int(st["slow_pairs"]),
            "records_sha256_sorted_lines": sha, "records_checksum": "%016x" % records_checksum,
            "records_checksum_kind": "binary fields (output too large for text lines in Python)" if big_output else "sha256 of text lines, summed",
            "hbm_traffic_by_kernel": hbm_by_kernel,
            "kernel_ceilings": kernel_ceilings(args.config, n_total, L, world, kernel_ms_per_step, hbm_by_kernel),
            "roofline": roofline, "valu": valu, "roofline_stage2": roofline_stage2,
            "eager_exchange": eager_note,
            "ranks": rank_views,
            "soak": soak,
            "input_gen_s": round(t_gen, 2),
            "staging_ms_untimed": round(t_stage * 1e3, 1),
            "value_incl_host_pack_and_pcie": round(total_records / (sec_per_step + t_stage), 2),
            "c3": "C3 skipped: MHAP_C3_FASTA is not set" if (args.config != "c3" and W.c3_fasta_path() is None) else
                  ("this run" if args.config == "c3" else "available: run --config c3"),
        }
        if not args.no_cpu_baseline and world == 1 and not force_dist:
            # the N > 1 code path (collective add + sharded search through a ONE-rank RCCL communicator) timed on the same staged reads,
            # after the timed region: the N = 1 point of a scaling curve takes the plain path above — this says what the other path
            # costs on one GPU, so that the two are never confused (round 4: 7.1 vs 5.1 ms of ordered kernel between them)
            out["dist_path_1rank"] = dist_path_leg(ms, args, recs, mhap_amd.KERNEL_NAMES)
        if not args.no_cpu_baseline and world == 1:
            out.update(host_legs(args, cfg, p, flt, filter_path, fa_bench, L, H, S, k, k2, total_records, sha, tmpdir))
        emit(out)
    ms.close()
    shutil.rmtree(tmpdir, ignore_errors=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def emit(obj):
    """The JSON line is the LAST line of stdout: whatever native libraries have printed through C stdio (RCCL's banner at the first
    communicator) sits in a buffer that would otherwise be flushed at exit, after the line the driver parses."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:   # noqa: BLE001
        pass
    sys.stdout.flush()
    print(json.dumps(obj), flush=True)


def gather_rank_views(ms, dist, rank, world, local_rank):
    """Every rank's own view of the job, gathered to all ranks: what RCCL says (ncclCommCount, ncclCommUserRank, the communicator's
    device), the device's PCI bus id, and what the launcher said (RANK / LOCAL_RANK).  `consistent` = N ranks, each counting N, user
    ranks 0..N-1, N distinct PCI ids."""
    mine = dict(ms.dist_info(), rank=rank, local_rank=local_rank, pid=os.getpid(),
                device_name=torch.cuda.get_device_name(local_rank) if torch.cuda.is_available() else None)
    views = [mine]
    if dist is not None and world > 1:
        views = [None] * world
        dist.all_gather_object(views, mine)
    ok = (len(views) == world and all(v["comm_count"] == world for v in views) and sorted(v["comm_user_rank"] for v in views) == list(range(world))
          and len({v["pci_bus_id"] for v in views}) == world and all(v["comm_device"] == v["handle_device"] for v in views))
    return {"world": world, "consistent": bool(ok), "per_rank": views}


def dry_collective(n):
    """`bench.py --gpus N --dry-collective`: the communicator and one checked all-gather, nothing else (see --help)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n:
        raise SystemExit(f"--dry-collective --gpus {n}: launch with torch.distributed.run --nproc-per-node {n} (WORLD_SIZE is {world})")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    os.environ.setdefault("MHAP_DIST_TIMEOUT_S", "120")
    rep = {"dry_collective": True, "n_gpus": world}
    t0 = time.perf_counter()
    ms = MinHashSearch(MhapParams(num_hashes=64, ordered_sketch_size=64, device=local_rank))
    err = None
    try:
        ms.dist_init(rank, world, mdist.broadcast_unique_id(dist, rank, MinHashSearch.dist_unique_id))
        t_init = time.perf_counter() - t0
        gather_ms = [ms.dist_selftest(1 << 20) for _ in range(3)]       # 1 MB per rank, three times (the first one warms the channels up)
        big_ms = ms.dist_selftest(64 << 20)                             # and 64 MB per rank once: a bandwidth figure
        views = gather_rank_views(ms, dist, rank, world, local_rank)
    except Exception as e:   # noqa: BLE001
        err = repr(e)
    peers_bad = []
    have = torch.cuda.device_count()
    for b in range(min(world, have)):
        if b != local_rank and not torch.cuda.can_device_access_peer(local_rank, b):
            peers_bad.append(b)
    mine = {"rank": rank, "error": err, "no_peer_access_to": peers_bad}
    if err is None:
        mine.update(init_s=round(t_init, 3), allgather_1MB_ms=[round(x, 3) for x in gather_ms], allgather_64MB_ms=round(big_ms, 3),
                    allgather_64MB_GBps_per_rank_received=round((world - 1) * 64 * 2**20 / (big_ms / 1e3) / 1e9, 2) if world > 1 and big_ms > 0 else None)
    alls = [mine]
    if dist is not None:
        alls = [None] * world
        try:
            dist.all_gather_object(alls, mine)
        except Exception as e:   # noqa: BLE001
            alls = [mine, {"error": "all_gather_object: " + repr(e)}]
    ok = all(a and a.get("error") is None and not a.get("no_peer_access_to") for a in alls) and (err is None and views["consistent"])
    if rank == 0:
        rep.update(ready=bool(ok), ranks=views if err is None else None, per_rank_results=alls)
        emit(rep)
    ms.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def exchange_only(args):
    """`bench.py --gpus N --exchange-only [--config c2]`: see --help.  One JSON line from rank 0."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--exchange-only --gpus {args.gpus}: launch with torch.distributed.run --nproc-per-node {args.gpus} (WORLD_SIZE is {world})")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    os.environ.setdefault("MHAP_DIST_TIMEOUT_S", "300")
    cfg = W.CONFIGS[args.config]
    H, S = args.hashes or cfg["hashes"], 1536
    fa = W.config_reads(args.config, shard=rank, nshards=world, reads=args.reads or None, length=args.length or None, error_rate=args.error_rate)
    n_local = len(fa)
    flt = None
    ms = MinHashSearch(W.params_for(args.config, device=local_rank, num_hashes=H), kmer_filter=flt)
    ms.stage(fa)
    ms.dist_init(rank, world, mdist.broadcast_unique_id(dist, rank, MinHashSearch.dist_unique_id))
    n_pad = n_local
    if dist is not None:
        t = torch.tensor([n_local], dtype=torch.int64, device=torch.device("cuda", local_rank))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n_pad = int(t.item())
    Hrow = ms.params.num_hashes
    ord_bytes, small_bytes = n_pad * S * 8, n_pad * (Hrow * 4 + 16 + 8)
    mine = {"rank": rank, "reads": n_local, "rows_padded": n_pad, "ordered_bytes_sent": ord_bytes, "small_bytes_sent": small_bytes, "error": None}
    try:
        ms.dist_selftest(1 << 20)                                        # (channels warm)
        # (a) the fabric by itself: the same bytes per rank, nothing else on the chip (the self-test takes at most 1 GiB per rank)
        alone_ord = [ms.dist_selftest(min(ord_bytes, 1 << 30)) for _ in range(3)]
        alone_small = [ms.dist_selftest(max(1, min(small_bytes, 1 << 30))) for _ in range(3)]
        # (b) the eager add: the gathers run under the add's kernels; then the search says what it still had to wait for
        ms.dist_set_eager(True)
        under = []
        for _ in range(max(1, args.steps) + 1):
            ms.clear()
            t0 = time.perf_counter()
            ms.add_staged(); ms.synchronize()
            t_add = time.perf_counter() - t0
            xt = ms.dist_exchange_timing()
            t1 = time.perf_counter()
            recs = ms.dist_find_matches()
            t_search = time.perf_counter() - t1
            tm = ms.dist_last_timing()
            under.append({"add_ms": round(t_add * 1e3, 3), "ordered_gather_ms": round(xt["ordered_gather_ms"], 3), "small_gather_ms": round(xt["small_gather_ms"], 3),
                          "search_ms": round(t_search * 1e3, 3), "exposed_ms": round(tm["gather_small_ms"] + tm["wait_ordered_ms"], 3),
                          "eager_searches": ms.dist_eager_searches(), "records": len(recs)})
        under = under[1:]                                                   # (the first step allocates)
        gb = lambda b, ms_: round((world - 1) * b / (ms_ / 1e3) / 1e9, 2) if world > 1 and ms_ > 0 else None   # noqa: E731
        mine.update(alone={"ordered_gather_ms": [round(x, 3) for x in alone_ord], "small_gather_ms": [round(x, 3) for x in alone_small],
                           "ordered_GBps_received": gb(min(ord_bytes, 1 << 30), min(alone_ord))},
                    under_the_add=under,
                    hidden_ms=round(min(u["ordered_gather_ms"] + u["small_gather_ms"] - u["exposed_ms"] for u in under), 3),
                    exposed_ms=round(min(u["exposed_ms"] for u in under), 3))
    except Exception as e:   # noqa: BLE001
        mine["error"] = repr(e)
    alls = [mine]
    if dist is not None:
        alls = [None] * world
        try:
            dist.all_gather_object(alls, mine)
        except Exception as e:   # noqa: BLE001
            alls = [mine, {"error": "all_gather_object: " + repr(e)}]
    ok = all(a and a.get("error") is None for a in alls)
    if rank == 0:
        emit({"exchange_only": True, "n_gpus": world, "config": args.config, "ok": bool(ok),
              "what": "alone = mhap_dist_selftest with this rank's real row bytes and nothing else on the chip; under_the_add = the same gathers "
                      "issued by the eager add while its kernels run (events on the exchange stream), exposed_ms = what the search still waited for "
                      "(phase_wall_ms.exchange of the timed line), hidden_ms = gather time minus that",
              "per_rank": alls})
    ms.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def dist_path_leg(ms, args, recs_plain, kernel_names):
    """world == 1 only: K steps through the N > 1 entry points (collective add with the eager exchange, mhap_dist_find_matches_self) on a
    communicator of one rank, on the reads already staged; same fences as the timed region."""
    try:
        ms.dist_init(0, 1, MinHashSearch.dist_unique_id())

        def dstep():
            ms.clear()
            ms.add_staged()
            return ms.dist_find_matches()
        dstep()
        ms.dist_set_eager(True)
        dstep()
        ms.reset_kernel_times()
        ms.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = max(args.steps, 1)
        for _ in range(K):
            r = dstep()
        ms.synchronize(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        kt = ms.kernel_times()
        same = len(r) == len(recs_plain) and sorted(mhap_amd.records_to_lines(r)) == sorted(mhap_amd.records_to_lines(recs_plain)) if len(r) < 4_000_000 else len(r) == len(recs_plain)
        view = ms.dist_info()
        ms.dist_finalize()
        return {"ms_per_step": round(dt * 1e3, 3), "records_equal_to_the_plain_path": bool(same), "eager_exchange": True,
                "kernel_ms_per_step": {kk: round(kt[kk]["ms"] / K, 3) for kk in kernel_names}, "rccl": view,
                "what": "the same step through mhap_dist_* (collective add, sharded search) on a one-rank RCCL communicator; `value` is the plain path's"}
    except Exception as e:   # noqa: BLE001
        return {"error": repr(e)}


def preflight(n):
    """What `bench.py --gpus n` under torch.distributed.run needs, checked without entering any collective."""
    import ctypes
    rep = {"gpus_requested": n, "checks": {}, "ready": True}

    def check(name, ok, detail):
        rep["checks"][name] = {"ok": bool(ok), "detail": detail}
        if not ok:
            rep["ready"] = False

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    check("devices", have >= n, f"{have} visible (HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')}, ROCR_VISIBLE_DEVICES={os.environ.get('ROCR_VISIBLE_DEVICES')})")
    check("torch_rccl_backend", torch.distributed.is_available() and torch.distributed.is_nccl_available(), "torch.distributed nccl (= RCCL) backend")
    try:
        lib = mhap_amd.load_library()
        buf = ctypes.create_string_buffer(128)
        rc = lib.mhap_dist_unique_id(buf, ctypes.c_size_t(128)) if have > 0 else -1
        check("library_rccl_entry_points", rc == 0, "mhap_dist_unique_id (dlopen of the process's librccl + ncclGetUniqueId): rc %d" % rc)
    except Exception as e:   # noqa: BLE001
        check("library_rccl_entry_points", False, repr(e))
    if have >= 2:
        bad = [(a, b) for a in range(min(n, have)) for b in range(min(n, have)) if a != b and not torch.cuda.can_device_access_peer(a, b)]
        check("peer_access", not bad, "every pair of the first %d devices" % min(n, have) if not bad else f"no peer access between {bad[:8]}")
    else:
        check("peer_access", n <= 1, "needs two devices to test")
    check("ipc_mode", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0", "HSA_ENABLE_IPC_MODE_LEGACY=%s (the host driver only supports dmabuf IPC: must be 0 for multi-process RCCL)"
          % os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
    if n > 1:
        rep["launch"] = (f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {n} --master-addr 127.0.0.1 --master-port 29500 "
                         f"bench.py --gpus {n} --steps K --warmup W")
    print(json.dumps(rep), flush=True)
    return 0 if rep["ready"] else 1


VALU_WAVE_INSTR_PER_S = 6.7e13 / 64.0   # full-rate VALU wave-instructions per second the chip sustains at ANY occupancy (tools/issue_probe.hip)
# bench.py kernel-time slot -> the kernels of the PMC summaries that run in it
SLOT_KERNELS = {"kmer_weight": ("kmer_weight_kernel",), "minhash": ("minhash_w1_kernel", "minhash_kernel_weighted", "minhash_kernel", "minhash_w1_finish_kernel"),
                "ordered": ("ordered_kernel",), "index_build": ("index_tile_kernel", "index_offsets_kernel", "index_bins_kernel", "index_group_kernel"),
                "index_query": ("index_query_kernel", "index_query_dense_kernel", "index_query_dense4_kernel"), "overlap": ("overlap_join_kernel", "overlap_kernel", "poshist_kernel")}


def kernel_ceilings(config, n_total, L, world, kernel_ms_per_step, hbm_by_kernel):
    """One ceiling statement per kernel slot of a C2 step: which pipe bounds it and how close it runs to that pipe.
    The counters come from the committed PMC summary of THIS source (profiles/rNN_pmc_pipes.json, tools/pmc_pipes.py; null when the
    sources changed since), the times are this run's.  valu = VALU wave-instructions per second against the 1.05e12/s the chip sustains
    on full-rate work (v_bitop3 / v_alignbit / multiplies cost more than one such unit, so a kernel can be VALU-bound below 1.0);
    lds = share of the kernel's cycles the CUs' LDS pipes are busy (and how much of that is bank conflicts); hbm = PMC bytes over time
    against 8 TB/s.  `bound` names the largest of the three."""
    if not (config == "c2" and n_total == 100000 and L == 10000 and world == 1):
        return None
    from mhap_amd import build as mbuild
    here = mbuild.source_digest()
    doc = None
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_pipes.json")), reverse=True):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            if d.get("source_digest") == here:
                doc = (name, d)
                break
        except Exception:   # noqa: BLE001
            continue
    if doc is None:
        return {"note": "no profiles/*_pmc_pipes.json carries this source's digest (run tools/profile_round.sh on the GPU box)"}
    out = {"source": "profiles/" + doc[0]}
    for slot, members in SLOT_KERNELS.items():
        ms = kernel_ms_per_step.get(slot, 0.0)
        ks = [doc[1]["kernels"][m] for m in members if m in doc[1]["kernels"]]
        if ms <= 0 or not ks:
            continue
        valu = sum(k["per_launch"].get("SQ_INSTS_VALU", 0.0) * k["launches"] for k in ks) / max(k["launches"] for k in ks[:1])
        salu = sum(k["per_launch"].get("SQ_INSTS_SALU", 0.0) * k["launches"] for k in ks) / max(k["launches"] for k in ks[:1])
        lds_busy = sum(k["per_launch"].get("SQ_LDS_IDX_ACTIVE", 0.0) * k["launches"] for k in ks)
        gui = sum(k["per_launch"].get("GRBM_GUI_ACTIVE", 0.0) * k["launches"] for k in ks)
        conf = sum(k["per_launch"].get("SQ_LDS_BANK_CONFLICT", 0.0) * k["launches"] for k in ks)
        e = {"ms": round(ms, 3), "valu_instr_per_step": float("%.4g" % valu), "salu_instr_per_step": float("%.4g" % salu),
             "valu_frac": round(valu / (ms / 1e3) / VALU_WAVE_INSTR_PER_S, 3),
             "lds_pipe_frac": round(lds_busy / (256.0 * gui / 8.0), 3) if gui else None,
             "lds_conflict_share": round(conf / lds_busy, 3) if lds_busy else None,
             "hbm_frac": round(hbm_by_kernel[slot]["GB_per_s"] / HBM_PEAK_GBS, 3) if slot in hbm_by_kernel else None,
             "vgpr": max(k["vgpr"] for k in ks), "scratch_bytes_per_lane": max(k["scratch_bytes_per_lane"] for k in ks)}
        cands = {"valu": e["valu_frac"], "lds": e["lds_pipe_frac"] or 0.0, "hbm": e["hbm_frac"] or 0.0}
        e["bound"] = max(cands, key=cands.get)
        e["frac_of_bound"] = cands[e["bound"]]
        out[slot] = e
    return out


def pmc_traffic(kernel, config, n_total, L, world):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (tools/pmc_summary.py).  The counters were
    collected on the default workload (c2: 100k x 10kb, 1 GPU) with the kernels of ONE build: the summary carries the digest of the
    library's sources (mhap_amd.build.source_digest) and the field stays null when the sources have changed since, or for any other
    shape — a stale byte count must not be divided by a live kernel time."""
    if not (config == "c2" and n_total == 100000 and L == 10000 and world == 1):
        return None, None
    from mhap_amd import build as mbuild
    here = mbuild.source_digest()
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")), reverse=True):
        try:
            doc = json.load(open(os.path.join(ROOT, "profiles", name)))
            if doc.get("source_digest") != here:
                continue
            d = doc["kernels"].get(kernel)
            if d:
                return int(d["hbm_bytes_per_launch"]), "profiles/" + name + " (" + d["fetch_rule"] + ")"
        except Exception:
            pass
    return None, None


def host_legs(args, cfg, p, flt, filter_path, fa_bench, L, H, S, k, k2, total_records, bench_sha, tmpdir):
    """N = 1 only: (a) GPU vs oracle on the same sample reads, (b) the oracle timed as the CPU baseline, (c) the native driver end to end."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    out = {}
    aff, quota, threads = host_cpus()
    n = args.cpu_sample_reads
    wfac = 3.0 if cfg.get("filter") else 1.0
    if n <= 0:
        # ~2*(L-k+1)*H*w xorshift steps per read at ~3.5e8 steps/s/thread; aim at ~15 s
        per_read_s = 2.0 * (L - k + 1) * H * wfac / 3.5e8
        n = int(max(200, min(20000, 15.0 * threads / per_read_s)))
    if args.config == "c3":
        sample = fa_bench.subset(np.arange(min(n, len(fa_bench))))
        sample_desc = f"first {len(sample)} reads of the C3 file"
    else:
        sample = mhap_amd.synth_reads(n, L, seed=cfg["seed"] + 1, error_rate=args.error_rate, repeats=cfg["repeats"])
        sample_desc = (f"{n} reads x {L} bp from the same generator at the same 30x coverage / error model (own genome, seed+1)"
                       + (", same -f filter" if flt is not None else ""))
    oflt = None
    if flt is not None:
        oflt = O.Filter(flt.hashes, flt.fractions, flt.filter_cutoff, flt.offset, flt.range, flt.no_tf)
    # (a) same-input parity: the GPU path on the sample ...
    with MinHashSearch(p, kmer_filter=flt) as ms2:
        ms2.add_data(sample)
        g_lines = mhap_amd.records_to_lines(ms2.find_matches())
    # ... (b) and the oracle on it, timed: the CPU baseline
    one = min(len(sample), max(8, int(1.2 * 3.5e8 / (2.0 * (L - k + 1) * H * wfac))))      # ~1 s single-thread micro-sample
    t1 = O.run_self(sample.subset(np.arange(one)), k=k, H=H, k2=k2, S=S, nthreads=1, flt=oflt, cap=1 << 22)
    t = time.perf_counter()
    res = O.run_self(sample, k=k, H=H, k2=k2, S=S, nthreads=threads, flt=oflt, cap=1 << 24)
    wall = time.perf_counter() - t
    o_lines = O.record_lines(res["records"])
    g_sha, o_sha = sorted_sha(g_lines), sorted_sha(o_lines)
    out["parity_check"] = {"reads": len(sample), "records": len(o_lines), "gpu_records": len(g_lines), "equal": g_sha == o_sha,
                           "sha256_sorted_lines": o_sha, "what": "GPU path (C ABI) and CPU oracle on the same sample reads, sorted record lines"}
    nrec = len(o_lines)
    busy = res["sketch_s"] + res["search_s"]
    sk1 = 2 * one / t1["sketch_s"] if t1["sketch_s"] > 0 else None
    skn = 2 * len(sample) / res["sketch_s"] if res["sketch_s"] > 0 else None
    out["cpu_baseline"] = {
        "value": round(nrec / busy, 2) if busy > 0 else None, "unit": "overlaps/s", "cores": threads, "kind": "port",
        "sample": f"{sample_desc}, same flags; {nrec} records in {busy:.2f} s ({res['sketch_s']:.2f} s sketch + {res['search_s']:.2f} s search)",
        "sketches_per_sec": round(skn, 1) if skn else None, "reads_per_sec": round(len(sample) / busy, 2) if busy > 0 else None,
        "wall_s": round(wall, 2),
        "host": {"os_cpu_count": os.cpu_count(), "sched_affinity": aff, "cgroup_cpu_max_cores": quota, "threads_used": threads},
        "single_thread_sketches_per_sec": round(sk1, 2) if sk1 else None,
        "parallel_efficiency": round(skn / (sk1 * threads), 3) if (sk1 and skn) else None,
        "note": "CPU stand-in: C++ restatement of MHAP's Java path (oracle/), std::thread; threads = cgroup CPU quota of this container "
                "(the machine reports more cores than the process may use).  Not Java: no JVM on this box" + jvm_probe()}
    # (c) end to end through the native driver: FASTA open -> last record flushed (parse, pack, H2D, sketch, search, text formatting)
    cli = os.path.join(ROOT, "mhap_amd", "lib", "mhap-hip")
    if os.path.exists(cli) and args.config != "c3":
        fasta = os.path.join(tmpdir, "reads.fasta")
        W.write_fasta(fa_bench, fasta)
        cmd = [cli, "-s", fasta, "--num-hashes", str(H)] + (["-f", filter_path, "--filter-threshold", "1e-5"] if filter_path else [])
        outp = os.path.join(tmpdir, "records.txt")
        best = None
        walls = []
        for _ in range(3):   # later runs: page cache and HIP context warm, as a resident service would be
            t = time.perf_counter()
            with open(outp, "w") as fh:
                r = subprocess.run(cmd, stdout=fh, stderr=subprocess.PIPE, text=True)
            w = time.perf_counter() - t
            walls.append(round(w, 3))
            best = w if best is None else min(best, w)
        # (the leg's wall time spread 0.31 -> 1.27 s between boxes and runs for a 0.095-s step.  Every run's wall is kept, and
        #  when the runs differ by more than a factor of two the driver's own phase marks — MHAP_HOST_PROF: mhap_create incl. the HIP runtime's
        #  start-up, FASTA scan, add — of one more run say which phase it was)
        timeline = None
        if max(walls) > 2.0 * min(walls) or min(walls) > 0.8:
            t = time.perf_counter()
            with open(outp + ".prof", "w") as fh:
                rp = subprocess.run(cmd, stdout=fh, stderr=subprocess.PIPE, text=True, env=dict(os.environ, MHAP_HOST_PROF="1"))
            timeline = {"wall_s": round(time.perf_counter() - t, 3),
                        "cli": [x.strip() for x in rp.stderr.split("\n") if x.startswith("[cli]")][:8],
                        "create_ms": None}
            marks = {}
            for x in rp.stderr.split("\n"):
                if x.startswith("[host] create"):
                    marks[x[7:35].strip()] = float(x.split()[-2])
            if "create begin" in marks and len(marks) > 1:
                timeline["create_ms"] = {k: round(v - marks["create begin"], 1) for k, v in marks.items() if k != "create begin"}
        nlines, e2e_sha = 0, None
        if r.returncode == 0:
            ls = [x for x in open(outp).read().split("\n") if x]
            nlines, e2e_sha = len(ls), sorted_sha(ls)
        out["end_to_end"] = {"wall_s": round(best, 3), "walls_s": walls, "slow_run_timeline": timeline, "records": nlines, "records_per_s": round(nlines / best, 1) if best else None,
                             "equal_to_bench_records": (e2e_sha == bench_sha) if bench_sha else None, "rc": r.returncode,
                             "what": "mhap-hip -s reads.fasta (process start, FASTA parse, 2-bit pack, H2D, sketch, search, record text to a file); best of 3, every run in walls_s"}
    return out


def jvm_probe():
    """If a JVM and MHAP_JAR are present, diff the golden fixture against the real jar (tests/golden/verify_against_jar.sh)."""
    jar = os.environ.get("MHAP_JAR")
    if not (shutil.which("java") and jar and os.path.exists(jar)):
        return "; jvm_diff: unavailable (no java / MHAP_JAR)"
    r = subprocess.run(["sh", os.path.join(ROOT, "tests", "golden", "verify_against_jar.sh")], capture_output=True, text=True)
    return "; jvm_diff: " + ("ok" if r.returncode == 0 else "mismatch") + " (" + (r.stdout.strip().split("\n")[-1] if r.stdout.strip() else r.stderr.strip()[-120:]) + ")"


if __name__ == "__main__":
    main()
