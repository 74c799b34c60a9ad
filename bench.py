#!/usr/bin/env python3
"""bench.py -- genome positions/sec of the min-unique-k search (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (N = 1): BASELINE.json configs[1] -- synthetic 100 Mbp single-record FASTA, uniform ACGT
(numpy default_rng(20260515)), search range 20:200, reference batch geometry (10 M positions per
segment + 199 bytes of lookahead, newmap/search.py:229-235).  One STEP = one pass of the hot path
over every position of the genome.  The index is built once on the host, uploaded once; the
sequence bytes are resident in HBM before the timed region starts.

N > 1: the genome and the index stay the same; every rank holds a replica of the index in its own
HBM and searches a contiguous 1/N slice of the positions (independent units, no data-path
collective), then the per-rank uint8 slices are gathered on rank 0 with one RCCL collective
(part of the timed step).  Total work is fixed -> "scaling": "strong".

Prints ONE JSON line on rank 0 (see README / DESIGN.md "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BATCH = 100_000_000          # positions per launch.  The reference's --kmer-batch-size default (10 M) exists "to control
#                              memory usage" on the host; with 288 GB of HBM a launch takes a whole 100 Mbp record.  The
#                              throughput at the reference's 10 M is measured in the same run (`reference_batch`).
REFERENCE_BATCH = 10_000_000
PMC_POSITIONS_PER_DISPATCH = 100_000_000     # launch size of the runs behind profiles/round1/pmc_*_summary.csv
# BASELINE.json configs: name -> (search range, description); configs[1] = c2 is the bench workload,
# the others are parity / capability cases that the same harness can run on request
CONFIGS = {
    "c2": ((20, 200), "configs[1]: synthetic {mbp:g} Mbp single-record FASTA (uniform ACGT, seed 20260515)"),
    "c3": ((24, 150), "configs[2]: synthetic {mbp:g} Mbp FASTA as 24 human-shaped records (uniform ACGT, seeds 20260516+i)"),
    "c5": ((20, 255), "configs[4]: synthetic {mbp:g} Mbp, 50 % tandem repeats (seed 20260517)"),
}
DEFAULT_MBP = {"c2": 100.0, "c3": 3088.3, "c5": 1000.0}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="BASELINE.json workload (default c2 = configs[1])")
    ap.add_argument("--mbp", type=float, default=None, help="genome size in Mbp (default: the config's own size)")
    ap.add_argument("--seed-length", default="auto",
                    help="device tables: auto (default: sized for throughput), auto-small (<= 17 GB, what the one-shot CLI uses), "
                         "file (the index's seed length, 12), or a seed length 0..16")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--index-builder", choices=["host", "device"], default="host",
                    help="suffix sort on the host cores (default) or on the GPU (same index file; not timed in `value`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-batch", action="store_true",
                    help="skip the extra passes at the reference's 10 M batch (profiling runs: one launch size per kernel)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--workdir", default=os.environ.get("NEWMAP_AMD_BENCH_DIR", "/tmp/newmap_amd_bench"))
    args = ap.parse_args()
    if args.mbp is None:
        args.mbp = DEFAULT_MBP[args.config]
    return args


def prepare_workload(args, rank, world, barrier):
    """rank 0 generates the FASTA and builds the index; everybody then opens the same file"""
    from newmap_amd import synth
    from newmap_amd._c_newmap_generate_index import generate_fm_index
    wd = Path(args.workdir) / f"{args.config}_{args.mbp:g}mbp_{args.index_builder}"
    fa, idx = wd / "genome.fa", wd / "genome.awfmi"
    t_gen = t_build = 0.0
    t0 = time.time()
    recs = synth.config_genome(args.config, None if args.mbp == DEFAULT_MBP[args.config] and args.config != "c2" else args.mbp)
    if rank == 0:
        log(f"[bench] generated {len(recs)} record(s), {sum(r.size for _, r in recs)} bases ({time.time() - t0:.1f}s)")
    if rank == 0:
        wd.mkdir(parents=True, exist_ok=True)
        if not (fa.exists() and idx.exists() and (wd / "ok").exists()):
            t0 = time.time()
            synth.write_fasta(fa, recs)
            t_gen = time.time() - t0
            t0 = time.time()
            generate_fm_index(str(fa), str(idx), 8, 12, device=0 if args.index_builder == "device" else None)
            t_build = time.time() - t0
            (wd / "ok").write_text("ok")
            log(f"[bench] wrote {fa} ({t_gen:.1f}s), built index ({t_build:.1f}s, {idx.stat().st_size / 1e6:.0f} MB)")
    barrier()
    return recs, fa, idx, t_build


def measured_traffic(kernel: str, config: str, positions_per_launch: float):
    """HBM-side read+write bytes per launch of the dominant kernel, from the PMC passes committed under
    profiles/ (rocprofv3 --pmc cannot run inside this process).  Reads: TCC_EA0_RDREQ x 128 B -- on
    gfx950 every read request of these kernels is a 128-byte one (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ
    within 3 %), which is the guide's "FETCH_SIZE reports half" correction stated exactly; writes:
    TCC_EA0_WRREQ x 64 B (WRITE_SIZE in KB for the older pair-kernel pass).
    The counts are per launch of the default bench run (configs[1], one launch of 100 M positions); other
    workloads report null."""
    files = {"k_min_unique_quad": "pmc_quad_kernel_summary.csv", "k_min_unique_pair": "pmc_pair_kernel_summary.csv"}
    if config != "c2" or kernel not in files:
        return None, None
    f = ROOT / "profiles" / "round1" / files[kernel]
    if not f.exists():
        return None, None
    vals = {}
    for line in f.read_text().splitlines()[1:]:
        k, v = line.rsplit(",", 1)
        vals[k] = float(v)
    if "TCC_EA0_RDREQ_sum" not in vals:
        return None, None
    scale = positions_per_launch / PMC_POSITIONS_PER_DISPATCH
    read_b = vals["TCC_EA0_RDREQ_sum"] * 128.0
    if "TCC_EA0_WRREQ_sum" in vals:
        write_b = vals["TCC_EA0_WRREQ_sum"] * 64.0
    else:
        write_b = vals.get("WRITE_SIZE", 0.0) * 1024.0  # rocprofv3 reports WRITE_SIZE in KB (exact for stores)
    return (read_b + write_b) * scale, str(f.relative_to(ROOT))


def verify_sample(ix, recs, rec_off, d_out, KMIN, KMAX, samples=20000):
    """No oracle fits a multi-Gbp genome: re-derive a sample of the outputs through the count seam --
    at the reported length the both-strand count is 1, one base shorter (if allowed) it is not."""
    rng = np.random.default_rng(7)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    checked = 0
    for (name, r), o in zip(recs[:3], rec_off[:-1]):
        m = min(r.size, 260_000_000)
        rec = r[:m].tobytes()
        out = d_out[int(o):int(o) + m].cpu().numpy()
        pos = rng.integers(0, max(m - KMAX, 1), samples)
        k = out[pos].astype(np.int64)
        ok = k > 0
        pos, k = pos[ok], k[ok]
        rc = rec.translate(comp)[::-1]
        tot = ix.count_from_sequence(rec, pos, k) + ix.count_from_sequence(rc, m - pos - k, k)
        assert (tot == 1).all(), f"{name}: reported length is not unique"
        longer = k > KMIN
        tot2 = ix.count_from_sequence(rec, pos[longer], k[longer] - 1) + \
            ix.count_from_sequence(rc, m - pos[longer] - (k[longer] - 1), k[longer] - 1)
        assert (tot2 > 1).all(), f"{name}: a shorter unique length exists"
        checked += int(pos.size)
    return {"sampled_positions": checked, "property": "count(k)==1 and count(k-1)>1 via nm_count_from_sequence"}


def cpu_baseline(args, genome: np.ndarray, gpu_out: np.ndarray, KMIN: int, KMAX: int):
    """The oracle's C/OpenMP port of the reference algorithm (forward-strand FM-index, restart per
    probe, forward + reverse-complement queries, 4^12 seed table) on a bounded sample."""
    from oracle import ref_driver as rd
    threads = rd.lib().or_num_threads()
    log(f"[bench] cpu baseline: building the oracle's forward-strand FM-index on {threads} threads ...")
    t0 = time.time()
    oracle = rd.OracleIndex([genome.tobytes()])
    log(f"[bench] cpu baseline: suffix array done ({time.time() - t0:.1f}s)")
    oracle.enable_fm(12)
    t_index = time.time() - t0
    log(f"[bench] cpu baseline: FM port + 4^12 seed table done ({t_index:.1f}s)")
    rec = genome
    calib = min(1_000_000, len(rec) // 4)
    t0 = time.time()
    rd.ref_binary_search_segment_c(oracle, rec[:calib + KMAX - 1].tobytes(), calib, KMIN, KMAX, fm=True)
    rate = calib / max(time.time() - t0, 1e-6)
    sample = int(min(max(rate * args.cpu_seconds, calib), len(rec) - KMAX, 10_000_000))
    t0 = time.time()
    got, _, stats = rd.ref_binary_search_segment_c(oracle, rec[:sample + KMAX - 1].tobytes(), sample, KMIN, KMAX, fm=True)
    dt = time.time() - t0
    same = bool(np.array_equal(got.astype(np.uint8), gpu_out[:sample]))
    log(f"[bench] cpu baseline: {sample} positions in {dt:.1f}s on {threads} threads "
        f"(oracle index {t_index:.1f}s); bit-exact vs GPU: {same}")
    if not same:
        raise SystemExit("GPU output differs from the CPU oracle on the baseline sample")
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": sample / dt, "unit": "positions/s", "cores": threads, "kind": "port",
            "host": f"{cpu_model}, {os.cpu_count()} logical CPUs visible, oracle built -O3 -march=x86-64-v3 (AVX2, POPCNT), OpenMP",
            "sample": f"first {sample} positions of the same workload, {KMIN}:{KMAX}, "
                      f"{stats['probes'] / sample:.1f} probes and {2 * stats['probe_len'] / sample:.0f} LF steps "
                      "per position (reference schedule), index build excluded"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch                                   # first: libnewmap_amd.so binds to torch's HIP runtime
    import torch.distributed as dist
    from newmap_amd.engine import Index

    # one rank per GPU.  NEWMAP_AMD_BENCH_REHEARSE=1 (rehearsal of the N > 1 code path on a box with fewer GPUs
    # than ranks): ranks share the devices and the collectives go over gloo -- RCCL refuses two ranks on one GPU
    rehearse = os.environ.get("NEWMAP_AMD_BENCH_REHEARSE") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    from newmap_amd import parallel
    (KMIN, KMAX), desc = CONFIGS[args.config]
    recs, fa, idx_path, t_build = prepare_workload(args, rank, world, barrier)
    lengths = [int(r.size) for _, r in recs]
    rec_off = np.concatenate(([0], np.cumsum(lengths))).astype(np.int64)
    n = int(rec_off[-1])
    t0 = time.time()
    ix = Index(idx_path, dev_index, args.seed_length if args.seed_length in ("auto", "auto-small", "file") else int(args.seed_length))
    t_open = time.time() - t0
    info = ix.info()
    if rank == 0:
        log(f"[bench] index open + upload + seed table: {t_open:.1f}s; {info}")

    # sequence bytes resident in HBM before timing (records laid end to end)
    d_seq = torch.empty(n, dtype=torch.uint8, device=dev)
    for (_, r), o in zip(recs, rec_off[:-1]):
        d_seq[int(o):int(o) + r.size].copy_(torch.from_numpy(r))
    d_out = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_status = torch.zeros(8, dtype=torch.int64, device=dev)
    # the same plan as newmap_amd.parallel: contiguous slice of the global position space per rank,
    # cut into reference-shaped units (<= batch positions + kmax-1 bytes of lookahead from the record)
    lo, hi = parallel.shard_bounds(n, world)[rank]
    n_units = max(1, -(-(hi - lo) // args.batch))
    unit = -(-(hi - lo) // n_units) if hi > lo else args.batch     # a rank's slice in equal launches of <= batch positions
    units = parallel.units_for_slice(lengths, lo, hi, unit, KMAX)
    segs = [(int(rec_off[u.record]) + u.start, u.seg_len, u.count) for u in units]
    per = -(-n // world)
    # Positions shard with no data-path collective (newmap_amd/parallel.py: every rank keeps / writes its own
    # slice), so the timed passes contain no communication -- as at N = 1, the results stay in the HBM of the
    # GPU that produced them.  ONE gather of the per-rank uint8 slices to rank 0 (RCCL over xGMI) follows the
    # timed region: the north star's "final gather", timed on its own (`final_gather_ms`) and used to verify
    # the other ranks' output on rank 0.
    stream = torch.cuda.current_stream().cuda_stream
    seq_ptr, out_ptr, st_ptr = d_seq.data_ptr(), d_out.data_ptr(), d_status.data_ptr()

    def step(work=None):
        for (p, seg_len, nk) in (segs if work is None else work):
            ix.min_unique_segment_dev(seq_ptr + p, seg_len, nk, KMIN, KMAX, True, 1, out_ptr + p, st_ptr, stream)

    def finish_steps():
        pass

    for _ in range(args.warmup):
        step()
    finish_steps()
    torch.cuda.synchronize()
    barrier()
    ix.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    finish_steps()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    n_launch, kern_ms, kern_max = ix.read_timing()
    ix.set_timing(False)
    if rank == 0:
        log(f"[bench] {args.steps} steps in {elapsed:.3f}s; {n_launch} search launches, "
            f"{kern_ms:.2f} ms in the kernel (max {kern_max:.3f} ms)")
    # the same passes cut into the reference's default batch (10 M positions per launch), for comparison
    ref_batch = None
    if args.batch > REFERENCE_BATCH and not args.no_reference_batch:
        ref_units = parallel.units_for_slice(lengths, lo, hi, REFERENCE_BATCH, KMAX)
        ref_segs = [(int(rec_off[u.record]) + u.start, u.seg_len, u.count) for u in ref_units]
        step(ref_segs)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(ref_segs)
        torch.cuda.synchronize()
        barrier()
        ref_elapsed = time.perf_counter() - t1
        if world > 1:
            tr = torch.tensor([ref_elapsed], dtype=torch.float64, device=torch.device("cpu") if rehearse else dev)
            dist.all_reduce(tr, op=dist.ReduceOp.MAX)
            ref_elapsed = float(tr.item())
        ref_batch = {"batch": REFERENCE_BATCH, "value": n * args.steps / ref_elapsed, "unit": "positions/s",
                     "ms_per_step": ref_elapsed / args.steps * 1e3, "launches_per_step": len(ref_segs)}
    final_gather_ms = None
    if world > 1:
        cdev = torch.device("cpu") if rehearse else dev    # gloo gathers CPU tensors only
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        pad = torch.zeros(per, dtype=torch.uint8, device=cdev)
        pad[:hi - lo].copy_(d_out[lo:hi])
        gathered = [torch.empty(per, dtype=torch.uint8, device=cdev) for _ in range(world)] if rank == 0 else None
        torch.cuda.synchronize()
        barrier()
        tg = time.perf_counter()
        dist.gather(pad, gathered, dst=0)
        torch.cuda.synchronize()
        final_gather_ms = (time.perf_counter() - tg) * 1e3
        if rank == 0:                                      # rank 0 now holds every rank's slice: verify on the whole output
            for r, (rlo, rhi) in enumerate(parallel.shard_bounds(n, world)):
                d_out[rlo:rhi].copy_(gathered[r][:rhi - rlo])
    status = d_status.cpu().numpy()
    if int(status[1]):
        raise SystemExit(f"k-mer not found in the index at position {int(status[2])}")

    # counter build of the kernel: LF steps / rank blocks / seed lookups per launch (untimed)
    ix.set_count_steps(True)
    tallies = np.zeros(8, dtype=np.int64)
    probe = {"lf_steps": 0, "rank_blocks": 0, "seed_lookups": 0, "settled": 0}
    for (p, seg_len, nk) in segs:
        ix.min_unique_segment_dev(seq_ptr + p, seg_len, nk, KMIN, KMAX, True, 1, out_ptr + p, st_ptr, stream)
        torch.cuda.synchronize()
        tallies += d_status.cpu().numpy()
        for k_, v_ in ix.probe_tally().items():
            probe[k_] += v_
    ix.set_count_steps(False)

    if rank == 0:
        my_pos = hi - lo
        steps_pp = tallies[3] / max(tallies[7], 1)
        # algorithmic bytes (DESIGN.md "Measurement"): per distinct rank structure an LF step reads (lo and
        # hi in one block count once) 16 B with LF blocks / 32 B with the packed rank blocks, 8 B per
        # table entry, 1 sequence byte and one output element per position
        rank_bytes = 16 if ix.info()["lf_blocks"] and not ix.info()["two_step_blocks"] else 32
        alg_bytes = tallies[4] * rank_bytes + tallies[5] * 8 + my_pos * (1 + 1)
        per_launch_bytes = alg_bytes / max(len(segs), 1)
        avg_launch_ms = kern_ms / max(n_launch, 1)
        achieved = per_launch_bytes / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        kernel_name = {1: "k_min_unique", 2: "k_min_unique_v2", 3: "k_min_unique_mp",
                       4: "k_min_unique_pair", 5: "k_min_unique_quad"}.get(ix.info()["last_range_kernel"], "?")
        traffic, traffic_src = measured_traffic(kernel_name, args.config, my_pos / max(len(segs), 1))
        result = {
            "metric": f"genome positions/sec (min-unique-k search, {KMIN}:{KMAX})",
            "value": n * args.steps / elapsed,
            "unit": "positions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": desc.format(mbp=n / 1e6) + f", search-range {KMIN}:{KMAX}, both strands",
                       "records": len(recs),
                       "positions": n, "batch": args.batch, "segments_per_rank": len(segs),
                       "seed_length": info["seed_length"], "pair_core_length": info["pair_core_length"],
                       "quad_core_length": info.get("quad_core_length", 0),
                       "index_bytes_hbm": info["device_bytes"],
                       "parallelism": f"positions sharded over {world} GPU(s), index replicated"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name,
                         "avg_launch_ms": avg_launch_ms, "launches": n_launch,
                         "algorithmic_bytes_per_launch": per_launch_bytes,
                         "lf_steps_per_position": float(steps_pp),
                         "rank_blocks_per_position": float(tallies[4] / max(tallies[7], 1)),
                         "table_words_per_position": float(tallies[5] / max(tallies[7], 1))},
            # k_repeat_probe runs before the range kernel (one lane per 64 positions); its work is not in `roofline`
            "repeat_probes": {"enabled": bool(info.get("repeat_probes", 0)),
                              "settled_fraction": probe["settled"] / max(my_pos, 1),
                              "lf_steps_per_position": probe["lf_steps"] / max(my_pos, 1),
                              "seed_lookups_per_position": probe["seed_lookups"] / max(my_pos, 1)},
            "host": {"index_build_s": t_build, "index_open_s": t_open},
        }
        if ref_batch is not None:
            result["reference_batch"] = ref_batch
        if final_gather_ms is not None:
            result["final_gather_ms"] = final_gather_ms     # one RCCL gather of all slices to rank 0, outside `value`
        if world == 1 and not args.no_cpu_baseline and args.config == "c2":      # the oracle's comparison-based
            # suffix sorter is built for the uniform benchmark genome, not for repeat-heavy inputs
            gpu_out = d_out[:lengths[0]].cpu().numpy()
            result["cpu_baseline"] = cpu_baseline(args, recs[0][1], gpu_out, KMIN, KMAX)
        if args.config != "c2" or world > 1:               # (at N > 1 d_out holds every rank's slice after the final gather)
            result["verify"] = verify_sample(ix, recs, rec_off, d_out, KMIN, KMAX)
        print(json.dumps(result), flush=True)
    barrier()
    ix.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
