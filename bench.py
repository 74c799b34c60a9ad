#!/usr/bin/env python3
"""bench.py -- genome positions/sec of the min-unique-k search (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher environment (WORLD_SIZE unset) the script starts its own N ranks -- a CHILD
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...`, before anything here imports torch or
touches HIP -- and exits with the child's code; launched by torch.distributed.run directly it is one rank per GPU.

The headline (`value`) is the configuration BASELINE.json's metric and north_star are quoted on, at EVERY N: the
synthetic ~3 Gbp genome (24 human-shaped records, 3.09 Gbp, uniform ACGT, seeds 20260516+i) searched at 20:200 on
a device-built both-strand index, the genome fixed as N grows ("scaling": "strong"): its positions are cut into
interleaved ~64 M chunks, chunk c owned by rank c mod N, no data-path collective, the index replicated in every GPU's
HBM.  One STEP = one pass of the hot path (k_sites with its encode stage, repeat probes, k_resolve) over every position
of the rank's work units, sequence bytes resident in HBM before the timed region; the segments of a pass are dealt
round-robin over --streams HIP streams (default 2).  After the timed region ONE RCCL gather (N > 1) collects the
per-rank uint8 results on rank 0 (`final_gather_ms`, outside `value`).

At N = 1 the same run also reports: `roofline` for the kernel with the largest total time (HIP events on the launch
stream; PMC traffic from the committed pass under profiles/), `cpu_baseline` (the oracle's C/OpenMP port of the
reference schedule on an FM-index of THIS genome, timed on a bounded sample and compared bit for bit with the GPU
output), `end_to_end` (the same genome through the native driver, FASTA in -> files out) and the nested block
`configs1` (BASELINE configs[1]: 100 Mbp single record, one launch per pass -- round 2's headline).

Rank 0 prints ONE JSON line on stdout; everything else goes to stderr.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import shutil
import statistics
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# HIP deals the streams of a process over GPU_MAX_HW_QUEUES hardware queues (default 4), round-robin by creation: a handle's
# own stream, the side streams of its lanes, torch's streams and the native driver's five share them, and two streams that land
# on one queue run their kernels one after the other.  Measured (profiles/round4/ab_hw_queues.json): the nested human-shaped
# block, whose two streams are the 20th-odd of the process, 29.4 G positions/s with 4 queues, 36.0 G with 8 (its stand-alone
# run: 37 G either way); the headline is unchanged.  Set before anything initialises HIP.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BATCH = 1 << 28                # positions per launch: one launch per record (the largest has 249 M) or per 64 M chunk.  The
#                                reference's --kmer-batch-size default (10 M) exists "to control memory usage" on the host; the
#                                throughput at the reference's 10 M is measured in the `configs1` block (`reference_batch`).
REFERENCE_BATCH = 10_000_000
CHUNK = 64 << 20               # interleaved chunk of the multi-GPU split (newmap_amd/parallel.py)
C2_SEED, C2_BASES = 20260515, 100_000_000
SW = 16                        # uint64 words of a segment's status row (include/newmap_amd.h NM_STATUS_WORDS)
PROFILES = ROOT / "profiles" / "round4"
# PMC passes committed under profiles/ (tools/profile_cfg.sh), by workload and by the kernel (family) the roofline block may name
PMC_SUMMARIES = {
    "ns": {"k_sites": [PROFILES / "pmc_ns_k_sites_summary.csv"]},
    "c2": {"k_sites": [PROFILES / "pmc_c2_k_sites_summary.csv"]},
    "c3": {"k_sites": [PROFILES / "pmc_c3_k_sites_summary.csv"]},
    "c5": {"k_sites": [PROFILES / "pmc_c5_k_sites_summary.csv"],
           "k_repeat_probe(_coarse)": [PROFILES / "pmc_c5_k_period_runs_summary.csv", PROFILES / "pmc_c5_k_repeat_probe_summary.csv"],
           "k_resolve": [PROFILES / "pmc_c5_k_resolve_summary.csv"]},
    "hs": {"k_sweep": [PROFILES / "pmc_hs_k_sweep_summary.csv"], "k_sites": [PROFILES / "pmc_hs_k_sites_summary.csv"]},
}
KERNEL_SOURCES = [ROOT / "newmap_amd" / "csrc" / f for f in ("nm_kernels.hip.h", "nm_core.h")]


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)         # 20 passes of ~17 ms over 3.09 Gbp: a timed region of ~0.35 s
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["ns", "c2", "c3", "c5", "hs"], default="ns",
                    help="headline workload: ns = the metric's own configuration (default: ~3 Gbp genome, 20:200); c2 / c3 / c5 = "
                         "BASELINE configs[1] / [2] / [4] as capability runs; hs = the human-shaped stand-in of configs[3]'s GRCh38 (repeat families, "
                         "soft-masking, N runs) at 20:200")
    ap.add_argument("--mbp", type=float, default=None, help="shrink the headline genome to this many Mbp (rehearsals, tests)")
    ap.add_argument("--seed-length", default="auto",
                    help="device tables: auto (default: sized for throughput), auto-small (<= 20 GB, what the one-shot CLI uses), "
                         "file (the index's seed length, 12), or a seed length 0..16")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--streams", type=int, default=None, choices=[1, 2, 3, 4, 5],
                    help="the segments of a pass are launched round-robin on this many HIP streams (the handle keeps one set of launch "
                         "scratch per stream, so neighbouring segments overlap); passes of one segment use one stream.  Default: 2, and 5 on "
                         "the human-shaped genome, whose launches end in the latency-bound tail of k_sweep (measured: 35.9 / 39.5 / 41.6 / 43.4 G "
                         "positions/s on 2 / 3 / 4 / 5 streams; the uniform headline 204.5 / 200.8 / 194.1 / 194.6 G)")
    ap.add_argument("--index-builder", choices=["host", "device"], default="device",
                    help="suffix sort on the GPU (default) or on the host cores (same index file; not timed in `value`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs1", action="store_true", help="skip the nested blocks, configs[1] and the human-shaped genome (profiling runs)")
    ap.add_argument("--no-hs", action="store_true", help="skip the nested human-shaped block (profiling runs)")
    ap.add_argument("--hs-mbp", type=float, default=None, help="shrink the human-shaped genome of the nested block (rehearsals)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the FASTA-in -> files-out leg (kernel traces)")
    ap.add_argument("--no-spread", action="store_true", help="skip the separately synchronised passes behind `pass_ms` (kernel traces)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the baseline sample")
    ap.add_argument("--workdir", default=os.environ.get("NEWMAP_AMD_BENCH_DIR", "/tmp/newmap_amd_bench"))
    args = ap.parse_args(argv)
    args.streams_given = args.streams is not None
    if args.streams is None:
        args.streams = 5 if args.config == "hs" else 2
    return args


def self_launch(args) -> int:
    """--gpus N > 1 from a bare shell: N child ranks under torch.distributed.run.  Nothing in THIS process has imported
    torch or touched HIP (a process that has initialised the GPU must not exec or be replaced); it only waits."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # dmabuf IPC (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    log(f"[bench] starting {args.gpus} ranks: {' '.join(cmd)}")
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------ workloads
class Workload:
    """records = [(name, seed, length, kind)], generated on demand (kind: 'uniform' | 'tandem')"""

    def __init__(self, key, desc, krange, records):
        self.key, self.desc, self.krange, self.records = key, desc, krange, records
        self.lengths = [r[2] for r in records]
        self.total = int(sum(self.lengths))
        self._cache = {}

    def record(self, i) -> np.ndarray:
        from newmap_amd import synth
        if i not in self._cache:
            _, seed, n, kind = self.records[i]
            self._cache[i] = synth.uniform_dna(n, seed) if kind == "uniform" else (synth.human_like_dna(n, seed) if kind == "human" else synth.tandem_dna(n, seed))
        return self._cache[i]

    def drop(self, i=None):
        if i is None:
            self._cache.clear()
        else:
            self._cache.pop(i, None)


def human_shaped(mbp, first_seed):
    from newmap_amd import synth
    total = sum(synth.HUMAN_SHAPED)
    f = 1.0 if mbp is None else mbp * 1e6 / total
    names = [f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY"]
    return [(nm, first_seed + i, max(1000, int(L * f)), "uniform") for i, (nm, L) in enumerate(zip(names, synth.HUMAN_SHAPED))]


def headline_workload(args) -> Workload:
    if args.config == "ns":
        recs = human_shaped(args.mbp, 20260516)
        w = Workload("ns", "", (20, 200), recs)
        w.key = f"ns_{w.total / 1e6:g}mbp"
        w.desc = (f"the metric's configuration: synthetic ~3 Gbp genome ({w.total / 1e6:g} Mbp, 24 human-shaped records, uniform ACGT, "
                  "seeds 20260516+i -- the genome of BASELINE configs[2])")
        return w
    if args.config == "c2":
        return configs1_workload(int((args.mbp or 100) * 1e6))
    if args.config == "c3":
        recs = human_shaped(args.mbp, 20260516)
        w = Workload("c3", "", (24, 150), recs)
        w.key = f"c3_{w.total / 1e6:g}mbp"
        w.desc = f"configs[2]: synthetic {w.total / 1e6:g} Mbp FASTA as 24 human-shaped records (uniform ACGT, seeds 20260516+i)"
        return w
    if args.config == "hs":
        return hs_workload(args.mbp)
    n = int((args.mbp or 1000) * 1e6)
    return Workload(f"c5_{n / 1e6:g}mbp", f"configs[4]: synthetic {n / 1e6:g} Mbp, 50 % tandem repeats (seed 20260517)", (20, 255),
                    [("rep1", 20260517, n, "tandem")])


def hs_workload(mbp=None) -> Workload:
    recs = [(nm, 20260600 + i, max(100_000, L), "human") for i, (nm, _, L, _) in enumerate(human_shaped(mbp, 0))]
    w = Workload("hs", "", (20, 200), recs)
    w.key = f"hs_{w.total / 1e6:g}mbp"
    w.desc = (f"human-shaped stand-in of configs[3]'s genome: {w.total / 1e6:g} Mbp in 24 records of synth.human_like_dna (25 % interspersed repeat "
              "families at 2-20 % divergence on both strands, segmental duplications, half of the bases soft-masked, telomere / centromere / gap runs of N)")
    return w


def configs1_workload(n=C2_BASES) -> Workload:
    return Workload(f"c2_{n / 1e6:g}mbp", f"configs[1]: synthetic {n / 1e6:g} Mbp single-record FASTA (uniform ACGT, seed 20260515)",
                    (20, 200), [("chr1", C2_SEED, n, "uniform")])


def prepare_index(args, wl: Workload, rank, barrier):
    """rank 0 writes the FASTA and builds the index (once per workdir); everybody then opens the same file"""
    from newmap_amd import synth
    from newmap_amd._c_newmap_generate_index import generate_fm_index
    wd = Path(args.workdir) / f"{wl.key}_{args.index_builder}"
    fa, idx = wd / "genome.fa", wd / "genome.awfmi"
    times = {"fasta_write_s": 0.0, "index_build_s": 0.0}
    if rank == 0 and not (fa.exists() and idx.exists() and (wd / "ok").exists()):
        wd.mkdir(parents=True, exist_ok=True)
        t0 = time.time()
        with open(fa, "wb") as fh:
            for i, (name, _, _, _) in enumerate(wl.records):
                tmp = wd / "rec.fa"
                synth.write_fasta(tmp, [(name, wl.record(i))])
                with open(tmp, "rb") as src:
                    shutil.copyfileobj(src, fh, 1 << 24)
                tmp.unlink()
        times["fasta_write_s"] = time.time() - t0
        t0 = time.time()
        generate_fm_index(str(fa), str(idx), 8, 12, device=0 if args.index_builder == "device" else None)
        times["index_build_s"] = time.time() - t0
        (wd / "ok").write_text("ok")
        log(f"[bench] {wl.key}: wrote {fa} ({times['fasta_write_s']:.1f}s), built index ({times['index_build_s']:.1f}s, "
            f"{idx.stat().st_size / 1e6:.0f} MB, {args.index_builder} builder)")
    barrier()
    return fa, idx, times


# ------------------------------------------------------------------------------------------ measurement
def source_hash() -> str:
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def measured_traffic(kernel: str, quad_m: int, positions_per_launch: float, summaries):
    """HBM-side read + write bytes per launch of the named kernel (family) from the PMC passes committed under profiles/
    (rocprofv3 --pmc cannot run inside this process).  A summary names the kernel sources it was measured on (sha256 of
    nm_kernels.hip.h + nm_core.h), the core length of the table the sites read and the launch size; a summary of OTHER
    sources, another table or another launch size reports null with the reason.  Reads: TCC_EA0_RDREQ x 128 B (on gfx950
    every read request of these gathers is a 128-byte one, TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ -- the guide's "FETCH_SIZE
    reports half" correction stated exactly; requests the counters report as 64- or 32-byte ones are priced as such); writes:
    TCC_EA0_WRREQ x 64 B.  A family of kernels (the probes) is the sum of its members' summaries."""
    if not summaries:
        return None, "no PMC pass for this kernel / workload"
    total, used = 0.0, []
    for summary in summaries:
        if not summary.exists():
            return None, f"{summary.relative_to(ROOT)} not collected"
        meta, vals = {}, {}
        for line in summary.read_text().splitlines():
            if line.startswith("#"):
                for kv in line[1:].split(","):
                    if "=" in kv:
                        k, v = kv.strip().split("=", 1)
                        meta[k] = v
            elif "," in line and not line.startswith("counter"):
                k, v = line.rsplit(",", 1)
                vals[k] = float(v)
        if meta.get("source_sha256") != source_hash():
            return None, f"{summary.name} was measured on other kernel sources ({meta.get('source_sha256')} != {source_hash()}): re-run the PMC pass"
        if kernel == "k_sites" and int(meta.get("site_core_length", -1)) != quad_m:
            return None, f"{summary.name} was measured with another table"
        if abs(float(meta.get("positions_per_launch", 0)) - positions_per_launch) > 0.01 * positions_per_launch:
            return None, f"{summary.name} was measured with another launch size"
        if "TCC_EA0_RDREQ_sum" not in vals:
            return None, f"{summary.name} lacks TCC_EA0_RDREQ_sum"
        rd = vals["TCC_EA0_RDREQ_sum"]
        total += rd * 128.0 + vals.get("TCC_EA0_WRREQ_sum", 0.0) * 64.0
        used.append(str(summary.relative_to(ROOT)))
    return total, " + ".join(used)


KINDS = {0: "search", 1: "segment", 2: "k_repeat_probe_coarse", 3: "k_repeat_probe", 4: "finish (k_open_words + k_sweep + k_resolve)", 5: "k_sweep"}


class Run:
    """one workload on this rank: its units resident in HBM, timed passes, counter pass"""

    def __init__(self, args, wl: Workload, idx_path, units, rank, world, dev, barrier, dist, rehearse, seed_length):
        import torch
        from newmap_amd.engine import Index
        self.torch, self.args, self.wl, self.units, self.rank, self.world = torch, args, wl, units, rank, world
        self.dev, self.barrier, self.dist, self.rehearse = dev, barrier, dist, rehearse
        self.KMIN, self.KMAX = wl.krange
        t0 = time.time()
        self.ix = Index(idx_path, dev.index, seed_length if seed_length in ("auto", "auto-small", "file") else int(seed_length))
        self.t_open = time.time() - t0
        self.info = self.ix.info()
        if rank == 0:
            log(f"[bench] {wl.key}: index open + upload + tables: {self.t_open:.1f}s; {self.info}")
        # the segments of my units, laid end to end in one device buffer (+ their outputs, + one status row per unit)
        seg_off = np.concatenate(([0], np.cumsum([(u.seg_len + 15) // 16 * 16 for u in units]))).astype(np.int64)
        out_off = np.concatenate(([0], np.cumsum([(u.count + 15) // 16 * 16 for u in units]))).astype(np.int64)
        self.seg_off, self.out_off = seg_off, out_off
        self.my_positions = int(sum(u.count for u in units))
        self.d_seq = torch.empty(max(int(seg_off[-1]), 16), dtype=torch.uint8, device=dev)
        self.d_out = torch.zeros(max(int(out_off[-1]), 16), dtype=torch.uint8, device=dev)
        self.d_status = torch.zeros((max(len(units), 1), SW), dtype=torch.int64, device=dev)
        t0 = time.time()
        for u, o in zip(units, seg_off[:-1]):
            rec = wl.record(u.record)
            self.d_seq[int(o):int(o) + u.seg_len].copy_(torch.from_numpy(rec[u.start:u.start + u.seg_len]))
        torch.cuda.synchronize()
        self.t_upload = time.time() - t0
        self.stream = torch.cuda.current_stream().cuda_stream
        self._extra_streams = [torch.cuda.Stream(device=dev) for _ in range(max(args.streams, 1) - 1)]
        self.streams = [self.stream] + [s_.cuda_stream for s_ in self._extra_streams]
        self.segs = [(int(so), u.seg_len, u.count, int(oo), i) for i, (u, so, oo) in enumerate(zip(units, seg_off[:-1], out_off[:-1]))]
        self.overlap_identical = None          # set by timed(): output of the overlapped passes == output of a one-stream pass

    def step(self, segs=None, streams=None):
        """one pass: every segment once.  Segments are independent (own input, output and status row); with several
        streams they are dealt round-robin, a stream keeps the order of its own segments"""
        sp, op, st = self.d_seq.data_ptr(), self.d_out.data_ptr(), self.d_status.data_ptr()
        segs = self.segs if segs is None else segs
        streams = self.streams if streams is None else streams
        if len(segs) < 2:
            streams = streams[:1]
        for j, (so, seg_len, cnt, oo, i) in enumerate(segs):
            self.ix.min_unique_segment_dev(sp + so, seg_len, cnt, self.KMIN, self.KMAX, True, 1, op + oo, st + 8 * SW * i, streams[j % len(streams)])

    def _reduce(self, x, op):
        if self.world == 1 and not self.dist.is_initialized():
            return x
        t = self.torch.tensor(x, dtype=self.torch.float64, device=self.torch.device("cpu") if self.rehearse else self.dev)
        self.dist.all_reduce(t, op=op)
        return t.cpu().tolist()

    def max_over_ranks(self, x: float) -> float:
        return float(self._reduce([x], self.dist.ReduceOp.MAX)[0])

    def sum_over_ranks(self, x: float) -> float:
        return float(self._reduce([x], self.dist.ReduceOp.SUM)[0])

    def timed(self, steps, warmup, segs=None, kernel_events=True):
        """`steps` passes between barrier + synchronize on both sides.  Kernel times (HIP events on the launch stream,
        read_timing kinds 0 .. 4) are taken in the timed passes themselves when a pass runs on ONE stream (one launch per
        pass); passes whose segments overlap on several streams get one extra pass on one stream for them afterwards
        (outside `elapsed`), since events around overlapping kernels time each other's work."""
        torch = self.torch
        n_segs = len(self.segs if segs is None else segs)
        overlapped = len(self.streams) > 1 and n_segs > 1
        for _ in range(warmup):
            self.step(segs)
        torch.cuda.synchronize()
        self.barrier()
        if kernel_events and not overlapped:
            self.ix.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(segs)
        torch.cuda.synchronize()
        self.barrier()
        elapsed = time.perf_counter() - t0
        kinds = {k: (0, 0.0, 0.0) for k in KINDS}
        if kernel_events:
            if overlapped:
                snap = self.d_out.clone()                         # what the overlapped passes left
                self.d_out.zero_()
                self.ix.set_timing(True)
                self.step(segs, self.streams[:1])
                torch.cuda.synchronize()
                self.overlap_identical = bool(torch.equal(snap, self.d_out))
                del snap
                if not self.overlap_identical:
                    raise SystemExit("segments overlapped on several streams gave a different output than on one stream")
            kinds = {k: self.ix.read_timing(k) for k in KINDS}
            self.ix.set_timing(False)
        return self.max_over_ranks(elapsed), kinds

    def spread(self, steps, segs=None):
        """min / median / max of `steps` further passes, each bracketed by its own barrier + synchronize (slowest rank
        per pass).  Separate from `value`: a synchronised pass cannot overlap its tail with the next one's head."""
        torch = self.torch
        ts = []
        for _ in range(steps):
            torch.cuda.synchronize()
            self.barrier()
            t0 = time.perf_counter()
            self.step(segs)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = self._reduce(ts, self.dist.ReduceOp.MAX)
        return {"min": min(ts), "median": statistics.median(ts), "max": max(ts), "passes": len(ts),
                "note": "each pass synchronised on its own (slowest rank per pass); the timed region above runs its passes back to back"}

    def check_status(self):
        st = self.d_status.cpu().numpy()
        bad = np.flatnonzero(st[:, 1])
        if bad.size:
            where = f"record {self.wl.records[self.units[int(bad[0])].record][0]}" if int(bad[0]) < len(self.units) else f"status row {int(bad[0])}"
            raise SystemExit(f"k-mer not found in the index: {where}, segment position {int(st[bad[0], 2])}")

    def more_status_rows(self, n):
        """`n` further status rows (segments cut differently from the resident units get their own); returns the first"""
        first = self.d_status.shape[0]
        self.d_status = self.torch.cat([self.d_status, self.torch.zeros((n, SW), dtype=self.torch.int64, device=self.dev)])
        return first

    def counters(self):
        """counter build of the kernels, one pass: [3] LF steps, [4] rank blocks, [5] table words read by k_sites,
        [6] table words read by k_resolve (second chance + seed entries), [7] positions searched"""
        torch = self.torch
        self.ix.set_count_steps(True)
        tallies = np.zeros(SW, dtype=np.int64)
        probe = {"lf_steps": 0, "rank_blocks": 0, "seed_lookups": 0, "settled": 0}
        for seg in self.segs:
            self.step([seg])
            torch.cuda.synchronize()
            row = self.d_status[seg[4]].cpu().numpy()
            tallies[:12] += row[:12]
            tallies[14:16] += row[14:16]
            tallies[12:14] = np.maximum(tallies[12:14], row[12:14])       # (k_sweep: longest chain / longest wave, in turns)
            for k_, v_ in self.ix.probe_tally().items():
                probe[k_] += v_
            ow = self.ix.open_words()
            self.open_words = [a_ + b_ for a_, b_ in zip(getattr(self, "open_words", [0] * 8), ow["words"] + ow["positions"])]
        self.ix.set_count_steps(False)
        return tallies, probe

    def verify_sample(self, samples=20000):
        """No oracle fits a multi-Gbp both-strand text: re-derive a sample of the outputs through the count seam -- at
        the reported length the both-strand count is 1, one base shorter (if allowed) it is not."""
        rng = np.random.default_rng(7 + self.rank)
        comp = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
        checked = 0
        for (so, seg_len, cnt, oo, i) in self.segs[:3]:
            u = self.units[i]
            m = min(seg_len, 120_000_000)
            rec = self.wl.record(u.record)[u.start:u.start + m].tobytes()
            n_pos = min(cnt, m - self.KMAX)
            if n_pos <= 0:
                continue
            out = self.d_out[oo:oo + n_pos].cpu().numpy()
            pos = rng.integers(0, n_pos, samples)
            k = out[pos].astype(np.int64)
            ok = k > 0
            pos, k = pos[ok], k[ok]
            rc = rec.translate(comp)[::-1]
            tot = self.ix.count_from_sequence(rec, pos, k) + self.ix.count_from_sequence(rc, m - pos - k, k)
            assert (tot == 1).all(), f"unit {i}: reported length is not unique"
            longer = k > self.KMIN
            tot2 = self.ix.count_from_sequence(rec, pos[longer], k[longer] - 1) + \
                self.ix.count_from_sequence(rc, m - pos[longer] - (k[longer] - 1), k[longer] - 1)
            assert (tot2 > 1).all(), f"unit {i}: a shorter unique length exists"
            checked += int(pos.size)
        return {"sampled_positions": int(self.sum_over_ranks(checked)), "ranks": self.world,
                "property": "count(k)==1 and count(k-1)>1 via nm_count_from_sequence, every rank on its own units"}

    def final_gather(self):
        """the north star's "final gather": one RCCL gather of the per-rank uint8 results to rank 0, outside `value`"""
        if self.world == 1 and not self.dist.is_initialized():
            return None
        torch, dist = self.torch, self.dist
        cdev = torch.device("cpu") if self.rehearse else self.dev    # gloo gathers CPU tensors only
        per = int(self.max_over_ranks(self.d_out.numel()))
        pad = torch.zeros(per, dtype=torch.uint8, device=cdev)
        pad[:self.d_out.numel()].copy_(self.d_out)
        gathered = [torch.empty(per, dtype=torch.uint8, device=cdev) for _ in range(self.world)] if self.rank == 0 else None
        torch.cuda.synchronize()
        self.barrier()
        tg = time.perf_counter()
        dist.gather(pad, gathered, dst=0)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - tg) * 1e3
        got = int(sum(int((g != 0).any()) for g in gathered)) if self.rank == 0 else 0
        return {"ms": ms, "bytes_per_rank": per, "ranks_with_results": got}

    def close(self):
        self.ix.close()
        del self.d_seq, self.d_out, self.d_status
        self.torch.cuda.empty_cache()


def kernels_block(kinds):
    return {KINDS[k]: {"launches": int(n), "total_ms": float(tot), "max_ms": float(mx)} for k, (n, tot, mx) in kinds.items() if n}


def roofline_block(run: Run, kinds, tallies, probe, pmc_summaries, ms_per_step=None):
    """`roofline` for the kernel with the LARGEST total time among the kernels of a segment: the search kernel (k_sites, or
    k_min_unique), the repeat probes (coarse + fine share their counters), k_sweep, or k_resolve (with k_open_words: what is
    left of the finishing stage).  achieved = algorithmic bytes per launch / mean launch duration (HIP events on the stream
    the kernel is launched on).  Passes whose segments overlap on several streams take these events in ONE further pass on one
    stream (events around overlapping kernels would time each other's work): `kernel_ms_per_step_one_stream` is the sum of
    that pass's segments, `ms_per_step` the timed passes', `overlap_gain` their ratio."""
    info = run.ix.info()
    search_name = {1: "k_min_unique", 5: "k_sites"}.get(info["last_range_kernel"], "?")
    n_seg = max(len(run.segs), 1)
    site_m = info.get("last_site_core_length", 0)
    lf_bytes = 16 if run.info["lf_blocks"] else 32
    cand = {}
    n0, t0, _ = kinds[0]
    if search_name == "k_sites":       # 8 B per table word read, 1 sequence byte in, one output element out per position
        cand[search_name] = (n0, t0, tallies[5] * 8 + run.my_positions * 2)
    else:
        cand[search_name] = (n0, t0, tallies[4] * lf_bytes + tallies[5] * 8 + run.my_positions * 2)
    # probes: one 32-byte encoded word + (seed entry 8 B) per probe that walks, 16 B per LF entry read; 4 B word out per stride
    n2, t2, _ = kinds[2]
    n3, t3, _ = kinds[3]
    if n2 or n3:
        cand["k_repeat_probe(_coarse)"] = (max(n3, 1), t2 + t3, probe["rank_blocks"] * lf_bytes + probe["seed_lookups"] * (8 + 32) + run.my_positions / 64 * 4)
    n4, t4, _ = kinds[4]
    n5, t5, _ = kinds.get(5, (0, 0.0, 0.0))
    if n5:                             # 32-byte rank blocks + seed entries; per word its bitmap word, list entry and two planes; one element per open position
        cand["k_sweep"] = (n5, t5, tallies[14] * 32 + tallies[15] * 8 + tallies[9] * 28 + tallies[10] * 0.5)
    if n4:                             # need bitmap in (1 bit per position), LF entries + seed / second-chance words, elements out
        cand["k_resolve"] = (n4, max(t4 - t5, 0.0), (tallies[4] - tallies[14]) * lf_bytes + (tallies[6] - tallies[15]) * 8 + run.my_positions / 8)
    name = max(cand, key=lambda k_: cand[k_][1])
    n_launch, total_ms, alg = cand[name]
    per_launch = alg / n_seg
    avg_ms = total_ms / max(n_launch, 1)
    achieved = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    if run.world == 1:
        traffic, src = measured_traffic(name, site_m, run.my_positions / n_seg, (pmc_summaries or {}).get(name))   # mean over the launches of a pass
    else:
        traffic, src = None, "PMC passes are taken with one rank"
    searched = max(int(tallies[7]), 1)
    n1, t1, _ = kinds[1]
    one_stream = t1 / max(n1, 1) * n_seg          # all kernels of a pass, one stream, HIP events
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": traffic, "traffic_source": src, "kernel": name,
           "scope": "the kernel with the largest total time of a pass (HIP events around it on its launch stream); all kernels are in `kernels` / `pipeline`",
           "avg_launch_ms": avg_ms, "launches": int(n_launch), "algorithmic_bytes_per_launch": per_launch,
           "kernel_ms_per_step_one_stream": one_stream, "dominant_kernel_ms_per_step_one_stream": total_ms / max(n_launch, 1) * n_seg,
           "ms_per_step": ms_per_step, "overlap_gain": one_stream / ms_per_step if ms_per_step else None,
           "events_taken": "in one further pass on one stream (the timed passes deal their segments over several streams)" if run.overlap_identical is not None
                           else "in the timed passes themselves (one stream)",
           "kernel_share_of_pass": {k_: v_[1] / max(sum(c[1] for c in cand.values()), 1e-12) for k_, v_ in cand.items()}}
    if name == "k_sites":
        out.update({"site_core_length": site_m, "positions_per_table_line": run.my_positions / max(tallies[5] / 4.0, 1.0),
                    "table_words_per_position": float(tallies[5] / searched)})
    return out


def pipeline_block(run: Run, kinds, tallies, probe):
    n_seg_launch, all_ms, _ = kinds[1]
    searched = max(int(tallies[7]), 1)
    rank_bytes = 16 if run.info["lf_blocks"] else 32
    alg_all = tallies[5] * 8 + tallies[6] * 8 + tallies[4] * rank_bytes + probe["seed_lookups"] * 8 + probe["rank_blocks"] * rank_bytes + run.my_positions * 2
    return {"kernels": "k_reset_status + k_sites (encodes the raw bytes itself) + k_repeat_probe(_coarse) + k_open_words + k_sweep + k_resolve", "avg_segment_ms": all_ms / max(n_seg_launch, 1),
            "segments": n_seg_launch, "algorithmic_bytes_per_segment": alg_all / max(len(run.segs), 1),
            "resolve": {"lf_steps_per_position": float(tallies[3] / searched), "rank_blocks_per_position": float(tallies[4] / searched),
                        "table_words_per_position": float(tallies[6] / searched),
                        "sweep": {"words": int(tallies[9]), "turns_per_word": float(tallies[10] / max(tallies[9], 1)), "busy_lane_share": float(tallies[10] / max(tallies[11], 1)),
                                  "longest_word_turns": int(tallies[12]), "longest_wave_turns": int(tallies[13]),
                                  "open_words_by_class": getattr(run, "open_words", [0] * 8)[:4], "open_positions_by_class": getattr(run, "open_words", [0] * 8)[4:]}},
            "repeat_probes": {"enabled": bool(run.info.get("repeat_probes", 0)), "settled_fraction": probe["settled"] / max(run.my_positions, 1),
                              "lf_steps_per_position": probe["lf_steps"] / max(run.my_positions, 1),
                              "seed_lookups_per_position": probe["seed_lookups"] / max(run.my_positions, 1)}}


def cpu_baseline(args, wl: Workload, gpu_out: np.ndarray, KMIN: int, KMAX: int):
    """The oracle's C/OpenMP port of the reference algorithm (forward-strand FM-index, restart per probe, forward +
    reverse-complement queries, 4^12 seed table, 32 queries in flight per thread with software prefetch --
    src/newmap-count.c:196 awFmParallelSearchCount as published; schedule newmap/search.py:464-544) on an FM-index of the
    WHOLE workload genome (all records, forward strand: what `newmap index` builds), timed on a bounded sample -- the
    first positions of the first record -- and compared bit for bit with the GPU output of the same positions."""
    from oracle import ref_driver as rd
    threads = rd.lib().or_num_threads()
    if wl.total + len(wl.records) >= (1 << 32) - 64:
        return {"value": None, "note": "the oracle's 32-bit suffix array holds texts below 2^32 symbols"}
    log(f"[bench] cpu baseline: building the oracle's forward-strand FM-index of {wl.total / 1e6:g} Mbp on {threads} threads ...")
    t0 = time.time()
    recs = [wl.record(i) for i in range(len(wl.records))]
    t_gen = time.time() - t0
    t0 = time.time()
    oracle = rd.OracleIndex(recs)
    t_sa = time.time() - t0
    t0 = time.time()
    oracle.enable_fm(12)
    oracle.drop_suffix_array()
    t_fm = time.time() - t0
    genome = recs[0]
    for i in range(1, len(wl.records)):
        wl.drop(i)
    del recs
    avail = len(genome) - KMAX
    # the host: how much CPU time the box grants (cgroup quota) beside how many logical CPUs it shows; the sample is timed with
    # as many threads as CPUs are granted AND with all visible ones (memory-latency bound: more threads than granted CPUs can
    # still pay), the better of the two is `value`
    quota = cpu_quota()
    visible = os.cpu_count() or threads
    counts = sorted({max(1, min(int(round(quota)), visible)) if quota else visible, visible})
    runs = []
    same = True
    for n_thr in counts:
        rd.lib().or_set_num_threads(n_thr)
        calib = max(min(1_000_000, avail // 4), 1)
        t0 = time.time()
        rd.ref_binary_search_segment_c(oracle, genome[:calib + KMAX - 1].tobytes(), calib, KMIN, KMAX, fm=True)
        rate = calib / max(time.time() - t0, 1e-6)
        sample = int(min(max(rate * args.cpu_seconds / len(counts), calib), avail, 100_000_000, gpu_out.size))
        sample = sample // 1_000_000 * 1_000_000 if sample > 2_000_000 else sample
        t0 = time.time()
        got, _, stats = rd.ref_binary_search_segment_c(oracle, genome[:sample + KMAX - 1].tobytes(), sample, KMIN, KMAX, fm=True)
        dt = time.time() - t0
        ok = bool(np.array_equal(got.astype(np.uint8), gpu_out[:sample]))
        same = same and ok
        runs.append({"threads": n_thr, "value": sample / dt, "sample_positions": sample, "sample_seconds": dt, "bit_exact_vs_gpu": ok,
                     "probes_per_position": stats["probes"] / sample, "lf_steps_per_position": 2 * stats["probe_len"] / sample})
        log(f"[bench] cpu baseline: {sample} positions in {dt:.1f}s on {n_thr} threads (CPU quota {quota}, {visible} visible; records {t_gen:.1f}s, "
            f"suffix array {t_sa:.1f}s, FM blocks + seed table {t_fm:.1f}s); bit-exact vs GPU: {ok}")
    best = max(runs, key=lambda r_: r_["value"])
    threads, sample, dt = best["threads"], best["sample_positions"], best["sample_seconds"]
    stats = {"probes": best["probes_per_position"] * sample, "probe_len": best["lf_steps_per_position"] * sample / 2}
    if not same:
        raise SystemExit("GPU output differs from the CPU oracle on the baseline sample")
    del oracle
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": sample / dt, "unit": "positions/s", "cores": threads, "threads": threads, "cpu_quota": quota, "cpus_visible": visible,
            "positions_per_s_per_granted_cpu": (sample / dt) / quota if quota else None, "runs": runs, "kind": "port",
            "host": f"{cpu_model}, {os.cpu_count()} logical CPUs visible, oracle built -O3 -march=x86-64-v3 (AVX2, POPCNT), OpenMP, "
                    "32 backward searches in flight per thread with software prefetch",
            "index": f"forward-strand FM-index of all {len(wl.records)} records of the workload genome ({wl.total} bases; 128-byte blocks of 256 rows, "
                     f"4^12 seed table), built in {t_sa + t_fm:.1f}s (not timed in `value`)",
            "sample": f"first {sample} positions of record {wl.records[0][0]} of the {wl.total}-base workload, {KMIN}:{KMAX}, "
                      f"{stats['probes'] / sample:.1f} probes and {2 * stats['probe_len'] / sample:.0f} LF steps "
                      "per position (reference schedule), index build excluded; output bit-exact vs the GPU's",
            "extrapolated": sample < wl.total, "sample_positions": sample, "sample_seconds": dt,
            "bit_exact_vs_gpu": same}


def cpu_quota():
    """CPUs of time the container is granted: cgroup v2 cpu.max (quota / period), v1 cfs_quota_us / cfs_period_us; None = no limit"""
    try:
        q, p_ = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        return None if q == "max" else float(q) / float(p_)
    except (OSError, ValueError):
        pass
    try:
        q = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
        p_ = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
        return None if q <= 0 else q / p_
    except (OSError, ValueError):
        return None


def end_to_end(args, wl: Workload, fa, idx, dev_index, gpu_run: Run):
    """the same genome through the native driver: FASTA in -> <id>.unique.uint8 files out (the one-shot CLI's small
    tables); files compared with the bench outputs of the first units"""
    from newmap_amd.engine import Index
    out = Path(args.workdir) / f"{wl.key}_e2e_out"
    shutil.rmtree(out, ignore_errors=True)
    out.mkdir(parents=True)
    t0 = time.time()
    with Index(idx, dev_index, "auto-small") as ix:
        t_open = time.time() - t0
        t0 = time.time()
        total = ix.search_fasta(fa, out, [wl.krange[0], wl.krange[1]], True, True, REFERENCE_BATCH)   # the CLI's default --kmer-batch-size
        t_search = time.time() - t0
        # (the first call of a handle also creates its lanes' scratch and the driver's pinned slots; a resident handle's next call,
        #  into an empty directory again -- truncating 3 GB of cached files is not the driver's time:)
        shutil.rmtree(out, ignore_errors=True)
        out.mkdir(parents=True)
        t0 = time.time()
        ix.search_fasta(fa, out, [wl.krange[0], wl.krange[1]], True, True, REFERENCE_BATCH)
        t_again = time.time() - t0
    checked = 0
    for (so, seg_len, cnt, oo, i) in gpu_run.segs[:2]:
        u = gpu_run.units[i]
        name = wl.records[u.record][0]
        n = min(cnt, 50_000_000)
        got = np.fromfile(out / f"{name}.unique.uint8", dtype=np.uint8, count=n, offset=u.start)
        want = gpu_run.d_out[oo:oo + n].cpu().numpy()
        if not np.array_equal(got, want):
            raise SystemExit(f"end-to-end output of {name} differs from the bench pass")
        checked += n
    files = sorted(out.iterdir())
    res = {"what": "nm_search_fasta: FASTA in -> <id>.unique.uint8 files out (auto-small tables), 1 GPU",
           "cli_search_s": t_search, "cli_search_s_second_call": t_again, "index_open_s": t_open, "positions": int(total["positions"]),
           "positions_per_s": total["positions"] / t_search, "positions_per_s_with_index_open": total["positions"] / (t_search + t_open),
           "files": len(files), "bytes_written": int(sum(f.stat().st_size for f in files)),
           "verified_bytes_equal_bench_pass": checked}
    shutil.rmtree(out, ignore_errors=True)
    return res


def configs1_block(args, rank, dev, barrier, dist, rehearse):
    """BASELINE configs[1] (round 2's headline): 100 Mbp single record, 20:200, ONE launch of 100 M positions per pass,
    plus the same passes cut into the reference's 10 M batch"""
    from newmap_amd import parallel
    wl = configs1_workload()
    KMIN, KMAX = wl.krange
    fa, idx_path, prep = prepare_index(args, wl, rank, barrier)
    ranges = [(0, wl.total)]
    units = parallel.units_for_ranges(wl.lengths, ranges, wl.total, KMAX)
    run = Run(args, wl, idx_path, units, rank, 1, dev, barrier, dist, rehearse, args.seed_length)
    steps = max(args.steps, 50)
    elapsed, kinds = run.timed(steps, args.warmup)
    run.check_status()
    ref_units = parallel.units_for_ranges(wl.lengths, ranges, REFERENCE_BATCH, KMAX)
    row0 = run.more_status_rows(len(ref_units))
    ref_segs = [(ru.start, ru.seg_len, ru.count, ru.start, row0 + j) for j, ru in enumerate(ref_units)]
    ref_elapsed, _ = run.timed(steps, 1, ref_segs, kernel_events=False)
    ref_batch = {"batch": REFERENCE_BATCH, "value": wl.total * steps / ref_elapsed, "unit": "positions/s",
                 "ms_per_step": ref_elapsed / steps * 1e3, "launches_per_step": len(ref_segs), "streams": len(run.streams)}
    if len(run.streams) > 1:
        saved, run.streams = run.streams, run.streams[:1]
        one_elapsed, _ = run.timed(steps, 1, ref_segs, kernel_events=False)
        run.streams = saved
        ref_batch["value_one_stream"] = wl.total * steps / one_elapsed
    run.check_status()
    tallies, probe = run.counters()
    block = {"workload": wl.desc + f", search-range {KMIN}:{KMAX}, both strands, one launch per pass", "positions": wl.total,
             "steps": steps, "warmup": args.warmup, "ms_per_step": elapsed / steps * 1e3, "value": wl.total * steps / elapsed, "unit": "positions/s",
             "index_bytes_hbm": run.info["device_bytes"], "index_open_s": run.t_open,
             "seed_length": run.info["seed_length"], "quad_core_length": run.info.get("quad_core_length", 0),
             "quad_small_core_length": run.info.get("quad_small_core_length", 0),
             "roofline": roofline_block(run, kinds, tallies, probe, PMC_SUMMARIES["c2"], elapsed / steps * 1e3), "kernels": kernels_block(kinds),
             "pipeline": pipeline_block(run, kinds, tallies, probe), "reference_batch": ref_batch,
             "verify": run.verify_sample(),
             "host": {"fasta_write_s": prep["fasta_write_s"], "index_build_s": prep["index_build_s"], "index_builder": args.index_builder}}
    run.close()
    wl.drop()
    return block


def hs_block(args, rank, dev, barrier, dist, rehearse):
    """The human-shaped genome (the stand-in of configs[3]'s GRCh38: repeat families, segmental duplications, soft-masking, N
    runs) beside the uniform headline: range mode 20:200 with its roofline (the dominant kernel is k_sweep) and pipeline
    counters, and configs[3]'s own mode -- fixed-k list mode, k = 36 and k = 100 -- on the same resident units"""
    from newmap_amd import parallel
    torch = __import__("torch")
    wl = hs_workload(args.hs_mbp)
    KMIN, KMAX = wl.krange
    fa, idx_path, prep = prepare_index(args, wl, rank, barrier)
    units = parallel.units_for_ranges(wl.lengths, [(0, wl.total)], max(args.batch, 1), KMAX)
    import copy
    hs_args = copy.copy(args)
    if not args.streams_given:
        hs_args.streams = 5                                 # (see --streams)
    run = Run(hs_args, wl, idx_path, units, rank, 1, dev, barrier, dist, rehearse, args.seed_length)
    wl.drop()
    steps = max(3, min(args.steps, 5))
    elapsed, kinds = run.timed(steps, 1)
    run.check_status()
    tallies, probe = run.counters()
    block = {"workload": wl.desc + f", search-range {KMIN}:{KMAX}, both strands, one launch per record", "positions": wl.total,
             "steps": steps, "warmup": 1, "ms_per_step": elapsed / steps * 1e3, "value": wl.total * steps / elapsed, "unit": "positions/s",
             "streams": len(run.streams), "index_bytes_hbm": run.info["device_bytes"], "index_open_s": run.t_open,
             "seed_length": run.info["seed_length"], "quad_core_length": run.info.get("quad_core_length", 0),
             "quad_small_core_length": run.info.get("quad_small_core_length", 0), "dict_length": run.info.get("dict_length", 0),
             "roofline": roofline_block(run, kinds, tallies, probe, PMC_SUMMARIES["hs"] if args.hs_mbp is None else None, elapsed / steps * 1e3),
             "kernels": kernels_block(kinds), "pipeline": pipeline_block(run, kinds, tallies, probe), "verify": run.verify_sample(),
             "unique_fraction": float((run.d_out[:min(run.d_out.numel(), 200_000_000)] != 0).float().mean().item()),
             "host": {"fasta_write_s": prep["fasta_write_s"], "index_build_s": prep["index_build_s"], "index_builder": args.index_builder}}
    # configs[3]'s mode on the same genome: one length per run, both strands (newmap/search.py:551-644)
    sp, op, st = run.d_seq.data_ptr(), run.d_out.data_ptr(), run.d_status.data_ptr()
    lists = {}
    for ks in ([36], [100], [24, 36, 50, 100]):
        def one_pass():
            for j, (so, seg_len, cnt, oo, i) in enumerate(run.segs):
                run.ix.fixed_k_segment_dev(sp + so, seg_len, cnt, ks, True, 1, op + oo, st + 8 * SW * i, run.streams[j % len(run.streams)])
        one_pass()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_pass()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        run.check_status()
        lists["k" + "_".join(str(k) for k in ks)] = {"value": wl.total / dt, "unit": "positions/s", "ms_per_step": dt * 1e3,
                                                     "nonzero_fraction": float((run.d_out[:min(run.d_out.numel(), 200_000_000)] != 0).float().mean().item())}
    block["list_mode"] = lists
    run.close()
    return block


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))          # nothing below has run: no torch import, no HIP call in this process
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    import torch                                   # first: libnewmap_amd.so binds to torch's HIP runtime
    import torch.distributed as dist
    from newmap_amd import parallel

    # one rank per GPU.  NEWMAP_AMD_BENCH_REHEARSE=1 (rehearsal of the N > 1 code path on a box with fewer GPUs
    # than ranks): ranks share the devices and the collectives go over gloo -- RCCL refuses two ranks on one GPU
    rehearse = os.environ.get("NEWMAP_AMD_BENCH_REHEARSE") == "1"
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    rccl_ranks = None
    launched = "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ      # by torch.distributed.run: the collectives run, with one rank too
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
            ones = torch.ones(1, device=dev)
            dist.all_reduce(ones)                   # the world RCCL actually formed
            rccl_ranks = int(ones.item())

    def barrier():
        if world > 1 or launched:
            dist.barrier()

    t_bench = time.time()
    wl = headline_workload(args)
    KMIN, KMAX = wl.krange
    fa, idx_path, prep = prepare_index(args, wl, rank, barrier)
    # fixed genome at every N: interleaved ~64 M chunks, chunk c owned by rank c mod N (one rank: the records themselves)
    ranges = parallel.interleaved_ranges(wl.total, world, CHUNK)[rank] if world > 1 else [(0, wl.total)]
    units = parallel.units_for_ranges(wl.lengths, ranges, max(args.batch, 1), KMAX)
    run = Run(args, wl, idx_path, units, rank, world, dev, barrier, dist, rehearse, args.seed_length)
    keep = {u.record for u in units[:3]} | {0}
    for i in range(len(wl.records)):               # the device holds the units now
        if i not in keep:
            wl.drop(i)
    elapsed, kinds = run.timed(args.steps, args.warmup)
    run.check_status()
    total_positions = run.sum_over_ranks(run.my_positions)
    if rank == 0:
        log(f"[bench] {wl.key}: {args.steps} steps in {elapsed:.3f}s; kernel events: {kernels_block(kinds)}")
    pass_ms = None if args.no_spread else run.spread(min(args.steps, 20))
    gather = run.final_gather()
    tallies, probe = run.counters()
    verify = run.verify_sample()
    result = None
    if rank == 0:
        result = {
            "metric": f"genome positions/sec (min-unique-k search, {KMIN}:{KMAX})",
            "value": total_positions * args.steps / elapsed,
            "unit": "positions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl.desc + f", search-range {KMIN}:{KMAX}, both strands; the genome is fixed as N grows: its positions are dealt "
                                             f"to the ranks in interleaved chunks of ~{CHUNK >> 20} M",
                       "records": len(wl.records), "positions": int(total_positions), "positions_this_rank": run.my_positions,
                       "batch": args.batch, "launches_per_step_per_rank": len(run.segs),
                       "streams": len(run.streams) if len(run.segs) > 1 else 1,
                       "streams_output_identical_to_one_stream": run.overlap_identical,
                       "seed_length": run.info["seed_length"], "quad_core_length": run.info.get("quad_core_length", 0),
                       "quad_small_core_length": run.info.get("quad_small_core_length", 0),
                       "index_bytes_hbm": run.info["device_bytes"], "index_open_s": run.t_open, "bwt_rows": run.info["bwt_length"],
                       "parallelism": f"independent work units over {world} GPU(s), index replicated, no data-path collective"},
            "pass_ms": pass_ms,
            "roofline": roofline_block(run, kinds, tallies, probe, PMC_SUMMARIES.get(args.config) if args.mbp is None else None, elapsed / args.steps * 1e3),
            "kernels": kernels_block(kinds),
            "pipeline": pipeline_block(run, kinds, tallies, probe),
            "verify": verify,
            "host": {"fasta_write_s": prep["fasta_write_s"], "index_build_s": prep["index_build_s"], "index_builder": args.index_builder,
                     "index_open_s": run.t_open, "sequence_upload_s": run.t_upload},
        }
        if world > 1 or launched:
            result["rccl_ranks"] = rccl_ranks
            result["collective_backend"] = "gloo (rehearsal: ranks share a GPU)" if rehearse else "nccl (RCCL)"
            result["final_gather_ms"] = gather["ms"]      # one gather of all results to rank 0, outside `value`
            result["final_gather_bytes_per_rank"] = gather["bytes_per_rank"]
            result["final_gather_ranks_with_results"] = gather["ranks_with_results"]
    if world == 1:
        gpu_sample = run.d_out[:min(run.segs[0][2], 100_000_000)].cpu().numpy() if run.segs and units[0].record == 0 and units[0].start == 0 else None
        if not args.no_end_to_end:
            result["end_to_end"] = end_to_end(args, wl, fa, idx_path, dev_index, run)
        run.close()
        del run
        if args.config == "hs":
            args.no_cpu_baseline = True                                   # (the oracle's comparison sort is slow on the repeat families; list / range parity: tests)
        if args.config == "c5" and not args.no_cpu_baseline:
            log("[bench] no CPU baseline on the tandem genome (the oracle's comparison sort of whole suffixes is quadratic in a 50 kb array)")
        if not args.no_cpu_baseline and gpu_sample is not None and args.config != "c5":
            result["cpu_baseline"] = cpu_baseline(args, wl, gpu_sample, KMIN, KMAX)
            result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"] if result["cpu_baseline"].get("value") else None
        wl.drop()
        if not args.no_configs1 and args.config == "ns":
            t1 = time.time()
            result["configs1"] = configs1_block(args, rank, dev, barrier, dist, rehearse)
            log(f"[bench] configs1 block: {time.time() - t1:.1f}s")
        if not args.no_hs and not args.no_configs1 and args.config == "ns":      # (--no-configs1: no nested blocks at all)
            if args.hs_mbp is None:
                args.hs_mbp = args.mbp                     # a shrunk headline (rehearsals, tests) shrinks this one too
            t1 = time.time()
            result["hs"] = hs_block(args, rank, dev, barrier, dist, rehearse)
            log(f"[bench] hs block: {time.time() - t1:.1f}s")
    else:
        run.close()
    if rank == 0:
        result["bench_wall_s"] = time.time() - t_bench
        print(json.dumps(result), flush=True)
    barrier()
    if world > 1 or launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
