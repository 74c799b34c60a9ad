#!/usr/bin/env python3
"""bench.py -- genome positions/sec of the min-unique-k search (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Two workloads per run, both with the sequence bytes resident in HBM before the timed region and one STEP = one pass of
the hot path (k_sites with its encode stage, repeat probes, k_resolve) over every position of the rank's work units.  A
pass of several segments (north star, `reference_batch`, --config c3 / c5) deals them round-robin over --streams HIP
streams (default 2): the handle keeps its launch scratch per stream, neighbouring segments overlap; the headline is ONE
launch per pass on one stream, and its kernel is timed with HIP events on that stream in the timed passes themselves:

  * headline (`value`): BASELINE.json configs[1] -- synthetic 100 Mbp single-record FASTA, uniform ACGT
    (numpy default_rng(20260515)), search range 20:200.  At N > 1 the path shards by independent units with no
    data-path collective, so the job grows with N ("scaling": "weak"): N records of 100 Mbp (seeds 20260515 + i) in ONE
    index, replicated in every GPU's HBM; rank r searches record r.  `value` = positions all ranks searched / the
    slowest rank's time.
  * `north_star`: the configuration BASELINE.json's north_star states its target on -- the ~3 Gbp genome of
    configs[2] (24 human-shaped records, 3.09 Gbp) searched at 20:200 on a device-built index, its work units dealt to
    the ranks in interleaved chunks (fixed genome: strong scaling), plus, at N = 1, the same genome through the native
    driver FASTA in -> files out (`end_to_end`).

After the timed region ONE RCCL gather (N > 1) collects the per-rank uint8 results on rank 0 (`final_gather_ms`,
outside `value`).  Rank 0 prints ONE JSON line (README / DESIGN.md "Measurement").
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import shutil
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BATCH = 100_000_000            # positions per launch.  The reference's --kmer-batch-size default (10 M) exists "to control
#                                memory usage" on the host; with 288 GB of HBM a launch takes a whole 100 Mbp record.  The
#                                throughput at the reference's 10 M is measured in the same run (`reference_batch`).
REFERENCE_BATCH = 10_000_000
C2_SEED, C2_BASES = 20260515, 100_000_000
PMC_SUMMARY = ROOT / "profiles" / "round2" / "pmc_sites_kernel_summary.csv"
PMC_SUMMARY_NS = ROOT / "profiles" / "round2" / "pmc_ns_sites_kernel_summary.csv"     # the north-star block (tools/profile_ns.sh)
KERNEL_SOURCES = [ROOT / "newmap_amd" / "csrc" / "nm_engine.hip", ROOT / "newmap_amd" / "csrc" / "nm_core.h"]


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)        # 100 passes of ~0.4 ms: a timed region of ~40 ms
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=["c2", "c3", "c5"], default="c2",
                    help="headline workload: c2 = BASELINE configs[1] (default); c3 / c5 = configs[2] / configs[4] at --mbp (capability runs)")
    ap.add_argument("--mbp", type=float, default=None, help="genome size in Mbp of a c3 / c5 capability run")
    ap.add_argument("--seed-length", default="auto",
                    help="device tables: auto (default: sized for throughput), auto-small (<= 20 GB, what the one-shot CLI uses), "
                         "file (the index's seed length, 12), or a seed length 0..16")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--streams", type=int, default=2, choices=[1, 2, 3, 4, 5],
                    help="the segments of a pass are launched round-robin on this many HIP streams (the handle keeps one set of launch "
                         "scratch per stream, so neighbouring segments overlap); passes of one segment use one stream")
    ap.add_argument("--index-builder", choices=["host", "device"], default="device",
                    help="suffix sort on the GPU (default) or on the host cores (same index file; not timed in `value`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-batch", action="store_true",
                    help="skip the extra passes at the reference's 10 M batch (profiling runs: one launch size per kernel)")
    ap.add_argument("--no-north-star", action="store_true", help="skip the 3 Gbp / 20:200 block (profiling runs)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the FASTA-in -> files-out leg of the north-star block (kernel traces)")
    ap.add_argument("--north-star-mbp", type=float, default=None, help="shrink the north-star genome (rehearsals)")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="target CPU time of the baseline sample (30 s: the whole 100 Mbp on the 128 threads of the GPU box)")
    ap.add_argument("--workdir", default=os.environ.get("NEWMAP_AMD_BENCH_DIR", "/tmp/newmap_amd_bench"))
    return ap.parse_args(argv)


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
            self._cache[i] = synth.uniform_dna(n, seed) if kind == "uniform" else synth.tandem_dna(n, seed)
        return self._cache[i]

    def drop(self, i):
        self._cache.pop(i, None)


def headline_workload(args, world) -> Workload:
    from newmap_amd import synth
    if args.config == "c2":
        if world == 1:
            return Workload("c2_100mbp", "configs[1]: synthetic 100 Mbp single-record FASTA (uniform ACGT, seed 20260515)",
                            (20, 200), [("chr1", C2_SEED, C2_BASES, "uniform")])
        recs = [(f"chr{i + 1}", C2_SEED + i, C2_BASES, "uniform") for i in range(world)]
        return Workload(f"c2_weak_{world}x100mbp",
                        f"configs[1] per GPU: {world} records of 100 Mbp (uniform ACGT, seeds 20260515+i) in one replicated index, rank r searches record r",
                        (20, 200), recs)
    if args.config == "c3":
        total = sum(synth.HUMAN_SHAPED)
        f = 1.0 if args.mbp is None else args.mbp * 1e6 / total
        names = [f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY"]
        recs = [(nm, 20260516 + i, max(1000, int(L * f)), "uniform") for i, (nm, L) in enumerate(zip(names, synth.HUMAN_SHAPED))]
        w = Workload(f"c3_{sum(r[2] for r in recs) / 1e6:g}mbp", "configs[2]: synthetic {mbp:g} Mbp FASTA as 24 human-shaped records (uniform ACGT, seeds 20260516+i)", (24, 150), recs)
        w.desc = w.desc.format(mbp=w.total / 1e6)
        return w
    n = int((args.mbp or 1000) * 1e6)
    return Workload(f"c5_{n / 1e6:g}mbp", f"configs[4]: synthetic {n / 1e6:g} Mbp, 50 % tandem repeats (seed 20260517)", (20, 255),
                    [("rep1", 20260517, n, "tandem")])


def north_star_workload(args) -> Workload:
    from newmap_amd import synth
    total = sum(synth.HUMAN_SHAPED)
    f = 1.0 if args.north_star_mbp is None else args.north_star_mbp * 1e6 / total
    names = [f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY"]
    recs = [(nm, 20260516 + i, max(1000, int(L * f)), "uniform") for i, (nm, L) in enumerate(zip(names, synth.HUMAN_SHAPED))]
    w = Workload(f"ns_{sum(r[2] for r in recs) / 1e6:g}mbp", "", (20, 200), recs)
    w.desc = (f"north_star: the synthetic ~3 Gbp genome of configs[2] ({w.total / 1e6:g} Mbp, 24 human-shaped records, uniform ACGT, "
              "seeds 20260516+i) searched at the north-star range 20:200")
    return w


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


def measured_traffic(kernel: str, quad_m: int, positions_per_launch: float, summary: Path = None):
    """HBM-side read + write bytes per launch of the dominant kernel from the PMC pass committed under profiles/
    (rocprofv3 --pmc cannot run inside this process).  The summary names the kernel sources it was measured on
    (sha256 of nm_engine.hip + nm_core.h), the core length of the table the sites read and the launch size; a summary
    of OTHER sources, another table or another launch size reports null with the reason.  Reads: TCC_EA0_RDREQ x 128 B
    (on gfx950 every read request of this gather is a 128-byte one, TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ -- the guide's
    "FETCH_SIZE reports half" correction stated exactly); writes: TCC_EA0_WRREQ x 64 B."""
    PMC_SUMMARY = summary or globals()["PMC_SUMMARY"]
    if not PMC_SUMMARY.exists():
        return None, f"{PMC_SUMMARY.relative_to(ROOT)} not collected yet"
    meta, vals = {}, {}
    for line in PMC_SUMMARY.read_text().splitlines():
        if line.startswith("#"):
            for kv in line[1:].split(","):
                if "=" in kv:
                    k, v = kv.strip().split("=", 1)
                    meta[k] = v
        elif "," in line and not line.startswith("counter"):
            k, v = line.rsplit(",", 1)
            vals[k] = float(v)
    if not meta.get("kernel", "").startswith(kernel):          # "k_sites" / "k_sites<true" (the > 2^31-row instantiation)
        return None, f"summary is for {meta.get('kernel')}, the run's dominant kernel is {kernel}"
    if meta.get("source_sha256") != source_hash():
        return None, f"summary was measured on other kernel sources ({meta.get('source_sha256')} != {source_hash()}): re-run tools/profile_c2.sh"
    if int(meta.get("site_core_length", -1)) != quad_m or abs(float(meta.get("positions_per_launch", 0)) - positions_per_launch) > 0.01 * positions_per_launch:
        return None, "summary was measured with another table or launch size"
    if "TCC_EA0_RDREQ_sum" not in vals:
        return None, "summary lacks TCC_EA0_RDREQ_sum"
    return vals["TCC_EA0_RDREQ_sum"] * 128.0 + vals.get("TCC_EA0_WRREQ_sum", 0.0) * 64.0, str(PMC_SUMMARY.relative_to(ROOT))


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
        self.d_status = torch.zeros((max(len(units), 1), 8), dtype=torch.int64, device=dev)
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
            self.ix.min_unique_segment_dev(sp + so, seg_len, cnt, self.KMIN, self.KMAX, True, 1, op + oo, st + 64 * i, streams[j % len(streams)])

    def max_over_ranks(self, x: float) -> float:
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.torch.device("cpu") if self.rehearse else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x: float) -> float:
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.torch.device("cpu") if self.rehearse else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def timed(self, steps, warmup, segs=None, kernel_events=True):
        """`steps` passes between barrier + synchronize on both sides.  Kernel times (HIP events on the launch stream,
        read_timing kinds 0 and 1) are taken in the timed passes themselves when a pass runs on ONE stream (the headline:
        one launch per pass); passes whose segments overlap on several streams get one extra pass on one stream for
        them afterwards (outside `elapsed`), since events around overlapping kernels time each other's work."""
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
        k0 = k1 = (0, 0.0, 0.0)
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
            k0 = self.ix.read_timing(0)
            k1 = self.ix.read_timing(1)
            self.ix.set_timing(False)
        return self.max_over_ranks(elapsed), k0, k1

    def check_status(self):
        st = self.d_status.cpu().numpy()
        bad = np.flatnonzero(st[:, 1])
        if bad.size:
            where = f"record {self.wl.records[self.units[int(bad[0])].record][0]}" if int(bad[0]) < len(self.units) else f"status row {int(bad[0])}"
            raise SystemExit(f"k-mer not found in the index: {where}, segment position {int(st[bad[0], 2])}")

    def more_status_rows(self, n):
        """`n` further status rows (segments cut differently from the resident units get their own); returns the first"""
        first = self.d_status.shape[0]
        self.d_status = self.torch.cat([self.d_status, self.torch.zeros((n, 8), dtype=self.torch.int64, device=self.dev)])
        return first

    def counters(self):
        """counter build of the kernels, one pass: [3] LF steps, [4] rank blocks, [5] table words read by k_sites,
        [6] table words read by k_resolve (second chance + seed entries), [7] positions searched"""
        torch = self.torch
        self.ix.set_count_steps(True)
        tallies = np.zeros(8, dtype=np.int64)
        probe = {"lf_steps": 0, "rank_blocks": 0, "seed_lookups": 0, "settled": 0}
        for seg in self.segs:
            self.step([seg])
            torch.cuda.synchronize()
            tallies += self.d_status[seg[4]].cpu().numpy()
            for k_, v_ in self.ix.probe_tally().items():
                probe[k_] += v_
        self.ix.set_count_steps(False)
        return tallies, probe

    def verify_sample(self, samples=20000):
        """No oracle fits a multi-Gbp genome: re-derive a sample of the outputs through the count seam -- at the
        reported length the both-strand count is 1, one base shorter (if allowed) it is not."""
        rng = np.random.default_rng(7 + self.rank)
        comp = bytes.maketrans(b"ACGT", b"TGCA")
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
        if self.world == 1:
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
        return {"ms": (time.perf_counter() - tg) * 1e3, "bytes_per_rank": per}

    def close(self):
        self.ix.close()
        del self.d_seq, self.d_out, self.d_status
        self.torch.cuda.empty_cache()


def roofline_block(run: Run, k0, tallies, config_key):
    n_launch, kern_ms, _ = k0
    kernel_name = {1: "k_min_unique", 5: "k_sites"}.get(run.ix.info()["last_range_kernel"], "?")
    n_seg = max(len(run.segs), 1)
    # algorithmic bytes of the DOMINANT kernel per launch (DESIGN.md "Measurement"): 8 B per table word it reads,
    # 1 sequence byte in and one output element out per position.  k_min_unique (no quad table): + 16 B per rank structure
    site_m = run.ix.info().get("last_site_core_length", 0)
    if kernel_name == "k_sites":
        alg = tallies[5] * 8 + run.my_positions * 2
    else:
        alg = tallies[4] * (16 if run.info["lf_blocks"] else 32) + tallies[5] * 8 + run.my_positions * 2
    per_launch = alg / n_seg
    avg_ms = kern_ms / max(n_launch, 1)
    achieved = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    if config_key == "c2_100mbp":
        traffic, src = measured_traffic(kernel_name, site_m, run.my_positions / n_seg)
    elif config_key.startswith("ns_") and run.world == 1:
        traffic, src = measured_traffic(kernel_name, site_m, run.my_positions / n_seg, PMC_SUMMARY_NS)   # mean over the launches of a pass
    else:
        traffic, src = None, "no PMC pass for this workload"
    searched = max(int(tallies[7]), 1)
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src, "kernel": kernel_name, "scope": "the dominant kernel alone (HIP events around it); "
            "encode pass, repeat probes and k_resolve are in `pipeline`", "avg_launch_ms": avg_ms, "launches": n_launch,
            "algorithmic_bytes_per_launch": per_launch, "site_core_length": site_m,
            "positions_per_table_line": run.my_positions / max(tallies[5] / 4.0, 1.0) if kernel_name == "k_sites" else None,
            "table_words_per_position": float(tallies[5] / searched)}


def pipeline_block(run: Run, k1, tallies, probe):
    n_seg_launch, all_ms, _ = k1
    searched = max(int(tallies[7]), 1)
    rank_bytes = 16 if run.info["lf_blocks"] else 32
    alg_all = tallies[5] * 8 + tallies[6] * 8 + tallies[4] * rank_bytes + probe["seed_lookups"] * 8 + probe["rank_blocks"] * rank_bytes + run.my_positions * 2
    return {"kernels": "k_reset_status + k_sites (encodes the raw bytes itself) + k_repeat_probe(_coarse) + k_resolve", "avg_segment_ms": all_ms / max(n_seg_launch, 1),
            "segments": n_seg_launch, "algorithmic_bytes_per_segment": alg_all / max(len(run.segs), 1),
            "resolve": {"lf_steps_per_position": float(tallies[3] / searched), "rank_blocks_per_position": float(tallies[4] / searched),
                        "table_words_per_position": float(tallies[6] / searched)},
            "repeat_probes": {"enabled": bool(run.info.get("repeat_probes", 0)), "settled_fraction": probe["settled"] / max(run.my_positions, 1),
                              "lf_steps_per_position": probe["lf_steps"] / max(run.my_positions, 1),
                              "seed_lookups_per_position": probe["seed_lookups"] / max(run.my_positions, 1)}}


def cpu_baseline(args, genome: np.ndarray, gpu_out: np.ndarray, KMIN: int, KMAX: int):
    """The oracle's C/OpenMP port of the reference algorithm (forward-strand FM-index, restart per probe, forward +
    reverse-complement queries, 4^12 seed table, 32 queries in flight per thread with software prefetch --
    src/newmap-count.c:196 awFmParallelSearchCount as published) on a bounded sample, checked against the GPU output."""
    from oracle import ref_driver as rd
    threads = rd.lib().or_num_threads()
    log(f"[bench] cpu baseline: building the oracle's forward-strand FM-index on {threads} threads ...")
    t0 = time.time()
    oracle = rd.OracleIndex([genome.tobytes()])
    oracle.enable_fm(12)
    t_index = time.time() - t0
    calib = min(2_000_000, len(genome) // 4)
    t0 = time.time()
    rd.ref_binary_search_segment_c(oracle, genome[:calib + KMAX - 1].tobytes(), calib, KMIN, KMAX, fm=True)
    rate = calib / max(time.time() - t0, 1e-6)
    sample = int(min(max(rate * args.cpu_seconds, calib), len(genome) - KMAX))
    sample = sample // 1_000_000 * 1_000_000 if sample > 2_000_000 else sample
    t0 = time.time()
    got, _, stats = rd.ref_binary_search_segment_c(oracle, genome[:sample + KMAX - 1].tobytes(), sample, KMIN, KMAX, fm=True)
    dt = time.time() - t0
    same = bool(np.array_equal(got.astype(np.uint8), gpu_out[:sample]))
    log(f"[bench] cpu baseline: {sample} positions in {dt:.1f}s on {threads} threads (oracle index {t_index:.1f}s); bit-exact vs GPU: {same}")
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
            "host": f"{cpu_model}, {os.cpu_count()} logical CPUs visible, oracle built -O3 -march=x86-64-v3 (AVX2, POPCNT), OpenMP, "
                    "32 backward searches in flight per thread with software prefetch",
            "sample": f"first {sample} positions of the {len(genome)}-base workload, {KMIN}:{KMAX}, "
                      f"{stats['probes'] / sample:.1f} probes and {2 * stats['probe_len'] / sample:.0f} LF steps "
                      "per position (reference schedule), index build excluded; output bit-exact vs the GPU's",
            "bit_exact_vs_gpu": same}


def end_to_end(args, wl: Workload, fa, idx, dev_index, gpu_run: Run):
    """the same genome through the native driver: FASTA in -> <id>.unique.uint8 files out (index already open, the
    one-shot CLI's small tables); files compared with the bench outputs of the first units"""
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
    res = {"what": "nm_search_fasta: FASTA in -> <id>.unique.uint8 files out, index already open (auto-small tables), 1 GPU",
           "cli_search_s": t_search, "index_open_s": t_open, "positions": int(total["positions"]),
           "positions_per_s": total["positions"] / t_search, "files": len(files), "bytes_written": int(sum(f.stat().st_size for f in files)),
           "verified_bytes_equal_bench_pass": checked}
    shutil.rmtree(out, ignore_errors=True)
    return res


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
    from newmap_amd import parallel

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

    t_bench = time.time()
    # ------------------------------------------------------------------ headline
    wl = headline_workload(args, world)
    KMIN, KMAX = wl.krange
    fa, idx_path, prep = prepare_index(args, wl, rank, barrier)
    if world == 1 or args.config != "c2":
        ranges = parallel.interleaved_ranges(wl.total, world, max(args.batch, 1))[rank] if world > 1 else [(0, wl.total)]
    else:                                           # weak scaling: rank r owns record r
        base = int(sum(wl.lengths[:rank]))
        ranges = [(base, base + wl.lengths[rank])]
    units = parallel.units_for_ranges(wl.lengths, ranges, args.batch, KMAX)
    run = Run(args, wl, idx_path, units, rank, world, dev, barrier, dist, rehearse, args.seed_length)
    elapsed, k0, k1 = run.timed(args.steps, args.warmup)
    run.check_status()
    total_positions = run.sum_over_ranks(run.my_positions)
    if rank == 0:
        log(f"[bench] {wl.key}: {args.steps} steps in {elapsed:.3f}s; {k0[0]} launches of the dominant kernel, {k0[1]:.2f} ms in it "
            f"(max {k0[2]:.3f} ms); all kernels of the segments {k1[1]:.2f} ms")
    ref_batch = None
    if args.batch > REFERENCE_BATCH and not args.no_reference_batch:
        # the same passes cut into the reference's default batch (10 M positions per launch), for comparison
        ref_units = parallel.units_for_ranges(wl.lengths, ranges, REFERENCE_BATCH, KMAX)
        by_unit = {}
        for i, u in enumerate(units):
            by_unit[(u.record, u.start)] = i
        ref_segs = []
        row0 = run.more_status_rows(len(ref_units))
        for ru in ref_units:                        # a reference-sized unit lies inside one of the resident units
            j = max(i for (r, s), i in by_unit.items() if r == ru.record and s <= ru.start)
            u = units[j]
            d = ru.start - u.start
            ref_segs.append((int(run.seg_off[j]) + d, ru.seg_len, ru.count, int(run.out_off[j]) + d, row0 + len(ref_segs)))
        ref_elapsed, _, _ = run.timed(args.steps, 1, ref_segs, kernel_events=False)
        ref_batch = {"batch": REFERENCE_BATCH, "value": total_positions * args.steps / ref_elapsed, "unit": "positions/s",
                     "ms_per_step": ref_elapsed / args.steps * 1e3, "launches_per_step": len(ref_segs), "streams": len(run.streams)}
        if len(run.streams) > 1:                    # the same on one stream
            saved, run.streams = run.streams, run.streams[:1]
            one_elapsed, _, _ = run.timed(args.steps, 1, ref_segs, kernel_events=False)
            run.streams = saved
            ref_batch["value_one_stream"] = total_positions * args.steps / one_elapsed
        run.check_status()
    gather = run.final_gather()
    tallies, probe = run.counters()
    result = None
    if rank == 0:
        result = {
            "metric": f"genome positions/sec (min-unique-k search, {KMIN}:{KMAX})",
            "value": total_positions * args.steps / elapsed,
            "unit": "positions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl.desc + f", search-range {KMIN}:{KMAX}, both strands",
                       "records": len(wl.records), "positions": int(total_positions), "positions_per_gpu": run.my_positions,
                       "batch": args.batch, "segments_per_rank": len(run.segs),
                       "streams": len(run.streams) if len(run.segs) > 1 else 1,
                       "streams_output_identical_to_one_stream": run.overlap_identical,
                       "seed_length": run.info["seed_length"], "quad_core_length": run.info.get("quad_core_length", 0),
                       "quad_small_core_length": run.info.get("quad_small_core_length", 0),
                       "index_bytes_hbm": run.info["device_bytes"],
                       "parallelism": f"independent work units over {world} GPU(s), index replicated, no data-path collective"},
            "roofline": roofline_block(run, k0, tallies, wl.key),
            "pipeline": pipeline_block(run, k1, tallies, probe),
            "host": {"fasta_write_s": prep["fasta_write_s"], "index_build_s": prep["index_build_s"], "index_builder": args.index_builder,
                     "index_open_s": run.t_open, "sequence_upload_s": run.t_upload},
        }
        if ref_batch is not None:
            result["reference_batch"] = ref_batch
        if gather is not None:
            result["final_gather_ms"] = gather["ms"]      # one RCCL gather of all results to rank 0, outside `value`
            result["final_gather_bytes_per_rank"] = gather["bytes_per_rank"]
    if world == 1 and not args.no_cpu_baseline and args.config == "c2":
        gpu_out = run.d_out[:wl.lengths[0]].cpu().numpy()
        result["cpu_baseline"] = cpu_baseline(args, wl.record(0), gpu_out, KMIN, KMAX)
    else:
        v = run.verify_sample()
        if rank == 0:
            result["verify"] = v
    run.close()
    del run
    # ------------------------------------------------------------------ north star: ~3 Gbp at 20:200
    if not args.no_north_star and args.config == "c2":
        ns = north_star_workload(args)
        t_ns = time.time()
        nfa, nidx, nprep = prepare_index(args, ns, rank, barrier)
        nranges = parallel.interleaved_ranges(ns.total, world, 64 << 20)[rank] if world > 1 else [(0, ns.total)]
        # (one launch per record and range: the largest record has 249 M positions)
        nunits = parallel.units_for_ranges(ns.lengths, nranges, max(args.batch, 1 << 28), ns.krange[1])
        nrun = Run(args, ns, nidx, nunits, rank, world, dev, barrier, dist, rehearse, args.seed_length)
        for i in range(len(ns.records)):           # the device holds the units now
            if ns.lengths[i] > 150_000_000:
                ns.drop(i)
        n_steps, n_warm = min(args.steps, 10), min(args.warmup, 2)
        n_elapsed, nk0, nk1 = nrun.timed(n_steps, n_warm)
        nrun.check_status()
        n_total = nrun.sum_over_ranks(nrun.my_positions)
        n_gather = nrun.final_gather()
        n_tallies, n_probe = nrun.counters()
        n_verify = nrun.verify_sample()
        if rank == 0:
            block = {"workload": ns.desc, "positions": int(n_total), "n_gpus": world, "scaling": "strong (fixed genome, units dealt to the ranks in interleaved chunks)",
                     "steps": n_steps, "warmup": n_warm, "ms_per_step": n_elapsed / n_steps * 1e3, "value": n_total * n_steps / n_elapsed,
                     "unit": "positions/s", "launches_per_step_per_rank": len(nrun.segs),
                     "streams": len(nrun.streams) if len(nrun.segs) > 1 else 1, "streams_output_identical_to_one_stream": nrun.overlap_identical,
                     "index_bytes_hbm": nrun.info["device_bytes"], "bwt_rows": nrun.info["bwt_length"],
                     "roofline": roofline_block(nrun, nk0, n_tallies, ns.key), "pipeline": pipeline_block(nrun, nk1, n_tallies, n_probe),
                     "verify": n_verify,
                     "host": {"fasta_write_s": nprep["fasta_write_s"], "index_build_s": nprep["index_build_s"], "index_builder": args.index_builder,
                              "index_open_s": nrun.t_open, "sequence_upload_s": nrun.t_upload}}
            if n_gather is not None:
                block["final_gather_ms"] = n_gather["ms"]
            cb = result.get("cpu_baseline")
            if cb:
                block["cpu_baseline_extrapolated"] = {
                    "value": cb["value"], "unit": "positions/s", "cores": cb["cores"], "kind": "port",
                    "note": "EXTRAPOLATED: the rate the CPU port reaches on the 100 Mbp genome of configs[1] at the same 20:200 range "
                            "(620 LF steps per position whatever the genome; its rank blocks for this genome would be 31x larger, so the real "
                            "rate on these host cores is lower) -- SURVEY.md section 8(d) sanctions the prefix / extrapolation",
                    "gpu_over_cpu": block["value"] / cb["value"]}
            result["north_star"] = block
        if world == 1 and not args.no_end_to_end:
            e2e = end_to_end(args, ns, nfa, nidx, dev_index, nrun)
            result["end_to_end"] = e2e
        nrun.close()
        if rank == 0:
            log(f"[bench] north star block: {time.time() - t_ns:.1f}s")
    if rank == 0:
        result["bench_wall_s"] = time.time() - t_bench
        print(json.dumps(result), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
