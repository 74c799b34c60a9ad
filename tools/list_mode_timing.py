"""Throughput of list mode with one length (BASELINE configs[3]'s shape: fixed-k count mode, k = 36 and 100)
on a bench genome resident in HBM: the range-kernel route (default) against the list kernel.

    python tools/list_mode_timing.py [--config c2|c3|c5|hs] [--mbp N] [--k 36 100]      (hs: the human-shaped stand-in, synth.human_like_dna)
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--mbp", type=float, default=None)
    ap.add_argument("--k", type=int, nargs="+", default=[36, 100])
    ap.add_argument("--lists", nargs="*", default=["24,36,50,100"], help="comma-separated length lists timed as well")
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--batch", type=int, default=100_000_000, help="positions per launch")
    a = ap.parse_args()
    import torch
    from newmap_amd import parallel, synth
    from newmap_amd._c_newmap_generate_index import generate_fm_index
    from newmap_amd.engine import Index
    recs = synth.config_genome(a.config, a.mbp)
    wd = Path("/tmp/newmap_amd_list")
    wd.mkdir(exist_ok=True)
    fa = synth.write_fasta(wd / "genome.fa", recs)
    generate_fm_index(str(fa), str(wd / "genome.awfmi"), 8, 12, device=0)
    lengths = [int(r.size) for _, r in recs]
    off = np.concatenate(([0], np.cumsum(lengths)))
    n = int(off[-1])
    dev = torch.device("cuda", 0)
    d_seq = torch.empty(n, dtype=torch.uint8, device=dev)
    for (_, r), o in zip(recs, off[:-1]):
        d_seq[int(o):int(o) + r.size].copy_(torch.from_numpy(r))
    d_out = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_st = torch.zeros(16, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    res = {"config": a.config, "positions": n, "batch": a.batch}
    with Index(wd / "genome.awfmi", 0) as ix:
        for spec in [[k] for k in a.k] + [[int(x) for x in l.split(",")] for l in a.lists]:
            k = max(spec)
            tag = "k" + "_".join(str(x) for x in spec)
            units = parallel.units_for_slice(lengths, 0, n, a.batch, k)
            segs = [(int(off[u.record]) + u.start, u.seg_len, u.count) for u in units]
            outs = {}
            for name, via in (("range_kernels", True), ("list_kernel", False)):
                ix.set_list_via_range(via)

                def one_pass():
                    for p, seg_len, cnt in segs:
                        ix.fixed_k_segment_dev(d_seq.data_ptr() + p, seg_len, cnt, spec, True, 1, d_out.data_ptr() + p,
                                               d_st.data_ptr(), stream)
                one_pass()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.passes):
                    one_pass()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / a.passes
                outs[name] = d_out.cpu().numpy().copy()
                res[f"{tag}_{name}_positions_per_s"] = n / dt
            res[f"{tag}_identical"] = bool(np.array_equal(outs["range_kernels"], outs["list_kernel"]))
            res[f"{tag}_nonzero_fraction"] = float((outs["list_kernel"] != 0).mean())
        host = np.concatenate([r for _, r in recs])
        res["ambiguous_fraction"] = float(np.isin(host, np.frombuffer(b"ACGTacgt", np.uint8), invert=True).mean())
        res["lower_case_fraction"] = float(((host >= 97) & (host <= 122)).mean())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
