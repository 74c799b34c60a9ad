"""Child process of tests/test_rccl_gpu.py: ONE rank under the launcher environment of torch.distributed.run (RANK,
WORLD_SIZE = 1, MASTER_*), backend "nccl" = RCCL.  Everything newmap_amd.parallel sends through a collective runs here on
the GPU -- the world all-reduce, the gather of padded results on a device tensor, the all-reduces of the record
fingerprints and of the failure flag, and `newmap search` with NEWMAP_AMD_GATHER=1 -- and the files are compared with
those of the single-process driver.  Prints one JSON line."""
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch
    import torch.distributed as dist
    from newmap_amd import parallel, search as S
    from newmap_amd._c_newmap_generate_index import generate_fm_index
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)                                   # bench.py: `rccl_ranks`
    out["rccl_ranks"] = int(ones.item())
    local = (np.arange(100_003) % 251).astype(np.uint16)
    full = parallel.gather_to_root(local, local.size, 1, 0, device=dev)         # ONE dist.gather on device tensors
    out["gather_to_root_equal"] = bool(np.array_equal(full, local))
    parallel._raise_together(1, 0, None)                    # all-reduce (MAX) of the failure flag
    try:
        parallel._raise_together(1, 0, ValueError("x"))
        out["raise_together"] = False
    except ValueError:
        out["raise_together"] = True
    golden = ROOT / "tests" / "golden" / "genome.fa"
    with tempfile.TemporaryDirectory() as td:
        rng = np.random.default_rng(3)
        alpha = np.frombuffer(b"ACGT", np.uint8)
        body = bytes(alpha[rng.integers(0, 4, 30_000)])
        fa = Path(td) / "g.fa"
        fa.write_bytes(golden.read_bytes() + b">big\n" + body + b"\n>n\n" + body[100:400] + b"NNNN" + body[5000:5600].lower() + b"\n")
        idx = Path(td) / "g.awfmi"
        generate_fm_index(str(fa), str(idx), 8, 12)
        from newmap_amd.engine import cached_index
        ix = cached_index(idx, 0)
        lens, fps = ix.records()
        info = [(int(n_), int(fp), 1) for n_, fp in zip(lens, fps)]
        out["unverified_records"] = parallel.unverified_records(ix, info, 1)      # all-reduce (SUM) of the fingerprint limbs: indexed records, nothing to guard
        for tag, ks, binary in (("range", list(range(20, 201)), True), ("list", [24, 36], False)):
            a, b = Path(td) / f"{tag}_one", Path(td) / f"{tag}_dist"
            a.mkdir(); b.mkdir()
            cfg = dict(fasta_filepaths=[fa], fmindex_filepaths=[idx], kmer_lengths=ks, is_binary_search=binary, kmer_batch_size=7000)
            S.write_unique_counts(S.SearchConfig(output_directory=a, **cfg))
            os.environ["NEWMAP_AMD_GATHER"] = "1"           # gather on rank 0 instead of direct writes: dist.gather on the device
            parallel.write_unique_counts_distributed(S.SearchConfig(output_directory=b, **cfg))
            os.environ["NEWMAP_AMD_GATHER"] = "0"
            c = Path(td) / f"{tag}_shard"
            c.mkdir()
            parallel.write_unique_counts_distributed(S.SearchConfig(output_directory=c, **cfg))   # native shard + the two all-reduces
            fa_ = {p.name: p.read_bytes() for p in sorted(a.iterdir())}
            out[f"{tag}_files"] = len(fa_)
            out[f"{tag}_gather_equal"] = fa_ == {p.name: p.read_bytes() for p in sorted(b.iterdir())}
            out[f"{tag}_shard_equal"] = fa_ == {p.name: p.read_bytes() for p in sorted(c.iterdir())}
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
