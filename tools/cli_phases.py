"""Where a one-shot `newmap search` of a bench genome spends its wall time (GPU box):
phases of nm_index_open (NEWMAP_AMD_VERBOSE) and of the native driver, from inside one process."""
import os
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
t_start = time.time()
os.environ["NEWMAP_AMD_VERBOSE"] = "1"
os.environ.setdefault("NEWMAP_AMD_SEED_LENGTH", "auto-small")
from newmap_amd import synth                                              # noqa: E402
from newmap_amd._c_newmap_generate_index import generate_fm_index         # noqa: E402
from newmap_amd.engine import cached_index                                # noqa: E402
from newmap_amd.search import SearchConfig, write_unique_counts           # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
wd = Path("/tmp/newmap_amd_cli")
wd.mkdir(exist_ok=True)
recs = synth.config_genome(cfg, None)
fa = synth.write_fasta(wd / "g.fa", recs)
generate_fm_index(str(fa), str(wd / "g.awfmi"), 8, 12, device=0)
(kmin, kmax) = {"c2": (20, 200), "c3": (24, 150), "c5": (20, 255)}[cfg]
out = wd / "out"
out.mkdir(exist_ok=True)
print(f"[phases] imports + genome + index build done at {time.time() - t_start:.2f}s", flush=True)
t0 = time.time()
ix = cached_index(str(wd / "g.awfmi"), 0)
print(f"[phases] index open: {time.time() - t0:.2f}s", flush=True)
t0 = time.time()
write_unique_counts(SearchConfig(fasta_filepaths=[str(fa)], fmindex_filepaths=[str(wd / "g.awfmi")], kmer_lengths=list(range(kmin, kmax + 1)),
                                 is_binary_search=True, kmer_batch_size=10_000_000, output_directory=out,
                                 use_reverse_complement=True))
print(f"[phases] write_unique_counts (FASTA in -> files out, index already open): {time.time() - t0:.2f}s", flush=True)
