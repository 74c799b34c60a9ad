"""Phase timings of nm_index_open in a fresh process (NEWMAP_AMD_VERBOSE=1 prints them on stderr).
    python tools/open_timing.py INDEX [QUAD_M]"""
import os
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if len(sys.argv) > 2:
    os.environ["NEWMAP_AMD_QUAD_M"] = sys.argv[2]
os.environ["NEWMAP_AMD_VERBOSE"] = "1"
from newmap_amd.engine import Index
t0 = time.time()
ix = Index(sys.argv[1], 0)
print(f"open total {time.time() - t0:.3f}s quad_m={ix.info()['quad_core_length']} bytes={ix.info()['device_bytes'] / 1e9:.1f} GB", flush=True)
t0 = time.time()
ix.close()
print(f"close {time.time() - t0:.3f}s", flush=True)
