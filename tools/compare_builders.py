"""Build the index of a BASELINE config genome with the host builder and with the device builder and compare
the two files byte for byte (GPU box; used once per round at full size).

    python tools/compare_builders.py c3            # 3.09 Gbp, 24 records
    python tools/compare_builders.py c5 --mbp 1000
"""
import argparse
import hashlib
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 26), b""):
            h.update(chunk)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--mbp", type=float, default=None)
    ap.add_argument("--workdir", default="/tmp/newmap_amd_cmp")
    a = ap.parse_args()
    from newmap_amd import synth
    from newmap_amd._c_newmap_generate_index import generate_fm_index
    wd = Path(a.workdir)
    wd.mkdir(parents=True, exist_ok=True)
    recs = synth.config_genome(a.config, a.mbp)
    fa = wd / "genome.fa"
    synth.write_fasta(fa, recs)
    out = {"config": a.config, "bases": int(sum(r.size for _, r in recs))}
    for name, kw in (("device", {"device": 0}), ("host", {})):
        t0 = time.time()
        generate_fm_index(str(fa), str(wd / f"{name}.awfmi"), 8, 12, **kw)
        out[f"{name}_build_s"] = round(time.time() - t0, 2)
        print(f"[compare] {name} builder: {out[f'{name}_build_s']} s", file=sys.stderr, flush=True)
        out[f"{name}_sha256"] = sha(wd / f"{name}.awfmi")
    out["identical"] = out["device_sha256"] == out["host_sha256"]
    print(json.dumps(out))
    if not out["identical"]:
        raise SystemExit("index files differ")


if __name__ == "__main__":
    main()
