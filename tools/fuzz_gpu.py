"""Randomised A/B soak on the GPU: the default range path (sites on the quad table + gated repeat probes + sweep / resolve,
list modes routed through it; random cap of the group size; the > 2^31-row instantiations on every other index)
against the plain one-lane-per-position kernels without probes (and, half of the time, the same segments again through the
device-pointer entry dealt over 2..7 streams = the handle's lanes, incl. lanes changing hands), on genomes with tandem arrays, dispersed and
reverse-complement copies, N runs and soft-masked stretches, cut into segments at random batch sizes.

    python tools/fuzz_gpu.py [--rounds 40] [--seed 1]
"""
import argparse
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from newmap_amd import _lib  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

ALPHA = np.frombuffer(b"ACGT", np.uint8)
COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def genome(rng):
    recs = []
    for r in range(int(rng.integers(1, 4))):
        n = int(rng.integers(2_000, 400_000))
        s = bytearray(bytes(ALPHA[rng.integers(0, 4, n)]))
        for _ in range(int(rng.integers(0, 12))):
            kind = int(rng.integers(0, 7))
            a = int(rng.integers(0, n))
            m = int(min(n - a, rng.integers(1, 30_000)))
            if m <= 0:
                continue
            if kind == 0:                                   # tandem array
                unit = bytes(ALPHA[rng.integers(0, 4, int(rng.integers(1, 300)))])
                s[a:a + m] = (unit * (m // len(unit) + 1))[:m]
            elif kind == 1:                                 # dispersed copy
                b = int(rng.integers(0, n - m + 1))
                s[a:a + m] = s[b:b + m]
            elif kind == 2:                                 # reverse-complement copy
                b = int(rng.integers(0, n - m + 1))
                s[a:a + m] = bytes(s[b:b + m]).translate(COMP)[::-1]
            elif kind == 3:                                 # ambiguous run
                m = int(min(m, rng.integers(1, 300)))
                s[a:a + m] = bytes(rng.choice(np.frombuffer(b"NnRYKMSWBDHV", np.uint8), m))
            elif kind == 4:                                 # soft-masked stretch
                s[a:a + m] = bytes(s[a:a + m]).lower()
            else:                                           # diverged copy (a repeat family member), either strand: the ends of
                b = int(rng.integers(0, n - m + 1))         # neighbouring positions' least unique strings coincide in runs
                src = bytes(s[b:b + m]) if kind == 5 else bytes(s[b:b + m]).translate(COMP)[::-1]
                cp = np.frombuffer(src, np.uint8).copy()
                hit = np.flatnonzero(rng.random(m) < float(rng.choice([0.02, 0.05, 0.1, 0.2])))
                cp[hit] = ALPHA[rng.integers(0, 4, hit.size)]
                s[a:a + m] = cp.tobytes()
        recs.append((f"r{r}".encode(), bytes(s)))
    if len(recs) > 1 and rng.random() < 0.5:                # a copy across records
        src = recs[0][1]
        m = min(len(src), 5000)
        recs[-1] = (recs[-1][0], recs[-1][1] + src[:m])
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import torch
    from newmap_amd._c_newmap_generate_index import generate_fm_index
    from newmap_amd.engine import Index
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    checks = 0
    with tempfile.TemporaryDirectory() as td:
        for rnd in range(a.rounds):
            recs = genome(rng)
            fa, idx = Path(td) / "g.fa", Path(td) / "g.awfmi"
            fa.write_bytes(b"".join(b">" + rid + b"\n" + data + b"\n" for rid, data in recs))
            generate_fm_index(str(fa), str(idx), 8, 12, device=0 if rng.random() < 0.5 else None)
            mode = "auto" if rng.random() < 0.7 else "auto-small"
            with Index(idx, 0, mode) as fast, Index(idx, 0, mode) as plain:
                plain.set_kernel(1)
                plain.set_repeat_probes(False)
                plain.set_list_via_range(False)
                plain.set_lf2(False)                       # the plain side walks base by base
                fast.set_force_big(bool(rng.random() < 0.5))
                w = (fast.info()["quad_small_core_length"] or fast.info()["quad_core_length"]) + 4
                for _ in range(6):
                    fast.set_site_d(int(rng.choice([59, 59, 59, 0, 1, 3, 17])))
                    fast.set_repeat_probes(bool(rng.random() < 0.8))
                    fast.set_site_table(int(rng.choice([0, 0, 1, 2])))
                    fast.set_dictionary(bool(rng.random() < 0.7))
                    fast.set_lf2(bool(rng.random() < 0.8))
                    fast.set_sweep(bool(rng.random() < 0.8))
                    fast.set_lcp(bool(rng.random() < 0.8))
                    kmin = int(rng.choice([w, w + 1, 20, 24, 36, 60, 61, 62, 64, 100, 124, 125, 190, 252, 253, int(rng.integers(1, 200))]))
                    kmax = int(kmin + rng.choice([0, 1, 5, 40, 130, 231, int(rng.integers(0, 3000))]))
                    batch = int(rng.choice([1 << 30, 10_007, 65_536, int(rng.integers(500, 200_000))]))
                    lists = [[kmin], [kmin, kmax], sorted({kmin, (kmin + kmax) // 2, kmax}), [kmax, kmin]]
                    ks = lists[int(rng.integers(0, len(lists)))]
                    for rid, data in recs:
                        n = len(data)
                        whole = []
                        for p in range(0, n, batch):
                            cnt = min(batch, n - p)
                            seg = data[p:min(p + cnt + kmax - 1, n)]
                            x, ax = fast.min_unique_segment(seg, cnt, kmin, kmax)
                            y, ay = plain.min_unique_segment(seg, cnt, kmin, kmax)
                            assert ax == ay and np.array_equal(x, y), ("range", rnd, rid, kmin, kmax, batch, p, mode)
                            whole.append(x)
                            x, ax = fast.fixed_k_segment(seg, cnt, ks)
                            y, ay = plain.fixed_k_segment(seg, cnt, ks)
                            assert ax == ay and np.array_equal(x, y), ("list", rnd, rid, ks, batch, p, mode)
                            checks += 2
                        if rng.random() < 0.5:
                            # the same segments through the device-pointer entry, dealt over several streams (lanes of the handle)
                            whole = np.concatenate(whole)
                            eb = whole.dtype.itemsize
                            streams = [torch.cuda.Stream() for _ in range(int(rng.integers(2, 8)))]
                            seq_t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
                            out_t = torch.full((n * eb,), 0xEE, dtype=torch.uint8, device="cuda")
                            starts = list(range(0, n, batch))
                            st_t = torch.zeros((len(starts), _lib.NM_STATUS_WORDS), dtype=torch.int64, device="cuda")
                            torch.cuda.synchronize()
                            for j, p in enumerate(starts):
                                cnt = min(batch, n - p)
                                seg_len = min(p + cnt + kmax - 1, n) - p
                                fast.min_unique_segment_dev(seq_t.data_ptr() + p, seg_len, cnt, kmin, kmax, True, eb, out_t.data_ptr() + p * eb,
                                                            st_t.data_ptr() + 8 * _lib.NM_STATUS_WORDS * j, streams[int(rng.integers(0, len(streams)))].cuda_stream)
                            torch.cuda.synchronize()
                            got = out_t.cpu().numpy().view(whole.dtype)
                            assert np.array_equal(got, whole), ("lanes", rnd, rid, kmin, kmax, batch, len(streams), mode)
                            assert not st_t[:, 1].any().item()
                            checks += len(starts)
            if rnd % 5 == 4:
                print(f"[fuzz] round {rnd + 1}/{a.rounds}: {checks} segment comparisons identical, {time.time() - t0:.0f}s", flush=True)
    print(f"fuzz ok: {checks} segment comparisons identical in {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
