#!/bin/bash
# round 4: the sweep (k_sweep) against one walk per position (k_resolve) -- parity subset, soak, then the human-shaped 3.09 Gbp genome
set -eo pipefail
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "min_unique_equals_oracle or fixed_k_equals_oracle or human_shaped_stand_in or config5_tandem or repeat_probes_change or zero_count or fixtures" > gpurun_out/r4a/pytest_subset.log 2>&1 || { tail -30 gpurun_out/r4a/pytest_subset.log; exit 1; }
tail -3 gpurun_out/r4a/pytest_subset.log
timeout -k 10 600 python tools/fuzz_gpu.py --rounds 25 --seed 41 > gpurun_out/r4a/fuzz.log 2>&1 || { tail -30 gpurun_out/r4a/fuzz.log; exit 1; }
tail -2 gpurun_out/r4a/fuzz.log
for sw in ${SWEEPS:-1 0}; do
  NEWMAP_AMD_SWEEP=$sw timeout -k 10 900 python bench.py --config hs --no-cpu-baseline --no-end-to-end --no-configs1 > gpurun_out/r4a/hs_sweep$sw.json 2> gpurun_out/r4a/hs_sweep$sw.log || { tail -30 gpurun_out/r4a/hs_sweep$sw.log; exit 1; }
  python tools/show_value.py gpurun_out/r4a/hs_sweep$sw.json || true
done
