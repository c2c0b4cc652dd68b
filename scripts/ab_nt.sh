#!/bin/bash
# A/B of the hop kernel's row traffic with the non-temporal hint (GVQA_HA_NT builds under graphvqa_amd/lib/nt<mask>/) against the default
# library on ONE box (build the variants first, here: for m in 7 6 4; do python -m graphvqa_amd.build --variant nt$m GVQA_HA_NT=$m; done):
# parity of every variant first (aggregate-first tests + a short randomised sweep), then bench.py alternated.
O=gpurun_out/nt; mkdir -p $O; export TMPDIR=/tmp
V="${VARIANTS:-7 6 4}"
for v in $V; do
  L=graphvqa_amd/lib/nt$v/libgvqa_hip.so
  GVQA_LIB=$L timeout 400 python -m pytest tests/test_gpu_gat.py -m gpu -q -x -k "aggregate_first or default_rule or config3" 2>&1 | tail -2 > $O/pytest_nt$v.txt
  GVQA_LIB=$L timeout 300 python scripts/fuzz_hopagg.py 1 77 > $O/fuzz_nt$v.json 2>/dev/null
done
for r in 1 2 3; do
  python bench.py --no-extras --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 > $O/base_$r.json
  for v in $V; do GVQA_LIB=graphvqa_amd/lib/nt$v/libgvqa_hip.so python bench.py --no-extras --no-cpu-baseline --no-pmc 2>/dev/null | tail -1 > $O/nt${v}_$r.json; done
done
python bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/base_pmc.json
for v in $V; do GVQA_LIB=graphvqa_amd/lib/nt$v/libgvqa_hip.so python bench.py --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > $O/nt${v}_pmc.json; done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable"); continue
    if "ms_per_step" in d:
        print(f, round(d["ms_per_step"], 4), round(d["roofline"]["avg_launch_us"], 1), d["roofline"].get("traffic"))
    else:
        print(f, json.dumps(d)[:200])
PY
cat $O/pytest_nt*.txt
