#!/bin/bash
# strong-scaling shards of the config-3 batch: the default rule (aggregate-first with column parts up to half a round of row groups) against the
# 8-wave kernel (GVQA_HOP_FUSION=1) and the column parts forced (6): scripts/ab_parts.sh
for w in 8 16 4 32; do for f in 3 1 6; do GVQA_HOP_FUSION=$f python bench.py --emulate-world $w --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | grep emulated_world | python -c "import json,sys; d=json.load(sys.stdin); print('world $w fusion $f graphs', d['graphs'], round(d['ms_per_step'],4), d['gpu_stage_ms_per_step'])"; done; done
GVQA_BENCH_FORCE_DIST=1 python bench.py --emulate-world 8 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | grep emulated_world | python -c "import json,sys; d=json.load(sys.stdin); print('world 8 default + segment mean + 1-rank RCCL all-gather', round(d['ms_per_step'],4))"
