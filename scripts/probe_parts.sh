export GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so
for d in 0 1 2 4 8 16 5 7 21 23; do GVQA_HOPAGG_DEBUG=$d GVQA_HOP_FUSION=6 python bench.py --emulate-world 8 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | grep emulated_world | python -c "import json,sys; d=json.load(sys.stdin); print('dbg $d', round(d['ms_per_step'],4), d['gpu_stage_ms_per_step']['proj'])"; done
