#!/bin/bash
# Evidence run on the GPU box (everything lands under gpurun_out/r/; copy what is to be judged into profiles/): scripts/collect_profiles.sh
# bench line (live PMC traffic, comparison legs, DVFS probe, CPU baseline), rocprofv3 kernel stats of the default path, BASELINE
# configs, emulated strong-scaling shards, the distributed step with one rank, SQ counters of the hop kernel, phase stamps of
# the persistent hop kernel (measurement build), training step.
O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
python bench.py > $O/bench_cfg3_n1.json 2> $O/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o ks -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --no-extras > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err )
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_cfg3_kernel_stats.csv 2>/dev/null
python scripts/bench_configs.py 2>/dev/null | tail -1 > $O/configs.json
python scripts/bench_lcgn_stages.py 2>/dev/null | grep "^{" > $O/lcgn_stages.jsonl
( GVQA_BENCH_ONE_DEVICE=1 GVQA_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 2>/dev/null | grep "\"metric\"" > $O/bench_n2_gloo_one_device.json )
# the three chained / fused hop forms on this box, config 3 and config 2 (hop kernel us, forward wall ms, stage split)
for c in 3 2; do for f in 3 1 2 4 5 0; do CONFIG=$c FUSION=$f python scripts/bench_hopagg.py 2>/dev/null | tail -1; done; done > $O/hop_forms_ab.jsonl
( export GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so; for d in 0 1 2 4 32; do GVQA_HOPAGG_DEBUG=$d python scripts/bench_hopagg.py 2>/dev/null | tail -1; done ) > $O/hopagg_loop_parts.jsonl
for f in 3 1 2; do for n in 2 4 8 16; do GVQA_HOP_FUSION=$f python bench.py --emulate-world $n --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -1; done; done > $O/emulated_shards.jsonl
GVQA_BENCH_FORCE_DIST=1 python bench.py --emulate-world 8 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | grep emulated_world > $O/emulated_shard8_rccl_1rank.json
GVQA_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | grep "\"metric\"" > $O/bench_cfg3_rccl_1rank.json
for f in 4 1 2; do MODES=$f ROUNDS=2 python scripts/bench_hop2.py 2>/dev/null | grep hop_kernel; done > $O/hop_kernels_ab.jsonl
for z in "" 1; do ZERO=$z MODES=1,2 ROUNDS=1 python scripts/bench_hop2.py 2>/dev/null | grep hop_kernel | sed -e "s/^{/{\"zero_operands\": \"$z\", /"; done > $O/dvfs_zero_operands.jsonl
for h in 1 4; do HOP=$h python scripts/probe_hop2.py 2>/dev/null | head -2; done > $O/hop2_phase_stamps.jsonl
python scripts/probe_hop2_loop.py 2>/dev/null | grep workgroups_per_cu > $O/hop2_loop_parts.jsonl
python scripts/bench_train.py 2>/dev/null | tail -1 > $O/train_cfg3.json
python scripts/bench_skinny.py 2>/dev/null | grep "^{" > $O/train_products.jsonl
CONFIG=2 TRAIN_ONLY=1 python scripts/bench_train.py 2>/dev/null | tail -1 > $O/train_cfg2.json
python scripts/bench_tn.py 2>/dev/null | grep "^{" > $O/train_tn_direct.jsonl            # the projection's gradient products, direct vs packed operands
bash scripts/ab_train_parts.sh 2>/dev/null > $O/train_parts_ab.txt                         # the training step with each round-5 change switched off
python scripts/prof_train_ops.py 2>/dev/null | grep -v Warning | tail -34 > $O/train_aten_ops.txt   # what torch still runs inside the step
( cd /tmp && TRAIN_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/tprof -o ks -- python $GRAFT_REPO_ROOT/scripts/bench_train.py > /dev/null 2>&1 )
cp $(find $O/tprof -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv 2>/dev/null
# SQ counters of the hop kernels (separate --pmc passes, kernel trace only)
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA"; do
  tag=$(echo $pass | cut -d' ' -f1)
  ( cd /tmp && MODES=5,4,1,2,0 ROUNDS=1 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/scripts/bench_hop2.py > /dev/null 2>&1 )
done
python - > $O/pmc_hop_kernels.txt <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if ("k_hop2<" in n) or "k_gat_mp_tiled" in n or "k_gat_alpha_groups_packed" in n or ("k_linear_split3<" in n and ", 2, 4, 2, 0, 2, 1>" in n) or "k_hopagg4" in n:
            acc[n[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-28s %14.0f  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
PY
rm -rf $O/prof $O/tprof $O/pmc_SQ_*
# evidence hygiene: a wall figure more than 3x its stage sum is an outlier of the run, not a result -- fail loudly
python - <<PY
import json, sys
bad = []
try:
    c = json.load(open("$O/configs.json"))
    for k, v in c.items():
        for sub in ([v] + [v[x] for x in ("fused", "unfused") if isinstance(v, dict) and x in v]):
            if not isinstance(sub, dict) or "stage_ms" not in sub: continue
            wall = sub.get("ms_per_forward", sub.get("ms_per_5_convs_plus_module"))
            ssum = sum(sub["stage_ms"].values())
            if wall and ssum and wall > 3 * ssum: bad.append((k, wall, ssum))
except Exception as e:
    bad.append(("configs.json unreadable", str(e), 0))
print("wall-vs-stage-sum check:", "OK" if not bad else bad)
sys.exit(1 if bad else 0)
PY
ls -la $O
