#!/bin/bash
# Evidence run on the GPU box (everything lands under gpurun_out/r06f/; copy what is to be judged into profiles/): scripts/collect_profiles.sh
# bench line (live PMC traffic, matrix-core floor, CPU baseline, configs), rocprofv3 kernel stats of the default path, of BASELINE configs 2 / 4 / 5,
# of the stand-alone message-passing kernel and of the 8-way shard, emulated strong-scaling shards, the distributed step with one rank,
# SQ counters of the hop kernels, training step.
O=gpurun_out/r06f; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
python -m graphvqa_amd.build > $O/build.txt 2>&1                     # (the product library and the measurement build the floor leg loads: never a stale one --
python -m graphvqa_amd.build --probes >> $O/build.txt 2>&1           #  VERDICT r05 weak #6)
python bench.py > $O/bench_cfg3_n1.json 2> $O/bench.err
prof() {  # prof <tag> <command...>: rocprofv3 --kernel-trace --stats of the command, the kernel-stats CSV kept as $O/<tag>_kernel_stats.csv
  tag=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$tag -o ks -- "$@" > $R/$O/${tag}_under_rocprof.json 2> $R/$O/${tag}_rocprof.err )
  cp $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv 2>/dev/null; rm -rf $O/prof_$tag
}
prof bench_cfg3 python $R/bench.py --no-cpu-baseline --no-pmc --no-extras
prof mp_standalone python $R/scripts/bench_mp_only.py
prof cfg2 env CONFIG=2 FUSION=3 python $R/scripts/bench_hopagg.py
prof cfg4 python $R/scripts/bench_gine.py
prof cfg5 python $R/scripts/bench_lcgn_stages.py
prof shard8 python $R/bench.py --emulate-world 8 --no-cpu-baseline --no-pmc --no-extras
python scripts/bench_configs.py 2>/dev/null | tail -1 > $O/configs.json
python scripts/bench_lcgn_stages.py 2>/dev/null | grep "^{" > $O/lcgn_stages.jsonl
for f in 1 0; do GVQA_GINE_FUSED=$f python scripts/bench_gine.py 2>/dev/null | tail -1; done > $O/gine_fused_ab.jsonl
for pk in 1 0; do for f in 3 5 4 1; do GVQA_PACKED_GROUPS=$pk CONFIG=2 FUSION=$f python scripts/bench_hopagg.py 2>/dev/null | tail -1 | sed -e "s/^{/{\"packed_groups\": $pk, /"; done; done > $O/cfg2_packed_ab.jsonl
( GVQA_BENCH_ONE_DEVICE=1 GVQA_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 2>/dev/null | grep "\"metric\"" > $O/bench_n2_gloo_one_device.json )
for c in 3 2; do for f in 3 1 2 4 5 0; do CONFIG=$c FUSION=$f python scripts/bench_hopagg.py 2>/dev/null | tail -1; done; done > $O/hop_forms_ab.jsonl
for f in 3 1 2; do for n in 2 4 8 16; do GVQA_HOP_FUSION=$f python bench.py --emulate-world $n --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | tail -1; done; done > $O/emulated_shards.jsonl
GVQA_BENCH_FORCE_DIST=1 python bench.py --emulate-world 8 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | grep emulated_world > $O/emulated_shard8_rccl_1rank.json
GVQA_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | grep "\"metric\"" > $O/bench_cfg3_rccl_1rank.json
( export GVQA_LIB=graphvqa_amd/lib/probes/libgvqa_hip.so GVQA_HOPAGG_STAMPS=/tmp/ha_stamps.bin; for d in 300 512; do D=$d python scripts/probe_hopagg_seq.py 2>/dev/null | tail -1; done ) > $O/hopagg_seq_stamps.jsonl
python scripts/bench_hipgraph.py 2>/dev/null > $O/hipgraph.jsonl
python scripts/bench_train.py 2>/dev/null | tail -1 > $O/train_cfg3.json
CONFIG=2 TRAIN_ONLY=1 python scripts/bench_train.py 2>/dev/null | tail -1 > $O/train_cfg2.json
( cd /tmp && TRAIN_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tprof -o ks -- python $R/scripts/bench_train.py > /dev/null 2>&1 )
cp $(find $O/tprof -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv 2>/dev/null; rm -rf $O/tprof
# SQ counters of the hop kernels (separate --pmc passes, kernel trace only)
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA"; do
  tag=$(echo $pass | cut -d' ' -f1)
  ( cd /tmp && MODES=5,4,1,0 ROUNDS=1 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$O/pmc_$tag -o pmc -- python $R/scripts/bench_hop2.py > /dev/null 2>&1 )
  ( cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$O/pmcg_$tag -o pmc -- python $R/scripts/bench_gine.py > /dev/null 2>&1 )
  ( cd /tmp && CONFIG=2 FUSION=3 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/$O/pmc2_$tag -o pmc -- python $R/scripts/bench_hopagg.py > /dev/null 2>&1 )
done
python - > $O/pmc_hop_kernels.txt <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc*_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_gat_mp_tiled" in n or ("k_linear_split3<" in n and ", 2, 4, 2, 0, 2, 1>" in n) or "k_hopagg4" in n or "k_gine_mlp" in n:
            acc[n[:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-28s %14.0f  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
PY
rm -rf $O/pmc_SQ_* $O/pmcg_SQ_* $O/pmc2_SQ_*
ls -la $O
