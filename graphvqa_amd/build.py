"""Builds libgvqa_hip.so (HIP, gfx950) and, as test infrastructure, nothing else.

    python -m graphvqa_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is written in-tree (graphvqa_amd/lib/) so that it
travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgvqa_hip.so")
SOURCES = ["capi.hip", "graph.hip", "gemm.hip", "gemm_bf16.hip", "split3.hip", "hop2.hip", "hopagg.hip", "gat.hip", "gat_bwd.hip", "bn_train.hip", "variants.hip", "gine_mlp.hip", "lcgn.hip", "head.hip", "encoder.hip", "collate.hip", "train.hip", "tn_direct.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# per-file additions.  hopagg.hip: no SLP vectorisation -- its K step interleaves scalar fp32 FMAs with MFMAs, and packed fp32 math
# (v_pk_fma_f32, what the SLP pass makes of adjacent FMAs) costs the matrix-core stream more than two plain FMAs (MI355X_MICROARCH.md)
EXTRA_FLAGS = {"hopagg.hip": ["-fno-slp-vectorize"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _resource_usage(stderr: str) -> dict:
    """hipcc -Rpass-analysis=kernel-resource-usage -> {mangled kernel name: {"VGPRs": .., "AGPRs": .., "ScratchSize": .., "Occupancy": ..,
    "SGPRs Spill": .., "VGPRs Spill": .., "LDS Size": ..}} (device pass only)."""
    out, cur = {}, None
    for line in stderr.splitlines():
        if "kernel-resource-usage" not in line or "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].rsplit("[-Rpass", 1)[0].strip()
        if body.startswith("Function Name:"):
            cur = out.setdefault(body.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            k = k.split("[")[0].strip()
            v = v.strip()
            cur[k] = int(v) if v.lstrip("-").isdigit() else v
    return out


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, probes: bool = False) -> str:
    """probes=True: the measurement build (-DGVQA_PROBES: in-kernel phase stamps and ablation switches that the product
    library does not contain) -> lib/probes/libgvqa_hip.so, loaded by scripts/ through GVQA_LIB."""
    if probes:
        return _build(os.path.join(LIBDIR, "probes"), FLAGS + ["-DGVQA_PROBES"], force, verbose)
    return _build(LIBDIR, FLAGS, force, verbose)


def build_variant(name: str, defines, force: bool = False, verbose: bool = False) -> str:
    """An A/B build with extra -D switches -> lib/<name>/libgvqa_hip.so (loaded through GVQA_LIB; scripts/ab_nt.sh uses
    `--variant nt6 GVQA_HA_NT=6` and friends).  Never loaded by the product path."""
    return _build(os.path.join(LIBDIR, name), FLAGS + ["-D" + d for d in defines], force, verbose)


def _build(LIBDIR: str, FLAGS, force: bool, verbose: bool) -> str:
    LIB = os.path.join(LIBDIR, "libgvqa_hip.so")
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_tile.h"),
               os.path.join(os.path.dirname(HERE), "include", "gvqa.h")]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers) or not os.path.exists(obj.replace(".o", ".res.json")):
            jobs.append([HIPCC, *FLAGS, *EXTRA_FLAGS.get(s, []), "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if "-c" in cmd:         # the compiler's per-kernel resource remarks -> <object>.res.json (registers, scratch, LDS: tests/test_host.py checks them)
            with open(cmd[cmd.index("-o") + 1].replace(".o", ".res.json"), "w") as f:
                json.dump(_resource_usage(r.stderr), f, indent=0, sort_keys=True)
        if verbose:
            rest = "\n".join(l for l in r.stderr.splitlines() if "kernel-resource-usage" not in l)
            if rest.strip():
                print(rest, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    merged = {}
    for o in objs:
        with open(o.replace(".o", ".res.json")) as f:
            merged.update(json.load(f))
    with open(os.path.join(LIBDIR, "kernel_resources.json"), "w") as f:
        json.dump(merged, f, indent=0, sort_keys=True)
    relink = bool(jobs) or force or _stale(LIB, objs + [os.path.join(CSRC, "exports.map")])
    if relink:
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl",
             "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", LIB])
    # what this call did, for whoever records the build check (one line on stdout, always)
    print(f"build_mode: {'recompiled ' + str(len(jobs)) + ' of ' + str(len(SOURCES)) + ' objects' if jobs else 'reused all ' + str(len(SOURCES)) + ' objects'}"
          f"{', relinked' if relink else ', library up to date'} -> {os.path.relpath(LIB, os.path.dirname(HERE))}", flush=True)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:          # python -m graphvqa_amd.build --variant nt6 GVQA_HA_NT=6
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if not a.startswith("--")], force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True, probes="--probes" in sys.argv))
