"""Shared helpers for the test-suite (fixture loading, parameter dicts)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return meta, {k: z[k] for k in z.files if k != "meta"}


def t(a, dtype=None, device=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None and x.is_floating_point():
        x = x.to(dtype)
    return x if device is None else x.to(device)


def tparams(p, dtype=torch.float32, device=None):
    return {k: t(v, dtype, device) for k, v in p.items()}


def maxabs(a, b):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.from_numpy(np.asarray(b)).double()
    return float((a - b).abs().max()) if a.numel() else 0.0


def integration_md_stub():
    """The fenced ```python block of INTEGRATION.md section 2 (the ctypes stub a maintainer would copy), as source text."""
    import re
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 2. Binding the C ABI directly"):]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    assert m and "def gat_seq_forward" in m.group(1), "INTEGRATION.md section 2 lost its stub"
    return m.group(1)


def header_struct_members(name):
    """[(member, kind)] of `typedef struct <name> { ... } <name>;` in include/gvqa.h; kind in {'ptr','i32','i64','f32','size'}.
    Comments stripped; `T a, b;` declares two members of T; anything with a '*' is a pointer."""
    import re
    h = open(os.path.join(ROOT, "include", "gvqa.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    m = re.search(r"typedef struct %s\s*\{(.*?)\}\s*%s\s*;" % (name, name), h, re.S)
    assert m, name
    kinds = {"int32_t": "i32", "int64_t": "i64", "float": "f32", "size_t": "size"}
    out = []
    for decl in m.group(1).split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        mm = re.match(r"(const )?([A-Za-z_0-9]+)\s*(.*)", decl)
        base, rest = mm.group(2), mm.group(3)
        for item in rest.split(","):
            item = item.strip()
            arr = re.match(r"(\**)\s*([A-Za-z_0-9]+)(\[(\d+)\])?$", item)
            assert arr, decl
            kind = "ptr" if arr.group(1) else kinds[base]
            out.append((arr.group(2), kind, int(arr.group(4)) if arr.group(4) else 0))
    return out
