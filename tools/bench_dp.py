#!/usr/bin/env python3
"""GCUPS of the semi-global DP kernels (config C3 shapes: 150 bp reads against 650-column default-DP windows and
224-column deep-DP windows).  Prints one JSON line per shape.  The DP is integer-VALU/LDS bound; its HBM traffic is the
2.1 B per cell of traceback table (H as int16 + one ballot word per 64 cells).
S3DP_BENCH_DP_PROBLEMS=n: problems per shape (counter passes under rocprofv3 --pmc use 4096: every dispatch is replayed per counter)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "soap3-dp_amd", "py"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import s3dp
import s3o
from dp_cases import make_problems

SC = (1, -2, -3, -1)
NP = int(os.environ.get("S3DP_BENCH_DP_PROBLEMS", "0"))
for (max_read, max_dna, n, scheme) in ((152, 650, NP or 32768, 1), (152, 224, NP or 65536, 1), (152, 650, NP or 32768, 2)):
    P = make_problems(n, max_read, max_dna, seed=7, read_len_range=(150, 150), clip=True, anchors=False)
    P["dna_len"][:] = max_dna - 2
    al = s3dp.DpAligner(n, max_read, max_dna, scheme=scheme, scores=SC)
    args = (P["dna"], P["dna_len"], P["reads"], P["read_len"], P["cutoffs"])
    kw = dict(clip_lt=P["clip_lt"], clip_rt=P["clip_rt"])
    al.align(*args, **kw)
    ms = []
    for _ in range(3):
        al.align(*args, **kw)
        ms.append(al.timing())
    cells = float((P["read_len"].astype(np.int64) * P["dna_len"].astype(np.int64)).sum()) * (2 if al.scheme == 2 else 1)
    # CPU restatement on a sample
    m = 256
    t0 = time.perf_counter()
    s3o.dp_align(P["dna"], P["dna_len"][:m], max_dna, al.max_table, P["reads"], P["read_len"][:m], max_read, P["cutoffs"][:m], scheme=al.scheme, scores=SC,
                 clip_lt=P["clip_lt"][:m], clip_rt=P["clip_rt"][:m])
    cpu = time.perf_counter() - t0
    cpu_cells = float((P["read_len"][:m].astype(np.int64) * P["dna_len"][:m].astype(np.int64)).sum()) * (2 if al.scheme == 2 else 1)
    print(json.dumps({"metric": "GCUPS semi-global DP", "value": cells / (min(ms) / 1e3) / 1e9, "unit": "GCUPS", "problems": n, "read_len": 150,
                      "window": int(max_dna - 2), "scheme": al.scheme, "kernel_ms": min(ms), "table_bytes_per_cell": 2.125, "table_bytes_GBs": 2.125 * cells / (2 if al.scheme == 2 else 1) / (min(ms) / 1e3) / 1e9,
                      "cpu_oracle_GCUPS_1thread": cpu_cells / cpu / 1e9}))
    al.close()
