#!/bin/bash
# round 4: resident rate of the single-end ladder against the batch size, and the per-launch laps of a 1 Mi batch (where do the fixed 40 ms go?)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04j /tmp/ix
python3 bench.py --index-dir /tmp/ix --build-only 1 > /dev/null 2>&1
for n in ${S3DP_SMALL_SIZES:-1048576}; do
S3DP_VERBOSE=1 python3 - $n <<'PY'
import sys, os, time, json
sys.path.insert(0, "soap3-dp_amd/py"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch, bench, s3dp
n = int(sys.argv[1])
dev = torch.device("cuda:0")
codes, _ = bench.make_genome(3100000000, 1, dev, 0.5)
hi = s3dp.HostIndex.from_files("/tmp/ix/bench_repeat-rich_3100000000.index")
reads2d, *_ = bench.make_reads(codes, n, 100, 2026, dev)
wpq = s3dp.word_per_query(100)
q = bench.pack_on_device(reads2d, wpq); lens = np.full(n, 100, dtype=np.uint32)
del reads2d, codes; torch.cuda.empty_cache()
ix = s3dp.Index(hi, device=0)
b = s3dp.Batch(ix, n, wpq); b.upload(q, lens)
for _ in range(2): b.align_single_allbest(4, 1000); b.sync()
t0 = time.perf_counter(); sm = cm = dm = 0
for _ in range(5):
    b.align_single_allbest(4, 1000); b.sync(); tot, s_ms, k = b.timing(); sm += s_ms; dm += tot; cm += b.ce_timing()[0]
dt = (time.perf_counter() - t0) / 5
print("reads %d: %.2f M reads/s, %.1f ms per step (stepping %.1f, ce %.1f, device %.1f)" % (n, n / dt / 1e6, dt * 1e3, sm / 5, cm / 5, dm / 5), flush=True)
b.enable_counters(True); b.align_single_allbest(4, 1000); b.sync(); b.enable_counters(False)
PY
done
