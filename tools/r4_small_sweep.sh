#!/bin/bash
# round 4: thresholds of the heavy-read machinery at a 1 Mi-read batch (the command line's chunk)
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/ix
timeout 300 python3 bench.py --index-dir /tmp/ix --build-only 1 > /dev/null 2>&1
run() {
  env "$@" timeout 120 python3 - <<'PY'
import sys, os, time
sys.path.insert(0, "soap3-dp_amd/py"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch, bench, s3dp
n = int(os.environ.get("N_READS", 1048576))
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
t0 = time.perf_counter(); sm = cm = 0
for _ in range(5):
    b.align_single_allbest(4, 1000); b.sync(); tot, s_ms, k = b.timing(); sm += s_ms; cm += b.ce_timing()[0]
dt = (time.perf_counter() - t0) / 5
print("%s reads %d: %.2f M reads/s, %.1f ms per step (stepping %.1f, ce %.1f)" % (os.environ.get("TAG", ""), n, n / dt / 1e6, dt * 1e3, sm / 5, cm / 5), flush=True)
PY
}
run TAG=default
run TAG=idle16 S3DP_ADOPT_IDLE=16
run TAG=idle8 S3DP_ADOPT_IDLE=8
run TAG=idle4 S3DP_ADOPT_IDLE=4
run TAG=idle1 S3DP_ADOPT_IDLE=1
run TAG=10M_default N_READS=10000000
run TAG=10M_idle8 N_READS=10000000 S3DP_ADOPT_IDLE=8
run TAG=10M_idle1 N_READS=10000000 S3DP_ADOPT_IDLE=1
