"""every S3DP_* environment switch the product reads is listed in INTEGRATION.md section 6 (one was not), and the section lists none that is gone"""
import os
import re

from conftest import ROOT


def read_switches():
    found = set()
    for sub in ("csrc", "host", os.path.join("py", "s3dp")):
        d = os.path.join(ROOT, "soap3-dp_amd", sub)
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".h", ".cpp", ".py")):
                txt = open(os.path.join(d, f)).read()
                found |= set(re.findall(r'(?:getenv|envSize|environ\.get|environ\[)\s*\(?\s*"(S3DP_[A-Z0-9_]+)"', txt))
    return found


def test_switches_are_documented():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 6. Environment switches"):doc.index("## 7.")]
    listed = set(re.findall(r"`(S3DP_[A-Z0-9_]+)", sec))
    read = read_switches()
    assert len(read) > 30
    assert not (read - listed), "read by the product, missing in INTEGRATION.md section 6: %s" % sorted(read - listed)
    gone = listed - read - {"S3DP_BENCH_SHARE_GPU", "S3DP_BENCH_FULL"}                 # (bench.py's own)
    assert not gone, "listed in INTEGRATION.md section 6, read nowhere: %s" % sorted(gone)
    for k in ("S3DP_PRUNE_UNITS", "S3DP_LONG_DEAL", "S3DP_SORT_READS", "S3DP_CHUNK_BIG", "S3DP_DP_PRIORITY"):      # pruned in round 6
        assert k not in read
