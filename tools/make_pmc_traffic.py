#!/usr/bin/env python3
"""profiles/pmc_traffic.json from rocprofv3 --pmc FETCH_SIZE counter files of `bench.py --steps 1 --warmup 0` runs (counted ladders, then one timed
ladder: the LAST four dispatches of every role -- one per phase of the ladder -- are the timed step).  FETCH_SIZE comes in kilobytes (MI355X_MICROARCH.md); 64-byte requests.
    tools/make_pmc_traffic.py <commit> <repeat_frac>=<counter_collection.csv> ...   > profiles/pmc_traffic.json"""
import collections
import csv
import json
import sys

KERNELS = ("firstKernel", "unitKernel", "ceKernel")
PHASES = 4                                   # search passes of the bench ladder (mismatch budgets 1 .. 4)


def role_of(name):
    """first / trunk (the two roles of s3::firstKernel: its last template argument) / unit / ce"""
    if "unitKernel" in name:
        return "unit"
    if "s3::ceKernel" in name:
        return "ce"
    if "firstKernel" in name:
        args = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",") if "<" in name else []
        return "trunk" if len(args) >= 3 and args[2] == "true" else "first"
    return None


def fetch_of_last_ladder(path):
    per, roles = collections.OrderedDict(), collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        k = next((k for k in KERNELS if ("s3::" + k) in r["Kernel_Name"]), None)
        if k is None:
            continue
        per.setdefault(k, collections.OrderedDict()).setdefault(r["Dispatch_Id"], 0.0)
        per[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
        ro = role_of(r["Kernel_Name"])
        roles.setdefault(ro, collections.OrderedDict()).setdefault(r["Dispatch_Id"], 0.0)
        roles[ro][r["Dispatch_Id"]] += float(r["Counter_Value"])
    out = {}
    by_role = {}
    for k, d in roles.items():
        last = list(d.values())[-PHASES:]
        by_role[k] = {"launches": len(last), "bytes": sum(last) * 1024.0}
    for k in per:
        rs = [r for r in by_role if (r in ("first", "trunk")) == (k == "firstKernel") and (r == "unit") == (k == "unitKernel") and (r == "ce") == (k == "ceKernel")]
        out[k] = {"launches": sum(by_role[r]["launches"] for r in rs), "bytes": sum(by_role[r]["bytes"] for r in rs)}
    return out, by_role


def main():
    commit = sys.argv[1]
    workloads = []
    for arg in sys.argv[2:]:
        frac, path = arg.split("=", 1)
        f, by_role = fetch_of_last_ladder(path)
        workloads.append({"genome_bp": 3100000000, "reads_per_gpu": 10000000, "read_len": 100, "max_mismatch": 4, "max_output_per_read": 1000,
                          "repeat_frac": float(frac), "fetch_bytes_per_step": sum(v["bytes"] for v in f.values()), "launches_per_step": 4,
                          "per_kernel": f, "per_role": by_role})
    print(json.dumps({"_comment": "HBM bytes fetched by the search kernels (s3::firstKernel: first pass + trunk pass, s3::unitKernel: unit pass, s3::ceKernel: "
                                  "check-and-extend) in one timed ladder of bench.py, rocprofv3 --pmc FETCH_SIZE in its own pass (KB x 1024; 64-byte requests), "
                                  "default configuration; tools/run_profiles.sh + tools/make_pmc_traffic.py",
                      "commit": commit, "workloads": workloads}, indent=1))


if __name__ == "__main__":
    main()
