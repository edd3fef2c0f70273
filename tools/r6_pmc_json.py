#!/usr/bin/env python3
"""Builds profiles/r06_pmc.json from the text summaries tools/r6_pmc.sh leaves in gpurun_out/<tag>/ (pmc_*.txt, kernel_stats.csv).
usage: tools/r6_pmc_json.py <tag>   (run in the repo root, on the tree the counters were measured on: the kernel source hash is taken here)"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def parse(path):
    out = {}; cur = None
    for line in open(path):
        m = re.match(r"^(?:void )?(k_accumulate(?:_sparse)?<\d+, \d+>)\s+dispatches=(\d+)", line)
        if m: cur = out.setdefault(m.group(1), {"dispatches": int(m.group(2))}); continue
        m = re.match(r"^\s+(\S+)\s+([\d.]+) per dispatch", line)
        if m and cur is not None: cur[m.group(1)] = float(m.group(2))
    return out


def main():
    tag = sys.argv[1]; d = os.path.join(ROOT, "gpurun_out", tag)
    per = {}
    for f in ("pmc_fetch.txt", "pmc_sq.txt", "pmc_sq2.txt", "pmc_sq3.txt"):
        for k, v in parse(os.path.join(d, f)).items(): per.setdefault(k, {}).update(v)
    full = {k: v for k, v in per.items() if k.endswith(", 2>")}        # the full 1000-query launches take the two-word instantiation at config 4
    names = {"k_accumulate_sparse": [k for k in full if "sparse" in k][0], "k_accumulate": [k for k in full if "sparse" not in k][0]}
    counters = {short: {c: v for c, v in full[k].items() if c != "dispatches"} for short, k in names.items()}
    trace = {}
    for r in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
        m = re.match(r"^(?:void )?(k_accumulate(?:_sparse)?<\d+, 2>)", r["Name"])
        if m: trace[m.group(1)] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6}
    fetch = {short: counters[short]["FETCH_SIZE"] * 1024.0 * 2.0 for short in counters}
    pair = lambda c: sum(counters[s].get(c, 0.0) for s in counters)
    out = {
        "kernel": "Stage-1 accumulation = " + " + ".join(names.values()),
        "kernel_source": "infidex_amd/csrc/stage1.hip.inc + stage1_sparse.hip.inc + launch_acc of infidex_amd/csrc/infidex_hip.hip",
        "kernel_source_sha16": bench.kernel_sha16(),
        "hbm_read_bytes_per_launch": sum(fetch.values()),
        "hbm_read_bytes_per_launch_by_kernel": fetch,
        "source": f"rocprofv3 --pmc FETCH_SIZE (own pass, tools/r6_pmc.sh, GPU call {tag}), per-dispatch mean over the {full[names['k_accumulate']]['dispatches']} full-batch launches of each kernel; "
                  "FETCH_SIZE counts KiB, x2 = the gfx950 correction of MI355X_MICROARCH.md (64 B requests counted as 32 B)",
        "kernel_trace_same_call": trace,
        "counters_per_launch": counters,
        "sum_over_the_pair": {c: pair(c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU")},
        "round5_single_kernel_for_comparison": {"SQ_INSTS_VALU": 3029375938.2, "SQ_INSTS_SALU": 1496260401.1, "SQ_INSTS_LDS": 242661267.6, "FETCH_SIZE": 5331430.8, "avg_ms": 7.11},
    }
    json.dump(out, open(os.path.join(ROOT, "profiles", "r06_pmc.json"), "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("kernel_source_sha16", "hbm_read_bytes_per_launch", "kernel_trace_same_call", "sum_over_the_pair")}, indent=1))


if __name__ == "__main__":
    main()
