#!/usr/bin/env python3
"""Turns gpurun_out/prof_r05/ (tools/prof_round5.sh) into the committed artefacts under profiles/: r05_kernel_stats.csv, r05_pmc.json (counters of the full-batch
k_accumulate instantiation + the sha256 prefix of the kernel source AND its launch code, bench.kernel_sha16), the bench lines, and a markdown kernel table on stdout."""
import csv, json, os, re, shutil, sys

here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(here, "gpurun_out", "prof_r05"); P = os.path.join(here, "profiles")


def counters(path, kernel):
    out = {}; cur = None
    if not os.path.exists(path):
        return out
    for line in open(path):
        if line.startswith("void ") or (line and not line.startswith(" ")):
            cur = line.split("  dispatches=")[0].strip(); n = int(line.split("dispatches=")[1]) if "dispatches=" in line else 0
            if cur == kernel: out["_dispatches"] = n
        elif cur == kernel:
            m = re.match(r"\s+(\S+)\s+([0-9.]+) per dispatch", line)
            if m: out[m.group(1)] = float(m.group(2))
    return out


def main():
    kernel = "void k_accumulate<8192, 2>"
    shutil.copy(os.path.join(O, "kernel_stats.csv"), os.path.join(P, "r05_kernel_stats.csv"))
    c = {}
    for f in ("pmc_fetch.txt", "pmc_tcc.txt", "pmc_sq.txt", "pmc_sq2.txt"):
        c.update(counters(os.path.join(O, f), kernel))
    sha = open(os.path.join(O, "kernel_sha16.txt")).read().strip()
    n = int(c.pop("_dispatches", 0))
    pmc = {"kernel": kernel.replace("void ", ""), "kernel_source": "infidex_amd/csrc/stage1.hip.inc + launch_acc of infidex_amd/csrc/infidex_hip.hip", "kernel_source_sha16": sha,
           "hbm_read_bytes_per_launch": c["FETCH_SIZE"] * 1024 * 2,
           "source": f"rocprofv3 --pmc FETCH_SIZE (own pass, tools/prof_round5.sh), per-dispatch mean over the {n} full-batch launches; FETCH_SIZE counts KiB, "
                     "x2 = the gfx950 correction of MI355X_MICROARCH.md (64 B requests counted as 32 B)",
           "counters_per_launch": c}
    rep = {}
    for f in ("pmc_sq_replay.txt", "pmc_sq2_replay.txt"):
        path = os.path.join(O, f)
        if not os.path.exists(path):
            continue
        cur = None
        for line in open(path):
            if "dispatches=" in line:
                cur = line.split("  dispatches=")[0].strip(); rep.setdefault(cur, {})["_dispatches"] = int(line.split("dispatches=")[1])
            elif cur:
                m = re.match(r"\s+(\S+)\s+([0-9.]+) per dispatch", line)
                if m: rep[cur][m.group(1)] = float(m.group(2))
    pmc["replay_kernels_counters_per_dispatch"] = rep
    pmc["replay_note"] = "per-dispatch means over ALL dispatches of a kernel in the run, the single-query launches of the latency probe included (they lower the means)"
    json.dump(pmc, open(os.path.join(P, "r05_pmc.json"), "w"), indent=1)
    if os.path.exists(os.path.join(O, "kernel_stats_cfg3.csv")):
        shutil.copy(os.path.join(O, "kernel_stats_cfg3.csv"), os.path.join(P, "r05_kernel_stats_cfg3.csv"))
    for extra in ("gputest.log", "smoke.log"):
        if os.path.exists(os.path.join(O, extra)):
            shutil.copy(os.path.join(O, extra), os.path.join(P, "r05_" + extra.replace(".log", ".txt")))
    for name in ("bench_20steps", "bench_256steps", "bench_20steps_sessions3", "bench_20steps_sessions5", "bench_cfg2", "bench_cfg3", "bench_cfg5", "bench_sharded_w1"):
        line = open(os.path.join(O, name + ".json")).read().strip().splitlines()[-1]
        open(os.path.join(P, "r05_" + name + ".json"), "w").write(line + "\n")
    line = open(os.path.join(O, "kt.json")).read().strip().splitlines()[-1]
    open(os.path.join(P, "r05_bench_under_kernel_trace.json"), "w").write(line + "\n")
    print("| kernel | calls | total ms | avg ms | max ms | % |\n|---|---|---|---|---|---|")
    for r in list(csv.DictReader(open(os.path.join(O, "kernel_stats.csv"))))[:22]:
        print(f"| `{r['Name'].split('(')[0].replace('void ', '')}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e6:.4f} | {int(r['MaxNs']) / 1e6:.3f} | {float(r['Percentage']):.2f} |")
    print(json.dumps(c, indent=1))
    for name in ("bench_20steps", "bench_256steps", "bench_20steps_sessions3", "bench_20steps_sessions5", "bench_cfg2", "bench_cfg3", "bench_cfg5", "bench_sharded_w1"):
        d = json.loads(open(os.path.join(P, "r05_" + name + ".json")).read())
        print(name, round(d["value"]), round(d["ms_per_step"], 2), "p50", round(d["p50_batch_latency_ms"], 1), "p95", round(d["p95_batch_latency_ms"], 1), d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic"),
              (d.get("cpu_baseline") or {}).get("identical_topk_sets"), [(x["kernel"], round(x["avg_launch_ms"], 2), round(x["frac"], 4)) for x in d.get("roofline_by_kernel", [])])


if __name__ == "__main__":
    main()
