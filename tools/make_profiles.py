#!/usr/bin/env python3
"""Turn the raw output of tools/gpu_job.sh (gpurun_out/<TAG>_*) into the tracked files under profiles/.

    python tools/make_profiles.py TAG [--round r2]

    <TAG>_trace_full  -> profiles/<round>_kernel_stats.txt         rocprofv3 --kernel-trace --stats of `python bench.py`
                         profiles/<round>_bench_line_profiled.json  the line that run printed
    <TAG>_full.log    -> profiles/<round>_bench_line.json           the same command without the profiler
    <TAG>_trace       -> profiles/<round>_kernel_stats_1batch.txt   one 1000-assembly batch per step (kernels run alone)
    <TAG>_pmc_*       -> profiles/<round>_pmc.txt, profiles/scan_pmc_<round>.json (what bench.py reports as roofline.traffic)
    tools/isa_cost.py -> profiles/fill_isa_cost_<round>.txt
"""
import argparse
import collections
import csv
import glob
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "gpurun_out"
PROF = ROOT / "profiles"


def short(name: str) -> str:
    if "rocprim" in name:
        return "rocprim segmented_radix_sort " + ("(block sort lambda)" if "lambda" in name else "")
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def kernel_stats(csv_path: Path, header: str) -> str:
    rows = list(csv.DictReader(open(csv_path)))
    lines = [f"# {header}", f"# source: {csv_path.relative_to(ROOT)} (rocprofv3 --kernel-trace --stats, csv)",
             f"{'kernel':<72} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>7}"]
    for r in rows:
        lines.append(f"{short(r['Name']):<72} {r['Calls']:>6} {float(r['TotalDurationNs']) / 1e6:>10.2f} "
                     f"{float(r['AverageNs']) / 1e3:>10.1f} {float(r['MinNs']) / 1e3:>10.1f} {float(r['MaxNs']) / 1e3:>10.1f} "
                     f"{float(r['Percentage']):>7.2f}")
    return "\n".join(lines) + "\n"


def bench_line(log: Path) -> dict | None:
    if not log.exists():
        return None
    for line in open(log):
        if line.startswith("{"):
            return json.loads(line)
    return None


def pmc(tag: str) -> dict:
    agg: dict = collections.defaultdict(lambda: collections.defaultdict(float))
    disp: dict = collections.defaultdict(lambda: collections.defaultdict(set))
    for d in sorted(glob.glob(str(OUT / f"{tag}_pmc_*"))):
        if not Path(d).is_dir():
            continue
        for f in glob.glob(f"{d}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k][r["Counter_Name"]].add((d, r["Dispatch_Id"]))  # (a counter may be collected in more than one pass)
    return {k: {c: (v, len(disp[k][c])) for c, v in cs.items()} for k, cs in agg.items()}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--round", default="r3")
    args = ap.parse_args()
    rd, tag = args.round, args.tag
    done = []
    # (the default command also runs `kaptive assembly` in a process of its own -- the e2e.cli_from_fasta leg --, which leaves a
    # second set of files: the bench's own is the one with the longest kernel trace)
    full = sorted(glob.glob(str(OUT / f"{tag}_trace_full/*/*kernel_stats.csv")),
                  key=lambda f: -Path(f.replace("kernel_stats", "kernel_trace")).stat().st_size)
    if full:
        (PROF / f"{rd}_kernel_stats.txt").write_text(kernel_stats(Path(full[0]), "python bench.py (defaults) under the profiler"))
        line = bench_line(OUT / f"{tag}_trace_full.log")
        if line:
            (PROF / f"{rd}_bench_line_profiled.json").write_text(json.dumps(line, indent=1) + "\n")
        done.append("kernel_stats")
    line = bench_line(OUT / f"{tag}_full.log")
    if line:
        (PROF / f"{rd}_bench_line.json").write_text(json.dumps(line, indent=1) + "\n")
        done.append("bench_line")
    one = glob.glob(str(OUT / f"{tag}_trace/*/*kernel_stats.csv"))
    if one:
        (PROF / f"{rd}_kernel_stats_1batch.txt").write_text(kernel_stats(
            Path(one[0]), "python bench.py --assemblies 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e: one batch per step, "
                          "so the kernels of a pass run with nothing beside them"))
        done.append("kernel_stats_1batch")
    counters = pmc(tag)
    if counters:
        lines = ["# rocprofv3 --pmc, separate passes over `python bench.py --assemblies 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e`",
                 "# (one batch per step: nothing overlaps).  Sums over the dispatches of a kernel; per launch = sum / dispatches.",
                 "# FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled for wide streaming reads (MI355X guide, HBM section).",
                 "# GRBM_GUI_ACTIVE sums the 8 XCDs: shader clock = GRBM_GUI_ACTIVE / 8 / dispatches / launch duration."]
        for k in sorted(counters):
            if not any(x in k for x in ("kp_sw", "kp_scan", "kp_expand", "rocprim", "kp_chain", "kp_occ", "kp_join", "kp_anchor", "kp_protein", "kp_reduce", "kp_hit")):
                continue
            lines.append(k)
            for c, (v, n) in sorted(counters[k].items()):
                lines.append(f"    {c:<24} {v:>16.6g}  over {n} dispatches  = {v / max(n, 1):.6g} per launch")
        (PROF / f"{rd}_pmc.txt").write_text("\n".join(lines) + "\n")
        scan_name = next((k for k in counters if k.startswith("kp_scan_dense_kernel<0")), None) or \
            next((k for k in counters if k.startswith("kp_scan_kernel")), None)
        scan = counters.get(scan_name)
        if scan and "FETCH_SIZE" in scan and "WRITE_SIZE" in scan:
            fetch = scan["FETCH_SIZE"][0] / scan["FETCH_SIZE"][1] * 1024 * 2
            write = scan["WRITE_SIZE"][0] / scan["WRITE_SIZE"][1] * 1024
            js = {"workload": {"db": "kpsc", "batch": 1000, "length": 5.0e6},
                  "traffic_bytes_per_launch": fetch + write, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                  "source": f"profiles/{rd}_pmc.txt: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x 2 "
                            f"(gfx950 note of the MI355X guide), averages over the dispatches of {scan_name}"}
            if "TCC_REQ_sum" in scan:
                js["tcc_req_per_launch"] = scan["TCC_REQ_sum"][0] / scan["TCC_REQ_sum"][1]
                js["tcc_hit_per_launch"] = scan["TCC_HIT_sum"][0] / scan["TCC_HIT_sum"][1]
            (PROF / f"scan_pmc_{rd}.json").write_text(json.dumps(js, indent=1) + "\n")
        done.append("pmc")
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "isa_cost.py"), "kp_sw_kernel", "--marker", "v_pk_max_u16,v_pk_maximum3_f16", "--top", "1"], capture_output=True, text=True)
    if r.returncode == 0:
        (PROF / f"fill_isa_cost_{rd}.txt").write_text(
            "# python tools/isa_cost.py kp_sw_kernel --top 1: the 8-step body of the 16-diagonal class (two tasks per register:\n"
            "# 8 steps x 64 lanes x 4 cells x 2 tasks = 4096 cells per wave), priced with the issue costs measured by\n"
            "# tools/microbench/valu_rate*.hip (profiles/valu_rate*_r2.txt; v_pk_maximum3_f16: profiles/pk_max3_r3.txt)\n" + r.stdout)
        done.append("isa_cost")
    print("wrote:", ", ".join(done))
    return 0


if __name__ == "__main__":
    sys.exit(main())
