#!/usr/bin/env python
"""Condenses rocprofv3 output directories into the tables kept under profiles/.

  python scripts/summarize_rocprof.py <stats_dir> <pmc_sq_dir> <pmc_fetch_dir> <pmc_write_dir> [traffic.json] > profiles/rNN_summary.md

With a fifth argument the HBM traffic per launch of the convolution kernels (implicit-GEMM tile variants, Winograd,
stem, narrow-N; launch-weighted) is also written as JSON; bench.py reports it as `roofline.traffic` for the same workload.

PMC passes are separate runs (SQ counters / FETCH_SIZE / WRITE_SIZE cannot share
a pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Corrections applied as that
guide prescribes: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced streaming reads, so read traffic is
quoted as 2 x FETCH_SIZE (upper-bound for narrow accesses; WRITE_SIZE is
uncalibrated).  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel
cycles), kernel cycles from GRBM_GUI_ACTIVE (summed over the 8 XCDs -> /8)."""
import collections
import csv
import glob
import json
import sys


def short(name):
    return name.replace("void ", "").split("(")[0]


def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            out[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
    return out, {k: len(v) for k, v in calls.items()}


def main():
    stats_dir, sq_dir, fetch_dir, write_dir = sys.argv[1:5]
    print("## kernel-trace --stats (%s)\n" % stats_dir)
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for f in glob.glob(stats_dir + "/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            print("| %s | %s | %.3f | %.2f | %s |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                   float(r["AverageNs"]) / 1e3, r["Percentage"]))
    sq, ncall = counters(sq_dir)
    fe, nf = counters(fetch_dir)
    wr, nw = counters(write_dir)
    print("\n## PMC (separate passes: %s, %s, %s)\n" % (sq_dir, fetch_dir, write_dir))
    print("| kernel | MFMA util | eff. clock-cycles/launch | LDS bank conflict cycles | WAIT_INST_ANY / WAVE_CYCLES | "
          "read MB/launch (2 x FETCH_SIZE) | write MB/launch (WRITE_SIZE) |")
    print("|---|---|---|---|---|---|---|")
    for k, c in sorted(sq.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        n = max(ncall.get(k, 1), 1)
        cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
        util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * cyc) if cyc else 0
        wait = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0
        rd = 2 * fe.get(k, {}).get("FETCH_SIZE", 0) * 1024 / max(nf.get(k, 1), 1) / 1e6
        ww = wr.get(k, {}).get("WRITE_SIZE", 0) * 1024 / max(nw.get(k, 1), 1) / 1e6
        print("| %s | %.1f %% | %.0f | %.0f | %.2f | %.1f | %.1f |" % (k, 100 * util, cyc / n, c.get("SQ_LDS_BANK_CONFLICT", 0) / n,
                                                                   wait, rd, ww))


    if len(sys.argv) > 5:
        extra = dict(a.split("=", 1) for a in sys.argv[6:] if "=" in a)      # e.g. batch=8: the workload the passes ran
        method = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python bench.py --steps 8 "
                  "--warmup 2 --no-cpu-baseline --secondary none`; KiB units; gfx950 correction: read bytes = 2 x FETCH_SIZE "
                  "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated")

        def traffic(fams):
            sel = lambda k: any(f in k for f in fams)
            rd = sum(v.get("FETCH_SIZE", 0) for k, v in fe.items() if sel(k)) * 2 * 1024
            nr = sum(n for k, n in nf.items() if sel(k))
            ww = sum(v.get("WRITE_SIZE", 0) for k, v in wr.items() if sel(k)) * 1024
            nwr = sum(n for k, n in nw.items() if sel(k))
            return {"kernels": list(fams), "read_bytes_per_launch": rd / max(nr, 1), "write_bytes_per_launch": ww / max(nwr, 1), "launches_sampled": nr}
        conv = ("conv_igemm_f32_kernel", "conv_igemm_b3_kernel", "conv_b3r_kernel", "conv_wino_f32_kernel", "conv_wino_b3_kernel", "conv_wino_b3s_kernel", "conv_halo_kernel", "conv_stem_f32_kernel", "conv1x1_ws_kernel",
                "conv_narrow_kernel", "conv_narrow3x3_kernel", "conv_igemm_f16_kernel")
        dom = traffic(("conv_igemm_b3_kernel", "conv_b3r_kernel"))      # the bf16x3 implicit-GEMM family: bench.py's roofline.kernel
        allc = traffic(conv)
        json.dump({"batch": int(extra.get("batch", 1)), "method": method, "dominant": dom, "all_conv": allc,
                   # (kept for readers of the round-1/2 files)
                   "kernel": "convolution kernels (all families)", "read_bytes_per_launch": allc["read_bytes_per_launch"],
                   "write_bytes_per_launch": allc["write_bytes_per_launch"], "launches_sampled": allc["launches_sampled"]},
                  open(sys.argv[5], "w"), indent=1)


if __name__ == "__main__":
    main()
