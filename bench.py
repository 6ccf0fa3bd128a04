#!/usr/bin/env python3
"""All-pairs KING throughput on MI355X (BASELINE.json metric: sample-pairs/s +
achieved HBM GB/s vs roofline).

    python bench.py [--gpus N --steps K --warmup W] [--config c1|c2|c3|c4|weak]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one synthetic cohort whose packed
bitset is already resident in HBM (reference layout, cuking.cu:507-523):
conversion of the bitset into the kernel-internal layout + the pair kernel
(filter variant on the matrix cores by default) over every (i < j) pair +
thresholded append of KingResult records.  EVERY timed step pays the
conversion, like a real job, which makes one pass per cohort (`value`); the
rate with the layout converted once and reused -- what round 3 quoted as
`value` -- is reported beside it as `value_layout_reused`, and the rate with
the host-to-device copy of the bitset added as `value_including_h2d`.
(Rounds 1-2 quoted configs[1] as the headline, round 3 on configs[2]:
figures of different rounds compare per configuration, under other_configs.)

Workloads (BASELINE.json configs).  The workload behind `value` is THE SAME AT
EVERY N -- configs[2], 100k samples x 100k sites (the largest configuration
BASELINE.json labels 1 x MI355X; 5.0e9 pairs, 0.6 s a pass on one GPU) -- so that
value(N) / value(1) is the strong-scaling speed-up:
  N = 1   the whole triangle on one GPU.  The same JSON line carries, under
          `other_configs`, configs[1] (10k x 100k) and configs[3]'s cohort
          (300k x 150k) on this ONE GPU.
  N > 1   STRONG scaling: the triangle is cut into N contiguous ranges of
          pair-space tiles (cuking_amd/dist.py), every rank holds the bitset
          (as every shard of the reference reads the whole input itself),
          and the records are gathered on rank 0 -- the only collective in
          the timed steps; the gather of one pass runs behind the kernel of
          the next.  After the timed region the same job is run ONCE more
          from a bitset that only rank 0 holds, in both broadcast forms
          (chunked + overlapped, and broadcast-then-compute): `with_broadcast`;
          and once on rank 0 alone: `single_gpu_same_run`, `speedup`.
          `other_configs` carries configs[1], configs[3] (and configs[4] from 4
          GPUs on) the same way, each with its own single-GPU pass and speed-up.
  --config c1|c3|c4 selects another headline; --config weak reproduces round
  1's weak-scaling series (10000 sqrt(N) samples x 100k sites).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import os

# The CPU baseline's OpenMP threads are pinned (read by libgomp when it is
# first loaded, i.e. by `import torch`).
os.environ.setdefault("OMP_PLACES", "cores")
os.environ.setdefault("OMP_PROC_BIND", "close")

import argparse  # noqa: E402
import json  # noqa: E402
import math  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402
from pathlib import Path  # noqa: E402

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# MI355X_MICROARCH.md, matrix cores: FP4 (block-scaled f8f6f4 MFMA) ~10 PF dense.
MFMA_FP4_PEAK_TFLOPS = 10000.0
# king_mfma.hip: plane products per pair and site, one multiply-add = 2 FLOP, by
# kernel variant (5: five-product form on the quad layout; 6: four-product form
# on the nibble layout; 7: the filter variant, king_filter.hip -- ONE product for
# every pair when the lean form runs with a threshold inside (0, 1/2), the exact
# sums only for the candidates that bound lets through; otherwise variant 6's
# kernel on the quadrants of its 256-sample tiles).
MFMA_MACS_PER_PAIR_SITE = {5: {"lean": 5, "full": 6}, 6: {"lean": 4, "full": 5},
                           7: {"lean": 1, "full": 5}}
MFMA_VARIANTS = (5, 6, 7)
MFMA_N4_MAX_SITES = 1 << 22
NOMINAL_CLOCK_HZ = 2.4e9        # MI355X_MICROARCH.md: max clock
NUM_SIMDS = 256 * 4
# king_kernels.hip, per pair per 32 sites: lean form 5 logic + 4 v_bcnt (used
# when kin_threshold > 0), full form 5 + 5.
VALU_OPS_PER_PAIR_WORD = {"lean": 9, "full": 10}
# VALU issue floor, measured on MI355X (tools/micro/king_step.hip, valu_phase.hip;
# archive/profiles/r01_valu_microbench.txt), cycles per wave64 instruction per SIMD:
# v_and 2.07, v_bitop3 2.37, v_bcnt_u32_b32 4.19 when each kind runs alone.
VALU_FLOOR_CYCLES_PER_PAIR_WORD = {"lean": 4 * 2.07 + 2.37 + 4 * 4.19,
                                   "full": 4 * 2.07 + 2.37 + 5 * 4.19}

CONFIGS = {
    "c1": dict(samples=10_000, sites=100_000, thr=0.05,
               name="BASELINE configs[1]: 10k samples x 100k sites, kin-threshold 0.05"),
    "c2": dict(samples=100_000, sites=100_000, thr=0.0884,
               name="BASELINE configs[2]: 100k samples x 100k sites, kin-threshold 0.0884 "
                    "(the reference's default, cuking.cu:43)"),
    "c3": dict(samples=300_000, sites=150_000, thr=0.0884,
               name="BASELINE configs[3]: 300k samples x 150k sites, kin-threshold 0.0884"),
    "c4": dict(samples=734_000, sites=200_000, thr=0.05,
               name="BASELINE configs[4]: 734k samples x 200k sites, kin-threshold 0.05, "
                    "IBS0/1/2 emitted"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="", choices=["", "c1", "c2", "c3", "c4", "weak"],
                    help="workload behind `value`; default c2 at every N (strong scaling)")
    ap.add_argument("--extra-configs", default=None,
                    help="comma list of further configs measured after the headline and "
                         "reported under other_configs (default c0,c1,c3 with the default "
                         "headline -- c0 = the real-input path through the C++ host, N = 1 "
                         "only --, + c4 from 4 GPUs on; 'none' disables)")
    ap.add_argument("--samples", type=int, default=0, help="override N samples")
    ap.add_argument("--sites", type=int, default=0, help="override the site count")
    ap.add_argument("--kin-threshold", type=float, default=None)
    ap.add_argument("--max-results", type=int, default=1 << 20)
    ap.add_argument("--kernel", default="tiled", choices=["tiled", "stream"])
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--band-rows", type=int, default=-1)
    ap.add_argument("--split-wgs", type=int, default=-1,
                    help="matrix-core kernel: pieces the remainder of a short launch is cut "
                         "into (-1 = library default, one per CU; 0 = never split)")
    ap.add_argument("--xcd-swizzle", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="matrix-core kernel: consecutive tiles per XCD (-1 = library default)")
    ap.add_argument("--counts-mode", type=int, default=-1, choices=[-1, 0, 1],
                    help="-1 automatic, 0 lean form (4 sums + recount of emitted pairs), "
                         "1 full form (5 sums)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="target CPU-baseline time (0 disables it)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="OpenMP threads of the CPU baseline (0 = min(16, available))")
    ap.add_argument("--dist-mode", default="resident",
                    choices=["resident", "staged", "simple"],
                    help="N>1, timed steps: the packed bitset is resident on every GPU "
                         "before the timed region, like the reference's shards that each "
                         "read the input themselves (resident, default); or rank 0 owns it "
                         "and every step distributes it first: chunked broadcast overlapped "
                         "with compute (staged) / broadcast then compute (simple)")
    ap.add_argument("--no-balance", action="store_true",
                    help="N>1: keep equal tile ranges (default: after the warm-up passes the "
                         "ranges are re-cut in proportion to each rank's measured kernel speed)")
    ap.add_argument("--no-single-gpu-pass", action="store_true",
                    help="N>1: skip the one pass of the WHOLE workload on rank 0 alone that "
                         "gives the single-GPU figure of the same run")
    ap.add_argument("--no-broadcast-pass", action="store_true",
                    help="N>1: skip the extra broadcast-inclusive passes after the timed region")
    ap.add_argument("--chunks", type=int, default=8, help="broadcast chunks (staged)")
    ap.add_argument("--streams", type=int, default=3, help="side streams for rectangle launches")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the planted-relatives check (timing-only tuning kernels)")
    ap.add_argument("--no-clock-pass", action="store_true",
                    help="skip the sustained-clock pass after the timed region")
    ap.add_argument("--max-launch-blocks", type=int, default=-1,
                    help="experiment: cap the workgroups per launch (library test hook)")
    ap.add_argument("--reuse-layout", action="store_true",
                    help="convert the bitset into the kernel-internal layout once per cohort and "
                         "reuse it in every timed step (round 3's `value`); default: every step "
                         "converts, like a real one-pass job, and the reused rate is reported "
                         "beside it as value_layout_reused")
    ap.add_argument("--convert-every-step", action="store_true",
                    help="(the default since round 4; accepted for old command lines)")
    ap.add_argument("--no-worst-case", action="store_true",
                    help="skip roofline.filter_worst_case (the headline cohort with 13 % missing "
                         "calls, default variant against variant 6)")
    ap.add_argument("--no-h2d-pass", action="store_true",
                    help="skip timing the host-to-device copy of the bitset "
                         "(value_including_h2d)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="library option set on the context (cuking_ctx_set_option), e.g. "
                         "filter_check1=0; repeatable (A/B runs)")
    ap.add_argument("--seed", type=int, default=20240229)
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(host_bits_fn, wps, gpu_records, thr, target_seconds, max_samples,
                 cpu_threads=0):
    """Times the oracle (oracle/king_oracle.c, -march=native, OpenMP over rows,
    threads pinned to cores) on a leading sub-block of the SAME cohort, and
    checks the GPU's records for that sub-block against it.
    Test-infrastructure use only."""
    import ctypes as C
    import tempfile
    import numpy as np
    from oracle import pyoracle
    # The GPU box's CPU share for one GPU is 16 hardware threads (its 8 GPUs
    # share the host); --cpu-threads overrides.
    threads = cpu_threads or min(16, len(os.sched_getaffinity(0)))
    out_dir = Path(tempfile.mkdtemp(prefix="cuking_oracle_"))
    lib = pyoracle.load(native=True, out_dir=out_dir)

    def run(s):
        bits = host_bits_fn(s)
        sm = pyoracle.submatrix(s)
        res = np.zeros(1 << 20, dtype=pyoracle.RESULT_DTYPE)
        ovf = C.c_uint32(0)
        t0 = time.perf_counter()
        n = lib.orc_compute_mt(C.byref(sm), wps, bits.ctypes.data_as(C.c_void_p),
                               thr, res.size, res.ctypes.data_as(C.c_void_p),
                               C.byref(ovf), threads)
        dt = time.perf_counter() - t0
        res = res[:n].copy()
        lib.orc_sort(res.ctypes.data_as(C.c_void_p), res.size)
        return s * (s - 1) // 2, dt, res

    run(256)
    pairs, dt, _ = run(1024)                     # calibration (threads warm)
    rate = pairs / dt
    s = int(min(max_samples, max(256, math.sqrt(2 * rate * target_seconds))))
    pairs, dt, res = run(s)
    sel = gpu_records[(gpu_records["sample_j"] < s)]
    if sel.tobytes() != res.tobytes():
        raise SystemExit(f"PARITY FAILURE: GPU records for the first {s} samples "
                         "differ from the CPU oracle")
    return {"value": pairs / dt, "unit": "sample-pairs/s", "cores": threads,
            "value_per_core": pairs / dt / threads,
            "kind": "port",
            "omp": {"OMP_PLACES": os.environ.get("OMP_PLACES"),
                    "OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"),
                    "threads": threads,
                    "hardware_threads_visible": len(os.sched_getaffinity(0))},
            "sample": f"first {s} samples ({pairs} pairs) of the same cohort, "
                      f"{dt:.1f} s, OpenMP x{threads} pinned (OMP_PLACES=cores) on "
                      f"{cpu_model()}, -O3 -march=native; records checked equal to the GPU's"}


def committed_profile(workload_key, kernel_name):
    """Figures from committed rocprofv3 passes of this command (profiles/): HBM
    bytes per launch (PMC, corrected as MI355X_MICROARCH.md prescribes) and the
    kernel's average duration in the --kernel-trace --stats pass.  They are
    CONSTANTS read from files, not measured by this run -- the JSON says so."""
    p = ROOT / "profiles" / "hbm_traffic.json"
    if not p.exists():
        return {}
    try:
        return json.loads(p.read_text()).get(f"{workload_key}:{kernel_name}", {})
    except Exception:
        return {}


def counts_form(args, thr, wps, variant):
    if args.counts_mode == 1:
        return "full"
    if args.counts_mode == 0:
        return "lean"
    c = 2.05 if args.kernel == "tiled" and variant in MFMA_VARIANTS else 1.6
    return "lean" if thr > 0 and thr * thr * 32 * wps >= c * c else "full"


def roofline_block(args, ctx, *, launch_pairs, sites, wps, thr, king_ms, prepare_ms,
                   launches, workload_key, clock_mhz, use_profile=True):
    """The `roofline` object for the pair kernel: algorithmic work of one launch
    / its average duration measured with HIP events on the launching stream."""
    import cuking_amd
    bpp = cuking_amd.bytes_per_pair(wps)
    variant = ctx.get_option("variant")
    if variant in (6, 7) and 32 * wps > MFMA_N4_MAX_SITES:
        variant = 5          # (the library hands wider bitsets to the five-product form)
    mfma = args.kernel == "tiled" and variant in MFMA_VARIANTS and 32 * wps <= (1 << 24)
    form = counts_form(args, thr, wps, variant)
    if variant == 7 and not (form == "lean" and 0.0 < thr < 0.5):
        variant = 6          # (no bound to apply: the four-product kernel, quadrant mode)
    kernel_name = ("king_stream_kernel" if args.kernel == "stream" else
                   "king_filter_kernel" if mfma and variant == 7 else
                   "king_mfma_kernel" if mfma else "king_tiled_kernel")
    achieved = launch_pairs * bpp / (king_ms * 1e-3) / 1e9 if king_ms > 0 else 0.0
    hbm_view = {
        "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "algorithmic_bytes_per_pair": bpp,
        "note": "algorithmic bytes (2 samples x words_per_sample x 8 B per pair, "
                "SURVEY.md 8d) / measured kernel time; operands are re-used from LDS and "
                "registers, so the quotient to the HBM peak is a REUSE FACTOR (far above 1 "
                "by design), not a fraction of a roofline: the matrix-core figure is the bound",
    }
    # (`frac` when HBM is the bound -- the VALU / stream kernels; for the matrix-core
    #  kernels the same quotient is called what it is)
    hbm_view["algorithmic_reuse_factor" if mfma else "frac"] = achieved / HBM_PEAK_GBPS
    prof = (committed_profile(workload_key + (":full" if form == "full" else ""), kernel_name)
            if use_profile else {})
    common = {
        "traffic": prof.get("traffic_bytes_per_launch"),
        "traffic_source": (f"committed rocprofv3 --pmc passes of this command "
                           f"({prof.get('source', 'profiles/hbm_traffic.json')}); a constant "
                           "read from the file, NOT measured by this run"
                           if prof.get("traffic_bytes_per_launch") is not None else
                           "not measured for this workload"),
        "kernel": kernel_name, "kernel_ms": king_ms,
        "kernel_ms_source": "HIP events around every launch on its stream, this run",
        # (committed figure of ANOTHER run, possibly another box: the median of the
        #  timed launches where the profile has it, else rocprofv3's average)
        "kernel_ms_rocprof": prof.get("rocprof_median_ms") or prof.get("rocprof_avg_ms"),
        "kernel_ms_rocprof_source": (prof.get("rocprof_source")
                                     if (prof.get("rocprof_median_ms") or prof.get("rocprof_avg_ms"))
                                     else None),
        "launches": launches, "prepare_ms": prepare_ms,
        "sustained_clock_mhz": clock_mhz,
        "sustained_clock_source": ("in-kernel s_memtime / s_memrealtime probe of ONE wavefront "
                                   "on one CU during a separate pass of the same steps "
                                   "(cuking_clock_probe): indicative only -- the chip-wide clock "
                                   "under the kernel is effective_clock_mhz_pmc in "
                                   "profiles/hbm_traffic.json" if clock_mhz else None),
    }
    if mfma:
        macs = MFMA_MACS_PER_PAIR_SITE[variant][form]
        per_mac = (launch_pairs * sites * 2 / (king_ms * 1e-3) / 1e12) if king_ms > 0 else 0.0
        tflops = per_mac * macs
        # rounds 1-2 quoted the five-product form (5 / 6 products): the same time
        # priced at that form's FLOP, for comparison across rounds
        equiv = per_mac * MFMA_MACS_PER_PAIR_SITE[5][form]
        return {
            "bound": "mfma", "achieved": tflops, "peak": MFMA_FP4_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": tflops / MFMA_FP4_PEAK_TFLOPS, **common,
            "form": form, "macs_per_pair_site": macs, "kernel_variant": variant,
            "frac_five_product_equivalent": equiv / MFMA_FP4_PEAK_TFLOPS,
            "note": "fp4 (E2M1) v_mfma_f32_32x32x64_f8f6f4, exact integer sums in f32; "
                    "algorithmic FLOP = pairs x sites x plane products the kernel issues x 2 "
                    "(padding of tiles and of the last k-step not counted); `frac` is the "
                    "share of the dense FP4 peak those products take, "
                    "`frac_five_product_equivalent` prices the same time at the five-product "
                    "form of rounds 1-2 (above 1.0 for the filter variant: it issues one "
                    "product where that form issued five); peak at the nominal 2.4 GHz, the "
                    "chip holds less under this load (sustained_clock_mhz)" +
                    ("; filter variant: kernel_ms spans the filter kernel AND the refine / "
                     "dense-quadrant kernels behind it (one event pair per call), the "
                     "algorithmic FLOP are the filter kernel's" if variant == 7 else ""),
            "hbm": hbm_view,
        }
    roofline = {"bound": "hbm", **hbm_view, **common}
    cyc = (king_ms * 1e-3 * NOMINAL_CLOCK_HZ * NUM_SIMDS * 64 /
           (launch_pairs * wps)) if king_ms > 0 else 0.0
    floor = VALU_FLOOR_CYCLES_PER_PAIR_WORD[form]
    roofline["valu"] = {
        "form": form, "ops_per_pair_word": VALU_OPS_PER_PAIR_WORD[form],
        "achieved_cycles_per_pair_word": cyc, "floor_cycles_per_pair_word": floor,
        "frac": floor / cyc if cyc else 0.0,
    }
    return roofline


def filter_counters(ctx):
    """(candidates, dense quadrants, tiles that left early, tiles that started at another
    phase) so far, or None when the context's variant is not
    the filter variant (diagnostic options of the library: they wait for the device)."""
    if ctx.get_option("variant") != 7:
        return None
    return (ctx.get_option("filter_candidates"), ctx.get_option("filter_dense_quadrants"),
            ctx.get_option("filter_early_exits"), ctx.get_option("filter_rotated_tiles"))


def records_of(results, count):
    import numpy as np
    import cuking_amd
    recs = results[:count].cpu().numpy().view(np.uint32).reshape(-1).view(
        cuking_amd.KING_RESULT_DTYPE).copy()
    return cuking_amd.sort_results(recs)


def check_planted(recs, cohort, thr, no_check):
    """Every planted relative the threshold admits must be reported (half
    siblings sit near 0.125: only asked for below 0.06)."""
    if no_check:
        return
    got = {(int(r["sample_i"]), int(r["sample_j"])) for r in recs}
    kinds = ("dup", "po", "sib", "half") if thr <= 0.06 else ("dup", "po", "sib")
    missing = [p for p in cohort.planted
               if p[2] in kinds and (min(p[0], p[1]), max(p[0], p[1])) not in got]
    if missing:
        raise SystemExit(f"{len(missing)} planted relatives not reported")


def single_gpu_workload(args, ctx, n, m, thr, steps, warmup, local_rank, clock_pass=True,
                        h2d_pass=False):
    """`steps` timed passes over a resident cohort on one GPU (every pass converts
    the bitset unless --reuse-layout).  Returns the measurements plus the records
    of the last pass and the device bitset."""
    import torch
    import cuking_amd
    from cuking_amd.synth import cohort_to_device, plan_cohort
    dev = f"cuda:{local_rank}"
    wps = cuking_amd.words_per_sample(m)
    sm = cuking_amd.Submatrix(n)
    pairs = sm.NumPairs()
    cohort = plan_cohort(n, args.seed)
    kind, pa, pb = cohort_to_device(cohort, local_rank)
    bits = ctx.synth_bitset(args.seed, kind, pa, pb, 0, n, m)
    results = torch.zeros((args.max_results, 6), dtype=torch.int32, device=dev)
    index_flag = torch.zeros(2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.invalidate()      # a new cohort (possibly behind a recycled pointer)

    def step():
        index_flag.zero_()
        ctx.compute_king(sm, wps, bits, thr, args.max_results, results,
                         index_flag[0:1], index_flag[1:2])

    ctx.timing_reset()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    warm = ctx.timing_collect()
    ctx.timing_reset()
    filt0 = filter_counters(ctx)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timing = ctx.timing_collect()
    filt1 = filter_counters(ctx)
    count, ovf = index_flag.tolist()
    if ovf:
        raise SystemExit("result overflow: raise --max-results")
    recs = records_of(results, count)
    check_planted(recs, cohort, thr, args.no_check)

    # The same steps once more with the layout converted once and reused (round 3's
    # `value`): what a host that runs several passes over one cohort would see.
    reused = None
    if not args.reuse_layout:
        ctx.set_option("reuse_prepared", 1)
        step()                                   # (converts if the workspace is stale)
        torch.cuda.synchronize()
        r0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        reused = (time.perf_counter() - r0) / steps
        ctx.set_option("reuse_prepared", 0)
        ctx.invalidate()

    # The boundary hands over host buffers (cuking_copy_to_device): the copy of the
    # packed bitset from page-locked host memory, timed with events, best of three.
    h2d_ms = None
    if h2d_pass and not args.no_h2d_pass:
        host = torch.empty(bits.shape, dtype=bits.dtype, pin_memory=True)
        host.copy_(bits)
        back = torch.empty_like(bits)
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            back.copy_(host, non_blocking=True)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1)
            best = t if best is None else min(best, t)
        h2d_ms = best
        del host, back

    clock = None
    if clock_pass and not args.no_clock_pass:
        # the same steps once more with the one-wave clock probe beside them
        # (it holds a wave slot of one CU, so it stays out of the timed region)
        span_us = max(2000, int(0.8 * elapsed * 1e6))
        read = ctx.clock_probe(span_us, torch.cuda.Stream())
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        clock = read()
    # (with the layout reused, the one conversion of the cohort happened in the
    #  first call -- a warm-up pass unless there was none)
    prep = timing if timing.prepare_launches else warm
    return dict(n=n, m=m, thr=thr, wps=wps, pairs=pairs, elapsed=elapsed, steps=steps,
                king_ms=timing.king_ms / max(timing.king_launches, 1),
                prepare_ms=prep.prepare_ms / max(prep.prepare_launches, 1),
                prepare_launches_timed=timing.prepare_launches,
                launches=timing.king_launches, recs=recs, bits=bits, cohort=cohort,
                clock_mhz=clock, reused_s_per_step=reused, h2d_ms=h2d_ms,
                filter=(None if filt0 is None else
                        {"candidates_per_pass": (filt1[0] - filt0[0]) / steps,
                         "dense_quadrants_per_pass": (filt1[1] - filt0[1]) / steps,
                         "tiles_left_at_the_check_point_per_pass": (filt1[2] - filt0[2]) / steps,
                         "tiles_per_pass": ctx.num_tiles(sm),
                         "tiles_started_at_another_phase_per_pass": (filt1[3] - filt0[3]) / steps,
                         "k_step_us_measured_by_the_tiles": ctx.get_option("filter_step_ticks16") / 1600.0,
                         "records_per_pass": len(recs),
                         "note": "filter variant: pairs its bound let through to the exact "
                                 "recount, 128 x 128 quadrants handed to the four-product "
                                 "kernel instead, 256 x 256 tiles that left at the rigorous "
                                 "check point inside the k loop (no pair of theirs could still "
                                 "become a candidate: king_filter.hip, or hand a few live "
                                 "pairs to the candidate list and leave), tiles that started "
                                 "their k loop where the tiles of their XCD were (rotated "
                                 "tiles) with the time per k-step of 256 sites they measured "
                                 "(100 MHz counter; 0 = launch too short to rotate), and the "
                                 "records of a pass"}))


def filter_worst_case(args, ctx, bits, n, m, thr, local_rank):
    """The headline cohort with 12.5 % extra missing calls (a random mask OR-ed into both
    planes: missing = 11, cuking.cu:507-523) -- beyond the reach of the filter's bound, so
    the default variant hands every tile to the four-product kernel: one warm-up and two
    timed passes of the default variant and of variant 6 on the SAME bitset, records
    compared.  What the default costs when it cannot help."""
    import torch
    import cuking_amd
    dev = f"cuda:{local_rank}"
    wps = cuking_amd.words_per_sample(m)
    half = wps // 2
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    dense = bits.clone()
    mask = None
    for _ in range(3):                       # density 2^-3
        r = torch.randint(-(1 << 63), (1 << 63) - 1, (n, half), dtype=torch.int64, device=dev,
                          generator=gen)
        mask = r if mask is None else mask & r
    dense[:, :half] |= mask
    dense[:, half:2 * half] |= mask
    del mask, r
    sm = cuking_amd.Submatrix(n)
    results = torch.zeros((args.max_results, 6), dtype=torch.int32, device=dev)
    index_flag = torch.zeros(2, dtype=torch.int32, device=dev)
    default_variant = ctx.get_option("variant")
    out = {}
    recs = {}
    for name, variant in (("default", default_variant), ("variant_6", 6)):
        ctx.set_option("variant", variant)
        ctx.invalidate()
        times = []
        for rep in range(3):
            index_flag.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.compute_king(sm, wps, dense, thr, args.max_results, results, index_flag[0:1],
                             index_flag[1:2])
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        count, ovf = index_flag.tolist()
        if ovf:
            raise SystemExit("result overflow in the worst-case pass")
        recs[name] = records_of(results, count)
        out[name + "_ms_per_step"] = min(times[1:])
    ctx.set_option("variant", default_variant)
    ctx.invalidate()
    if recs["default"].tobytes() != recs["variant_6"].tobytes() and not args.no_check:
        raise SystemExit("PARITY FAILURE: worst-case pass, default variant != variant 6")
    out["ratio"] = out["default_ms_per_step"] / out["variant_6_ms_per_step"]
    out["records"] = int(len(recs["default"]))
    out["what"] = ("the headline cohort with 12.5 % extra missing calls (13.4 % in all): the "
                   "filter's bound cannot thin it out and the default hands every tile to the "
                   "four-product kernel; wall ms of one pass (conversion inside), best of two, "
                   "the default variant and variant 6 on the same bitset, records identical")
    return out


def dist_workload(args, env, ctx, key, n, m, thr, steps, warmup, headline):
    """One workload sharded over the ranks (strong scaling): every rank holds the
    bitset, takes a range of the tile enumeration, records gathered on rank 0.
    Returns the measurements (rank 0: a dict; others: None).  `headline` adds the
    broadcast-inclusive passes and the unpipelined gather timing."""
    import torch
    import torch.distributed as dist
    import cuking_amd
    from cuking_amd.dist import (GpuStagedOps, PipelinedGather, all_pairs_king,
                                 all_pairs_king_staged, gather_results,
                                 gather_results_device, rank_tile_share, tile_partition,
                                 weighted_tile_partition)
    from cuking_amd.synth import cohort_to_device, plan_cohort
    world, rank, local_rank, dev = env["world"], env["rank"], env["local_rank"], env["dev"]
    host_or_dev = env["host_or_dev"]
    wps = cuking_amd.words_per_sample(m)
    sm = cuking_amd.Submatrix(n)
    pairs = sm.NumPairs()
    cohort = plan_cohort(n, args.seed)
    kind, pa, pb = cohort_to_device(cohort, local_rank)
    bits = torch.zeros((n, wps), dtype=torch.int64, device=dev)
    if rank == 0 or args.dist_mode == "resident":
        ctx.synth_bitset(args.seed, kind, pa, pb, 0, n, m, out=bits)
    torch.cuda.synchronize()
    ctx.invalidate()      # (a new cohort, possibly behind a recycled pointer)

    results = torch.zeros((args.max_results, 6), dtype=torch.int32, device=dev)
    index_flag = torch.zeros(2, dtype=torch.int32, device=dev)
    # second record buffer: the gather of one pass overlaps the kernel of the
    # next (cuking_amd.dist.PipelinedGather)
    results_b = torch.zeros_like(results)
    index_flag_b = torch.zeros_like(index_flag)
    num_tiles = ctx.num_tiles(sm) if args.kernel == "tiled" else 0
    my_tiles = list(tile_partition(num_tiles, world)[rank])
    # nccl gathers straight from the kernel's own counters; gloo (one-GPU
    # rehearsals) needs host tensors and takes the staging path
    device_gather = dist.get_backend() == "nccl"

    def compute_tiles(bit_sets, begin, end):
        index_flag.zero_()
        ctx.compute_king(sm, wps, bit_sets, thr, args.max_results, results,
                         index_flag[0:1], index_flag[1:2], tile_range=(begin, end))
        if device_gather:
            return results, index_flag     # counts stay on the device until the gather
        count, ovf = index_flag.tolist()   # waits for the kernel
        return results, count, ovf

    gathered = [None]
    tile = ctx.tile_samples()
    staged_ops = (GpuStagedOps(ctx, sm, wps, bits, thr, args.max_results,
                               num_streams=args.streams)
                  if args.kernel == "tiled" else None)
    pipelined = (device_gather and args.dist_mode == "resident" and args.kernel == "tiled" and
                 os.environ.get("CUKING_BENCH_NO_PIPELINE") != "1")
    pipe = PipelinedGather() if pipelined else None
    pending = [None]
    parity = [0]

    def pipelined_step():
        # pass k: kernel into buffer k % 2, its gather starts behind it; then the
        # gather of pass k - 1 is collected while this pass's kernel runs
        buf, flag = ((results, index_flag), (results_b, index_flag_b))[parity[0]]
        parity[0] ^= 1
        flag.zero_()
        error = None
        try:
            ctx.compute_king(sm, wps, bits, thr, args.max_results, buf, flag[0:1], flag[1:2],
                             tile_range=my_tiles)
        except Exception as e:  # noqa: BLE001 - raised on every rank by finish()
            error = e
        handle = pipe.begin(buf, flag, error=error)
        if pending[0] is not None:
            gathered[0] = pipe.finish(pending[0])
        pending[0] = handle

    def drain():
        if pending[0] is not None:
            gathered[0] = pipe.finish(pending[0])
            pending[0] = None

    def step(mode=None):
        mode = mode or args.dist_mode
        if pipelined and mode == "resident":
            return pipelined_step()
        if mode == "staged" and staged_ops is not None:
            ctx.invalidate()   # the bitset arrives anew: every chunk is converted again
            gathered[0], _ = all_pairs_king_staged(staged_ops, n, tile, bits,
                                                   num_chunks=args.chunks)
        elif mode == "resident" and args.kernel == "tiled":
            # this rank's (possibly re-balanced) range, then the gather
            out_ = compute_tiles(bits, *my_tiles)
            gathered[0] = (gather_results_device(*out_) if device_gather
                           else gather_results(*out_))
        else:
            if mode != "resident":
                ctx.invalidate()
            gathered[0], _ = all_pairs_king(compute_tiles, num_tiles, bits,
                                            broadcast=mode != "resident",
                                            record_capacity=args.max_results,
                                            device_counts=device_gather)

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    balance = {"applied": False}
    ctx.timing_reset()
    first_prepare_ms = None
    for w in range(warmup):
        if w == warmup - 1:
            early = ctx.timing_collect()        # (the cohort's one conversion, if it is in here)
            if early.prepare_launches:
                first_prepare_ms = early.prepare_ms / early.prepare_launches
            ctx.timing_reset()          # the last warm-up pass doubles as calibration
        step()
    if pipelined:
        drain()
    barrier()
    if (warmup >= 1 and args.dist_mode == "resident" and args.kernel == "tiled"
            and not args.no_balance and world > 1):
        # The GPUs of a node sustain different clocks under this load (several
        # percent); with equal ranges the slowest sets the pace.  Re-cut the tile
        # ranges in proportion to what each rank just measured for itself.
        cal = ctx.timing_collect()
        if cal.prepare_launches and first_prepare_ms is None:
            first_prepare_ms = cal.prepare_ms / cal.prepare_launches
        rate = torch.tensor([(my_tiles[1] - my_tiles[0]) / max(cal.king_ms, 1e-6)],
                            dtype=torch.float64, device=host_or_dev)
        rates = [torch.zeros_like(rate) for _ in range(world)]
        dist.all_gather(rates, rate)
        rates = [float(r) for r in rates]
        balance.update(rank_tiles_per_ms=rates, spread=max(rates) / min(rates) - 1.0)
        if min(rates) > 0 and balance["spread"] > 0.015:
            my_tiles = list(weighted_tile_partition(num_tiles, rates)[rank])
            balance["applied"] = True
        barrier()
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if pipelined:
        drain()          # the last pass's records are on rank 0 before the clock stops
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=host_or_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)
    timing = ctx.timing_collect()
    recs = gathered[0]

    # every rank's pair-kernel time per step (HIP events on its own stream)
    mine = torch.tensor([timing.king_ms / max(steps, 1),
                         timing.prepare_ms / max(steps, 1)],
                        dtype=torch.float64, device=host_or_dev)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(per_rank, mine)
    rank_kernel_ms = [float(x[0]) for x in per_rank]
    rank_prepare_ms = [float(x[1]) for x in per_rank]

    gather_ms = None
    with_broadcast = None
    if headline:
        # the gather alone: one unpipelined pass, timed from "my kernel is done"
        barrier()
        out_tiles = compute_tiles(bits, *my_tiles)
        torch.cuda.synchronize()
        dist.barrier()
        g0 = time.perf_counter()
        if device_gather:
            gather_results_device(out_tiles[0], out_tiles[1])
        else:
            gather_results(*out_tiles)
        gather_ms = (time.perf_counter() - g0) * 1e3

        # broadcast-inclusive passes: only rank 0 holds the bitset; the others
        # start from zeros and receive it through RCCL inside the measured time
        if not args.no_broadcast_pass and args.dist_mode == "resident" and staged_ops is not None:
            with_broadcast = {}
            for mode in ("staged", "simple"):
                if rank != 0:
                    bits.zero_()
                barrier()
                b0 = time.perf_counter()
                step(mode)
                barrier()
                tb = torch.tensor([time.perf_counter() - b0], dtype=torch.float64,
                                  device=host_or_dev)
                dist.all_reduce(tb, op=dist.ReduceOp.MAX)
                ok = gathered[0] is None or gathered[0].tobytes() == recs.tobytes()
                if rank == 0 and not ok:
                    raise SystemExit(f"{mode} broadcast pass: records differ from the resident pass")
                with_broadcast[mode] = {
                    "seconds": float(tb), "value": pairs / float(tb), "unit": "sample-pairs/s",
                    "bitset_bytes_broadcast": int(bits.numel() * 8),
                    "what": ("chunked RCCL broadcast from rank 0 overlapped with rectangle "
                             "kernels (all_pairs_king_staged)" if mode == "staged" else
                             "RCCL broadcast from rank 0, then equal tile ranges "
                             "(all_pairs_king)") + "; one pass, records identical to the "
                                                   "resident pass"}
            ctx.invalidate()

    # The single-GPU figure of the SAME run: rank 0 alone evaluates every pair
    # (the others wait), so that the strong-scaling ratio can be read off one JSON
    # line instead of two runs on possibly different boxes.
    single = None
    if not args.no_single_gpu_pass and args.kernel == "tiled":
        barrier()
        if rank == 0:
            reps = 3 if pairs < 10**9 else 1          # (short passes: take the best of three)
            best = None
            for _ in range(reps):
                index_flag.zero_()
                torch.cuda.synchronize()
                s0 = time.perf_counter()
                ctx.compute_king(sm, wps, bits, thr, args.max_results, results,
                                 index_flag[0:1], index_flag[1:2])
                torch.cuda.synchronize()
                s1 = time.perf_counter() - s0
                best = s1 if best is None else min(best, s1)
            cnt1, ovf1 = index_flag.tolist()
            same = (not ovf1) and records_of(results, cnt1).tobytes() == recs.tobytes()
            if not same and not args.no_check:
                raise SystemExit("single-GPU pass: records differ from the sharded pass")
            single = {"seconds": best, "value": pairs / best, "unit": "sample-pairs/s",
                      "what": "rank 0 alone, the whole workload, after the timed region "
                              f"(best of {reps}); records identical to the sharded pass"}
        barrier()

    out = None
    if rank == 0:
        check_planted(recs, cohort, thr, args.no_check)
        if args.dist_mode == "staged":
            share = rank_tile_share((n + tile - 1) // tile, world, 0)
        else:
            share = (my_tiles[1] - my_tiles[0]) / max(num_tiles, 1)
        king_ms = timing.king_ms / max(steps, 1)
        roofline = roofline_block(
            args, ctx, launch_pairs=pairs * share, sites=m, wps=wps, thr=thr, king_ms=king_ms,
            prepare_ms=(timing.prepare_ms / timing.prepare_launches if timing.prepare_launches
                        else (first_prepare_ms or 0.0)),
            launches=timing.king_launches, workload_key=f"{n}x{m}", clock_mhz=None,
            use_profile=False)
        roofline["note_rank"] = ("rank 0's launches over its share of the pairs "
                                 f"({share:.4f}); per-rank figures under config")
        value = pairs * steps / elapsed
        out = dict(key=key, n=n, m=m, thr=thr, pairs=pairs, steps=steps, warmup=warmup,
                   elapsed=elapsed, value=value, recs=recs, roofline=roofline,
                   rank_kernel_ms=rank_kernel_ms, rank_prepare_ms=rank_prepare_ms,
                   gather_ms=gather_ms, balance=balance, with_broadcast=with_broadcast,
                   single=single, bitset_bytes=int(bits.numel() * 8),
                   speedup=(value / single["value"]) if single else None)
    del bits, results, results_b, staged_ops
    torch.cuda.empty_cache()
    return out


def genotypes_from_bits(bits, num_sites):
    """int8 [samples, sites] (0 / 1 / 2 alternate alleles, -1 missing) from a
    reference-layout bitset (cuking.cu:507-523: het plane, hom-alt plane, both = missing)."""
    import numpy as np
    half = bits.shape[1] // 2
    het = np.unpackbits(np.ascontiguousarray(bits[:, :half]).view(np.uint8), axis=1,
                        bitorder="little")[:, :num_sites].astype(bool)
    hom = np.unpackbits(np.ascontiguousarray(bits[:, half:2 * half]).view(np.uint8), axis=1,
                        bitorder="little")[:, :num_sites].astype(bool)
    geno = np.zeros(het.shape, dtype=np.int8)
    geno[het & ~hom] = 1
    geno[hom & ~het] = 2
    geno[het & hom] = -1
    return geno


def run_cuking(in_dir, out_dir, threads, *extra):
    """One run of the C++ host (cuking_amd/bin/cuking) as a child process: wall seconds
    and its JSON summary line."""
    import subprocess
    cli = ROOT / "cuking_amd" / "bin" / "cuking"
    t0 = time.perf_counter()
    p = subprocess.run([str(cli), "--input_uri", str(in_dir), "--output_uri", str(out_dir),
                        f"--num_reader_threads={threads}", *extra],
                       capture_output=True, text=True, timeout=900)
    wall = time.perf_counter() - t0
    if p.returncode != 0:
        raise SystemExit(f"cuking failed on the real-input path: {p.stderr[-1500:]}")
    return wall, json.loads(p.stdout.strip().splitlines()[-1])


def real_input_path(args, ctx, local_rank):
    """BASELINE configs[0]: 1k samples x 10k sites as REAL input -- 8 zstd Parquet files
    with OPTIONAL columns in a Spark-style directory (cuking_amd/inputs.py = the layout of
    mt_to_cuking_inputs.py:26-47) -- end to end through the C++ host `cuking`
    (cuking.cu:435-882: list, decode, pack, kernel, sort, Snappy Parquet): wall seconds,
    triples/s of decode + pack, the split the binary reports, the records compared with
    the CPU oracle's, and the oracle's own time for the pair pass beside it.  Then, on
    boxes that expose at least 8 hardware threads, a larger real-Parquet point (2000
    samples x 50000 sites = 1e8 triples, 8 files of several row groups) through the host
    pack and the pipelined device pack."""
    import shutil
    import tempfile
    import numpy as np
    import pyarrow.parquet as pq
    import torch
    import cuking_amd
    from concurrent.futures import ProcessPoolExecutor
    from cuking_amd.inputs import write_input_tables
    from cuking_amd.synth import cohort_to_device, plan_cohort
    from oracle import pyoracle
    cpus = len(os.sched_getaffinity(0))
    threads = max(1, min(16, cpus))
    n, m, thr = 1000, 10_000, 0.05
    d = Path(tempfile.mkdtemp(prefix="cuking_c0_"))
    try:
        cohort = plan_cohort(n, args.seed)
        kind, pa, pb = cohort_to_device(cohort, local_rank)
        bits = ctx.synth_bitset(args.seed, kind, pa, pb, 0, n, m)
        torch.cuda.synchronize()
        host_bits = np.ascontiguousarray(bits.cpu().numpy().view(np.uint64))
        del bits
        geno = genotypes_from_bits(host_bits, m)
        ids = [f"S{k:07d}" for k in range(n)]
        t0 = time.perf_counter()
        write_input_tables(d / "in", geno, ids, num_files=8, compression="zstd", nullable=True,
                           spark_layout=True)
        generate_s = time.perf_counter() - t0
        parquet_bytes = sum(p.stat().st_size for p in (d / "in").glob("*.parquet"))
        runs = []
        for rep in range(3):                      # (the first run pages the binary and the files in)
            wall, summ = run_cuking(d / "in", d / f"out{rep}", threads, f"--kin_threshold={thr}")
            runs.append((wall, summ))
        wall, summ = min(runs[1:], key=lambda r: r[0])
        # the CPU oracle on the same bitset: its records against the file's, and its time
        pyoracle.compute(pyoracle.submatrix(64), host_bits[:64], thr, 1 << 16, threads=threads)
        t0 = time.perf_counter()
        exp, ovf, _ = pyoracle.compute(pyoracle.submatrix(n), host_bits, thr, 1 << 20,
                                       threads=threads)
        oracle_s = time.perf_counter() - t0
        t = pq.read_table(d / "out2" / "part-00000.snappy.parquet")
        same = (t.num_rows == len(exp) and
                t.column("i").to_pylist() == [ids[x] for x in exp["sample_i"]] and
                t.column("j").to_pylist() == [ids[x] for x in exp["sample_j"]] and
                np.array_equal(t.column("kin").to_numpy().view(np.uint32), exp["kin"].view(np.uint32))
                and all(np.array_equal(t.column(c).to_numpy().astype(np.uint32), exp[c])
                        for c in ("ibs0", "ibs1", "ibs2")))
        if not same and not args.no_check:
            raise SystemExit("PARITY FAILURE: configs[0] through cuking differs from the CPU oracle")
        out = {
            "workload": "BASELINE configs[0]: 1k samples x 10k sites as real input (8 zstd "
                        "Parquet files, OPTIONAL columns, Spark-style directory) through "
                        "cuking_amd/bin/cuking, kin-threshold 0.05",
            "wall_seconds": wall, "triples": summ["triples"], "parquet_MB": parquet_bytes / 1e6,
            "read_pack_seconds": summ["read_pack_seconds"],
            "triples_per_second": summ["triples_per_second"],
            "decode_thread_seconds": summ["decode_thread_seconds"],
            "pack_thread_seconds": summ["pack_thread_seconds"],
            "kernel_seconds": summ["kernel_seconds"], "pack": summ["pack"],
            "reader_threads": threads, "hardware_threads_visible": cpus,
            "results": summ["results"], "pairs": summ["pairs"],
            "checks": "output file's records identical to the CPU oracle's (ids, kin bits, IBS0/1/2)",
            "wall_seconds_all_runs": [r[0] for r in runs],
            "cpu_oracle": {"pair_pass_seconds": oracle_s, "threads": threads,
                           "value": summ["pairs"] / oracle_s, "unit": "sample-pairs/s",
                           "what": "oracle/king_oracle.c on the same packed bitset (pair pass only: "
                                   "no decode, no pack)"},
            "input_generation_seconds": generate_s,
        }
        if cpus >= 8:
            sys.path.insert(0, str(ROOT / "tools"))
            import cli_timing
            n2, m2, files = 2000, 50_000, 8
            big = d / "big"
            big.mkdir()
            (big / "metadata.json").write_text(json.dumps(
                {"num_sites": m2, "samples": [f"S{k:07d}" for k in range(n2)]}))
            bounds = np.linspace(0, m2, files + 1).astype(int)
            jobs = [(str(big), f, int(bounds[f]), int(bounds[f + 1]), n2, 1, 2_000_000)
                    for f in range(files)]
            t0 = time.perf_counter()
            with ProcessPoolExecutor(min(8, cpus)) as ex:
                triples = sum(ex.map(cli_timing.write_part, jobs))
            gen2 = time.perf_counter() - t0
            point = {"workload": f"{n2} samples x {m2} sites as real input ({files} zstd files of "
                                 "several row groups)", "triples": triples,
                     "parquet_MB": sum(p.stat().st_size for p in big.glob("*.parquet")) / 1e6,
                     "reader_threads": threads, "input_generation_seconds": gen2}
            files_out = {}
            for pack in ("host", "device"):
                best = None
                for rep in range(2):
                    w, s2 = run_cuking(big, d / f"big_{pack}", threads, f"--pack={pack}",
                                       "--kin_threshold=0.05")
                    if best is None or s2["read_pack_seconds"] < best[1]["read_pack_seconds"]:
                        best = (w, s2)
                files_out[pack] = (d / f"big_{pack}" / "part-00000.snappy.parquet").read_bytes()
                point[pack + "_pack"] = {k: best[1][k] for k in
                                         ("read_pack_seconds", "triples_per_second",
                                          "decode_thread_seconds", "pack_thread_seconds",
                                          "kernel_seconds", "decode_tasks")}
                point[pack + "_pack"]["wall_seconds"] = best[0]
            if files_out["host"] != files_out["device"] and not args.no_check:
                raise SystemExit("PARITY FAILURE: host-packed and device-packed outputs differ")
            point["checks"] = "host-packed and device-packed output files identical"
            out["larger_point"] = point
        else:
            out["larger_point"] = None
            out["larger_point_note"] = (f"skipped: {cpus} hardware threads visible (generating and "
                                        "decoding 1e8 triples is sized for at least 8)")
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def visible_gpus(timeout=120.0):
    """GPUs HIP shows to this environment, counted in a CHILD process (ctypes ->
    libcuking_amd.so -> hipGetDeviceCount): the caller must not initialise the GPU
    itself (it is about to start the ranks), and importing torch for a number can
    take a minute on a fresh box."""
    import subprocess
    lib = ROOT / "cuking_amd" / "libcuking_amd.so"
    if not lib.exists():
        raise SystemExit(f"bench.py: {lib} has not been built (python -m cuking_amd.build)")
    code = ("import ctypes, sys; "
            f"sys.stdout.write(str(ctypes.CDLL({str(lib)!r}).cuking_device_count()))")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                           timeout=timeout)
        return int(r.stdout.strip() or 0) if r.returncode == 0 else 0
    except (subprocess.TimeoutExpired, ValueError):
        return 0


def self_launch(args):
    """`python bench.py --gpus N` started like the one-GPU command (no torchrun around
    it): this process NEVER touches the GPU -- it counts the devices in a child, then
    starts the N ranks as a child `python -m torch.distributed.run ... bench.py <same
    flags>` (one rank per GPU over RCCL; with CUKING_BENCH_REHEARSAL=1 all ranks on
    cuda:0 over gloo), relays what the ranks print -- rank 0's JSON line -- and exits
    with the launcher's code.  Nothing is ever exec'ed from a process that has
    initialised the GPU."""
    import signal
    import socket
    import subprocess
    n = args.gpus
    rehearsal = os.environ.get("CUKING_BENCH_REHEARSAL") == "1"
    seen = visible_gpus()
    need = 1 if rehearsal else n
    if seen < need:
        sys.stderr.write(f"bench.py: {n} GPUs requested, {seen} visible"
                         + (" (rehearsal: all ranks share cuda:0)" if rehearsal else "") + "\n")
        return 2
    if rehearsal and n > 6:
        sys.stderr.write("bench.py: a rehearsal puts every rank on cuda:0; at most 6 ranks\n")
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")  # (torchrun sets it anyway; said here, not warned about)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(Path(__file__).resolve()), *sys.argv[1:]]
    limit = float(os.environ.get("CUKING_BENCH_LAUNCH_TIMEOUT", "0") or 0)
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=limit if limit > 0 else None)
    except subprocess.TimeoutExpired:
        sys.stderr.write(f"bench.py: the {n} ranks did not finish within {limit:.0f} s; "
                         "stopping them\n")
    except KeyboardInterrupt:
        pass
    # (the process group this function started, nothing else)
    for sig in (signal.SIGTERM, signal.SIGKILL):
        try:
            os.killpg(proc.pid, sig)
        except ProcessLookupError:
            break
        try:
            proc.wait(timeout=10)
            break
        except subprocess.TimeoutExpired:
            continue
    return 1


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # ONE JSON line on stdout: libraries that write to file descriptor 1 (RCCL
    # prints a version banner there) are sent to stderr; the line goes to the
    # real stdout.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist
    import cuking_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (started by "
                         "torch.distributed.run with another --nproc-per-node?)")
    # Rehearsal on a one-GPU box: CUKING_BENCH_REHEARSAL=1 puts every rank on
    # cuda:0 and uses gloo for the collectives (RCCL refuses two ranks on one
    # device).  Never the measured configuration.
    rehearsal = os.environ.get("CUKING_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():   # (counting does not initialise the GPU)
        raise SystemExit(f"bench.py: {args.gpus} GPUs requested, "
                         f"{torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    # CUKING_BENCH_FORCE_DIST=1: run the multi-GPU code path (process group,
    # collectives) even with one rank -- a one-GPU check of the RCCL calls.
    force_dist = os.environ.get("CUKING_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))

    # The SAME workload behind `value` at every N (strong scaling): configs[2],
    # the largest configuration BASELINE.json labels 1 x MI355X.
    config = args.config or "c2"
    if config == "weak":
        n, m, thr = int(round(10000 * math.sqrt(world))), 100_000, 0.05
        workload_name = (f"weak scaling series: {n} samples x {m} sites "
                         f"(10000 sqrt(N) samples), kin-threshold {thr}")
        scaling = "weak"
    else:
        c = CONFIGS[config]
        n, m, thr, workload_name = c["samples"], c["sites"], c["thr"], c["name"]
        scaling = "strong"
    n = args.samples or n
    m = args.sites or m
    thr = args.kin_threshold if args.kin_threshold is not None else thr
    custom = bool(args.samples or args.sites or args.kin_threshold is not None)
    if custom:
        workload_name = f"custom: {n} samples x {m} sites, kin-threshold {thr}"

    ctx = cuking_amd.KingContext(local_rank)
    ctx.set_kernel(args.kernel)
    if args.variant >= 0:
        ctx.set_option("variant", args.variant)
    if args.band_rows > 0:
        ctx.set_option("band_rows", args.band_rows)
    if args.counts_mode >= 0:
        ctx.set_option("counts_mode", args.counts_mode)
    if args.xcd_swizzle >= 0:
        ctx.set_option("xcd_swizzle", args.xcd_swizzle)
    if args.split_wgs >= 0:
        ctx.set_option("split_wgs", args.split_wgs)
    if args.max_launch_blocks >= 0:
        ctx.set_option("max_launch_blocks", args.max_launch_blocks)
    for kv in args.option:
        key, _, value = kv.partition("=")
        ctx.set_option(key, int(value))
    # The cohort is not rewritten between steps: the kernel-internal layout is
    # converted by the first call and reused by the others (--convert-every-step
    # restores one conversion per step; it is reported either way).
    reuse = args.reuse_layout
    ctx.set_option("reuse_prepared", 1 if reuse else 0)
    prepared_note = ("kernel-internal layout converted once per cohort and reused by every step "
                     "(--reuse-layout, cuking option reuse_prepared: the bitset is not rewritten "
                     "between steps); roofline.prepare_ms is that one conversion" if reuse else
                     "every timed step converts the bitset into the kernel-internal layout "
                     "(roofline.prepare_ms, inside ms_per_step and `value`), as a one-pass job "
                     "does; value_layout_reused is the rate with the layout converted once")
    ctx.timing_enable(True)
    dtype_of = lambda roof: ("fp4 products, f32 accumulate (exact integers)"
                             if roof["bound"] == "mfma" else "u32")
    extras = args.extra_configs
    if extras is None:
        if config == "c2" and not custom:
            extras = "c1,c3,c4" if (use_dist and world >= 4) else "c1,c3" if use_dist else "c0,c1,c3"
        else:
            extras = "none"
    extra_keys_c0 = [k for k in extras.split(",") if k == "c0"]
    extra_keys = [k for k in extras.split(",") if k and k not in ("none", "c0")]
    # (steps, warm-up) of the configurations carried beside the headline
    extra_plan = {"c1": (20, 2), "c2": (4, 1), "c3": (2, 1), "c4": (1, 1)}

    # ------------------------------------------------------------------ N = 1
    if not use_dist:
        r = single_gpu_workload(args, ctx, n, m, thr, args.steps, args.warmup, local_rank,
                                h2d_pass=True)
        roofline = roofline_block(
            args, ctx, launch_pairs=r["pairs"], sites=m, wps=r["wps"], thr=thr,
            king_ms=r["king_ms"], prepare_ms=r["prepare_ms"], launches=r["launches"],
            workload_key=f"{n}x{m}", clock_mhz=r["clock_mhz"])
        if r["filter"] is not None:
            roofline["filter"] = r["filter"]
            if not custom and config == "c2" and not args.no_worst_case:
                roofline["filter_worst_case"] = filter_worst_case(args, ctx, r["bits"], n, m, thr,
                                                                  local_rank)
        out = {
            "metric": "sample-pairs/s (all-pairs KING)",
            "value": r["pairs"] * args.steps / r["elapsed"],
            "unit": "sample-pairs/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": r["elapsed"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": dtype_of(roofline), "data": "synthetic",
            "config": {"workload": workload_name, "samples": n, "sites": m,
                       "pairs": r["pairs"], "kin_threshold": thr,
                       "results_per_step": int(len(r["recs"])), "kernel": args.kernel,
                       "parallelism": "pair-space tiles over 1 GPU",
                       "prepared_layout": prepared_note,
                       "scaling_note": "the same workload is behind `value` at every N "
                                       "(strong scaling: the pair space is cut over the ranks)"},
            "roofline": roofline,
        }
        if r["reused_s_per_step"] is not None:
            out["value_layout_reused"] = {
                "value": r["pairs"] / r["reused_s_per_step"], "unit": "sample-pairs/s",
                "ms_per_step": r["reused_s_per_step"] * 1e3,
                "what": "the same steps with the kernel-internal layout converted once per "
                        "cohort and reused (library option reuse_prepared); round 3 quoted "
                        "this figure as `value`"}
        if r["h2d_ms"] is not None:
            ms = r["elapsed"] / args.steps * 1e3 + r["h2d_ms"]
            out["value_including_h2d"] = {
                "value": r["pairs"] / (ms * 1e-3), "unit": "sample-pairs/s", "ms_per_step": ms,
                "h2d_ms": r["h2d_ms"],
                "h2d_GBps": r["bits"].numel() * 8 / (r["h2d_ms"] * 1e-3) / 1e9,
                "what": "ms_per_step + one host-to-device copy of the packed bitset from "
                        "page-locked memory (HIP events, best of three); never `value`"}
        if args.cpu_seconds > 0:
            bits = r["bits"]

            def host_bits(s):
                return np.ascontiguousarray(bits[:s].cpu().numpy().view(np.uint64))
            out["cpu_baseline"] = cpu_baseline(host_bits, r["wps"], r["recs"], thr,
                                               args.cpu_seconds, n, args.cpu_threads)
        else:
            out["cpu_baseline"] = None
        del r
        torch.cuda.empty_cache()

        others = {}
        for key in extra_keys:
            c = CONFIGS[key]
            k, w = extra_plan[key]
            if c["samples"] > 100_000:
                k, w = 1, 1      # (one warm-up pass: the first call allocates the workspace)
            e = single_gpu_workload(args, ctx, c["samples"], c["sites"], c["thr"], k, w,
                                    local_rank)
            roof = roofline_block(
                args, ctx, launch_pairs=e["pairs"], sites=c["sites"], wps=e["wps"],
                thr=c["thr"], king_ms=e["king_ms"], prepare_ms=e["prepare_ms"],
                launches=e["launches"], workload_key=f"{c['samples']}x{c['sites']}",
                clock_mhz=e["clock_mhz"])
            if e["filter"] is not None:
                roof["filter"] = e["filter"]
            others[key] = {
                "workload": c["name"] + ", on ONE GPU", "value": e["pairs"] * k / e["elapsed"],
                "unit": "sample-pairs/s", "steps": k, "warmup": w,
                "ms_per_step": e["elapsed"] / k * 1e3, "pairs": e["pairs"],
                "value_layout_reused": (e["pairs"] / e["reused_s_per_step"]
                                        if e["reused_s_per_step"] else None),
                "results_per_step": int(len(e["recs"])),
                "bitset_GB": e["bits"].numel() * 8 / 1e9,
                "checks": "planted relatives all reported",
                "roofline": {kk: roof[kk] for kk in
                             ("bound", "achieved", "peak", "unit", "frac", "kernel_ms",
                              "prepare_ms", "sustained_clock_mhz", "form", "kernel",
                              "macs_per_pair_site", "filter")
                             if kk in roof},
            }
            del e
            torch.cuda.empty_cache()
        if "c0" in extra_keys_c0:
            others["c0"] = real_input_path(args, ctx, local_rank)
        if others:
            out["other_configs"] = others
        print(json.dumps(out), file=json_out, flush=True)
        return

    # ------------------------------------------------------------------ N > 1
    env = dict(world=world, rank=rank, local_rank=local_rank, dev=dev,
               host_or_dev="cpu" if rehearsal else dev)
    h = dist_workload(args, env, ctx, config, n, m, thr, args.steps, args.warmup, headline=True)
    others = {}
    for key in extra_keys:
        c = CONFIGS[key]
        k, w = extra_plan[key]
        e = dist_workload(args, env, ctx, key, c["samples"], c["sites"], c["thr"], k, w,
                          headline=False)
        if e is not None:
            others[key] = {
                "workload": c["name"] + f", over {world} GPU(s)", "value": e["value"],
                "unit": "sample-pairs/s", "steps": k, "warmup": w,
                "ms_per_step": e["elapsed"] / k * 1e3, "pairs": e["pairs"],
                "results_per_step": int(len(e["recs"])),
                "bitset_GB_per_rank": e["bitset_bytes"] / 1e9,
                "checks": "planted relatives all reported; records identical to the "
                          "single-GPU pass",
                "single_gpu_same_run": e["single"], "speedup": e["speedup"],
                "rank_kernel_ms_per_step": e["rank_kernel_ms"],
                "tile_range_balance": e["balance"],
                "roofline": {kk: e["roofline"][kk] for kk in
                             ("bound", "achieved", "peak", "unit", "frac", "kernel_ms",
                              "prepare_ms", "form", "note_rank") if kk in e["roofline"]},
            }
    out = None
    if rank == 0:
        out = {
            "metric": "sample-pairs/s (all-pairs KING)", "value": h["value"],
            "unit": "sample-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": h["elapsed"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": dtype_of(h["roofline"]), "data": "synthetic",
            "config": {"workload": workload_name, "samples": n, "sites": m,
                       "pairs": h["pairs"], "kin_threshold": thr,
                       "results_per_step": int(len(h["recs"])), "kernel": args.kernel,
                       "parallelism": (f"pair-space tiles over {world} GPU(s), "
                                       + ("bitset resident on every GPU, records gathered on "
                                          "rank 0 (gather pipelined behind the next pass)"
                                          if args.dist_mode == "resident" else
                                          f"{args.dist_mode} bitset broadcast inside every step")),
                       "prepared_layout": prepared_note,
                       "scaling_note": "the same workload is behind `value` at every N "
                                       "(strong scaling: the pair space is cut over the ranks)",
                       "rccl_ranks": world, "backend": dist.get_backend(),
                       "rank_kernel_ms_per_step": h["rank_kernel_ms"],
                       "rank_prepare_ms_per_step": h["rank_prepare_ms"],
                       "gather_ms_unpipelined": h["gather_ms"],
                       "tile_range_balance": h["balance"],
                       "bitset_bytes_per_rank": h["bitset_bytes"]},
            "with_broadcast": h["with_broadcast"],
            "single_gpu_same_run": h["single"],
            "speedup": h["speedup"],
            "roofline": h["roofline"],
            "cpu_baseline": None,
        }
        if others:
            out["other_configs"] = others
    dist.barrier()
    dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), file=json_out, flush=True)


if __name__ == "__main__":
    main()
