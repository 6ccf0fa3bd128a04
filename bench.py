#!/usr/bin/env python3
"""All-pairs KING throughput on MI355X (BASELINE.json metric: sample-pairs/s +
achieved HBM GB/s vs roofline).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one synthetic cohort whose packed
bitset is already resident in HBM (reference layout, cuking.cu:507-523):
layout preparation + the pair kernel (matrix-core variant by default) over
every (i < j) pair + thresholded append of KingResult records.  For N > 1 a
step is the sharded pass of cuking_amd/dist.py: every rank holds the bitset
(as every shard of the reference reads the whole input itself), evaluates its
range of pair-space tiles, and the records are gathered on rank 0 -- the only
collective; the gather of one pass runs behind the kernel of the next.  --dist-mode staged / simple instead start from a bitset that only
rank 0 holds and count its RCCL broadcast in the step.

N = 1 workload: BASELINE.json configs[1], 10k samples x 100k sites,
kin-threshold 0.05.  N > 1 (weak scaling): the same sites and threshold with
round(10000 * sqrt(N)) samples, i.e. the same number of pairs per GPU.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# MI355X_MICROARCH.md, matrix cores: FP4 (block-scaled f8f6f4 MFMA) ~10 PF dense.
MFMA_FP4_PEAK_TFLOPS = 10000.0
# king_mfma.hip: five plane products per pair and site (opp = A.R + R.A, bh,
# hi, hj), one multiply-add = 2 FLOP.
MFMA_MACS_PER_PAIR_SITE = {"lean": 5, "full": 6}
MFMA_VARIANT = 5
NOMINAL_CLOCK_HZ = 2.4e9        # MI355X_MICROARCH.md: max clock
NUM_SIMDS = 256 * 4
# king_kernels.hip, per pair per 32 sites: lean form 5 logic + 4 v_bcnt (used
# when kin_threshold > 0), full form 5 + 5.
VALU_OPS_PER_PAIR_WORD = {"lean": 9, "full": 10}
# VALU issue floor, measured on MI355X (tools/micro/king_step.hip, valu_phase.hip;
# profiles/r01_valu_microbench.txt), cycles per wave64 instruction per SIMD:
# v_and 2.07, v_bitop3 2.37, v_bcnt_u32_b32 4.19 when each kind runs alone.  The
# phased kernel (logic phase / popcount phase, waves of a workgroup paired per
# SIMD) is priced against the SUM of its parts: 4 x 2.07 + 2.37 + n_bcnt x 4.19.
# (Unsynchronised waves mixing the kinds cost ~4.06 per instruction: 36.5 / 40.7.)
VALU_FLOOR_CYCLES_PER_PAIR_WORD = {"lean": 4 * 2.07 + 2.37 + 4 * 4.19,
                                   "full": 4 * 2.07 + 2.37 + 5 * 4.19}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--samples", type=int, default=0, help="override N samples")
    ap.add_argument("--sites", type=int, default=100000)
    ap.add_argument("--kin-threshold", type=float, default=0.05)
    ap.add_argument("--max-results", type=int, default=1 << 20)
    ap.add_argument("--kernel", default="tiled", choices=["tiled", "stream"])
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--band-rows", type=int, default=-1)
    ap.add_argument("--counts-mode", type=int, default=-1, choices=[-1, 0, 1],
                    help="-1 automatic, 0 lean form (4 sums + recount of emitted pairs), "
                         "1 full form (5 sums)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="target CPU-baseline time (0 disables it)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="OpenMP threads of the CPU baseline (0 = min(16, available))")
    ap.add_argument("--dist-mode", default="resident",
                    choices=["resident", "staged", "simple"],
                    help="N>1: the packed bitset is resident on every GPU before the "
                         "timed region, like the reference's shards that each read the "
                         "input themselves (resident, default); or rank 0 owns it and "
                         "every step distributes it first: chunked broadcast overlapped "
                         "with compute (staged) / broadcast then compute (simple)")
    ap.add_argument("--chunks", type=int, default=8, help="broadcast chunks (staged)")
    ap.add_argument("--streams", type=int, default=3, help="side streams for rectangle launches")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the planted-relatives check (timing-only tuning kernels)")
    ap.add_argument("--seed", type=int, default=20240229)
    return ap.parse_args()


def cpu_baseline(host_bits_fn, wps, gpu_records, thr, target_seconds, max_samples,
                 cpu_threads=0):
    """Times the oracle (oracle/king_oracle.c, -march=native, OpenMP over rows)
    on a leading sub-block of the SAME cohort, and checks the GPU's records for
    that sub-block against it.  Test-infrastructure use only."""
    import tempfile
    import numpy as np
    from oracle import pyoracle
    # The GPU box's CPU share for one GPU is 16 hardware threads (its 8 GPUs
    # share the host); --cpu-threads overrides.
    threads = cpu_threads or min(16, len(os.sched_getaffinity(0)))
    out_dir = Path(tempfile.mkdtemp(prefix="cuking_oracle_"))
    lib = pyoracle.load(native=True, out_dir=out_dir)
    os.environ.setdefault("OMP_PROC_BIND", "true")

    def run(s):
        import ctypes as C
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

    pairs, dt, _ = run(256)                      # calibration
    rate = pairs / dt
    s = int(min(max_samples, max(256, math.sqrt(2 * rate * target_seconds))))
    pairs, dt, res = run(s)
    sel = gpu_records[(gpu_records["sample_j"] < s)]
    if sel.tobytes() != res.tobytes():
        raise SystemExit(f"PARITY FAILURE: GPU records for the first {s} samples "
                         "differ from the CPU oracle")
    cpu_model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": pairs / dt, "unit": "sample-pairs/s", "cores": threads,
            "kind": "port",
            "sample": f"first {s} samples ({pairs} pairs) of the same cohort, "
                      f"{dt:.1f} s, OpenMP x{threads} on {cpu_model} "
                      f"({len(os.sched_getaffinity(0))} hardware threads visible), "
                      "-O3 -march=native; records checked equal to the GPU's"}


def load_traffic(workload_key, kernel_name):
    """HBM bytes per launch from committed rocprofv3 PMC passes (profiles/),
    corrected as MI355X_MICROARCH.md prescribes; None if not measured for this
    workload and kernel."""
    p = ROOT / "profiles" / "hbm_traffic.json"
    if not p.exists():
        return None
    try:
        entry = json.loads(p.read_text()).get(f"{workload_key}:{kernel_name}", {})
        return entry.get("traffic_bytes_per_launch")
    except Exception:
        return None


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import cuking_amd
    from cuking_amd.dist import (GpuStagedOps, PipelinedGather, all_pairs_king,
                                 all_pairs_king_staged, rank_tile_share,
                                 tile_partition)
    from cuking_amd.synth import cohort_to_device, plan_cohort

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}; launch with "
                         "torch.distributed.run --nproc-per-node N")
    # Rehearsal on a one-GPU box: CUKING_BENCH_REHEARSAL=1 puts every rank on
    # cuda:0 and uses gloo for the collectives (RCCL refuses two ranks on one
    # device).  Never the measured configuration.
    rehearsal = os.environ.get("CUKING_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
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

    n = args.samples or int(round(10000 * math.sqrt(world)))
    m = args.sites
    thr = args.kin_threshold
    wps = cuking_amd.words_per_sample(m)
    sm = cuking_amd.Submatrix(n)
    pairs = sm.NumPairs()

    ctx = cuking_amd.KingContext(local_rank)
    ctx.set_kernel(args.kernel)
    if args.variant >= 0:
        ctx.set_option("variant", args.variant)
    if args.band_rows > 0:
        ctx.set_option("band_rows", args.band_rows)
    if args.counts_mode >= 0:
        ctx.set_option("counts_mode", args.counts_mode)
    ctx.timing_enable(True)

    # Synthetic cohort, generated on the device (rank 0 owns the "packed
    # input"; the other ranks receive it by broadcast inside every step).
    cohort = plan_cohort(n, args.seed)
    kind, pa, pb = cohort_to_device(cohort, local_rank)
    bits = torch.zeros((n, wps), dtype=torch.int64, device=dev)
    if rank == 0 or args.dist_mode == "resident":
        ctx.synth_bitset(args.seed, kind, pa, pb, 0, n, m, out=bits)
    torch.cuda.synchronize()

    results = torch.zeros((args.max_results, 6), dtype=torch.int32, device=dev)
    index_flag = torch.zeros(2, dtype=torch.int32, device=dev)
    # second record buffer: with nccl the gather of one pass overlaps the
    # kernel of the next (cuking_amd.dist.PipelinedGather)
    results_b = torch.zeros_like(results) if use_dist else None
    index_flag_b = torch.zeros_like(index_flag) if use_dist else None
    num_tiles = ctx.num_tiles(sm) if args.kernel == "tiled" else 0
    my_tiles = tile_partition(num_tiles, world)[rank] if use_dist else None

    def compute_tiles(bit_sets, begin, end):
        index_flag.zero_()
        ctx.compute_king(sm, wps, bit_sets, thr, args.max_results, results,
                         index_flag[0:1], index_flag[1:2], tile_range=(begin, end))
        if device_gather:
            return results, index_flag     # counts stay on the device until the gather
        count, ovf = index_flag.tolist()   # waits for the kernel
        return results, count, ovf

    # nccl gathers straight from the kernel's own counters; gloo (one-GPU
    # rehearsals) needs host tensors and takes the staging path
    device_gather = use_dist and dist.get_backend() == "nccl"
    gathered = [None]
    staged = use_dist and args.dist_mode == "staged" and args.kernel == "tiled"
    tile = ctx.tile_samples()
    staged_ops = (GpuStagedOps(ctx, sm, wps, bits, thr, args.max_results,
                               num_streams=args.streams)
                  if staged else None)

    pipelined = (use_dist and device_gather and args.dist_mode == "resident" and
                 args.kernel == "tiled" and
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
        ctx.compute_king(sm, wps, bits, thr, args.max_results, buf, flag[0:1], flag[1:2],
                         tile_range=my_tiles)
        handle = pipe.begin(buf, flag)
        if pending[0] is not None:
            gathered[0] = pipe.finish(pending[0])
        pending[0] = handle

    def drain():
        if pending[0] is not None:
            gathered[0] = pipe.finish(pending[0])
            pending[0] = None

    def step():
        if pipelined:
            return pipelined_step()
        if not use_dist:
            index_flag.zero_()
            ctx.compute_king(sm, wps, bits, thr, args.max_results, results,
                             index_flag[0:1], index_flag[1:2])
        elif staged:
            gathered[0], _ = all_pairs_king_staged(staged_ops, n, tile, bits,
                                                   num_chunks=args.chunks)
        else:
            gathered[0], _ = all_pairs_king(compute_tiles, num_tiles, bits,
                                            broadcast=args.dist_mode != "resident")

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if pipelined:
        drain()
    barrier()
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if pipelined:
        drain()          # the last pass's records are on rank 0 before the clock stops
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    timing = ctx.timing_collect()
    # Records of the last step (rank 0): sanity + parity material.
    if not use_dist:
        count, ovf = index_flag.tolist()
        if ovf:
            raise SystemExit("result overflow: raise --max-results")
        recs = results[:count].cpu().numpy().view(np.uint32).reshape(-1).view(
            cuking_amd.KING_RESULT_DTYPE).copy()
        recs = cuking_amd.sort_results(recs)
    else:
        recs = gathered[0]

    out = None
    if rank == 0:
        got = {(int(r["sample_i"]), int(r["sample_j"])) for r in recs}
        missing = [p for p in cohort.planted
                   if (min(p[0], p[1]), max(p[0], p[1])) not in got]
        if missing and not args.no_check:
            raise SystemExit(f"{len(missing)} planted relatives not reported")

        ms_per_step = elapsed / args.steps * 1e3
        value = pairs * args.steps / elapsed
        bpp = cuking_amd.bytes_per_pair(wps)
        # Dominant kernel = the pair kernel; rank 0's launches cover its own
        # share of the pairs.
        if not use_dist:
            launch_pairs, launches = pairs, timing.king_launches
        else:
            # rank 0's share of the pairs per step; its kernel time per step is
            # the sum over its launches (rectangles on two streams may overlap,
            # so this is an upper bound on the time the pair kernel was busy)
            if staged:
                share = rank_tile_share((n + tile - 1) // tile, world, 0)
            else:
                share = (my_tiles[1] - my_tiles[0]) / max(num_tiles, 1)
            launch_pairs, launches = pairs * share, args.steps
        king_ms = timing.king_ms / max(launches, 1)
        achieved = launch_pairs * bpp / (king_ms * 1e-3) / 1e9 if king_ms > 0 else 0.0
        workload = f"{n} samples x {m} sites, kin-threshold {thr}"
        key = f"{n}x{m}"
        kernel_name = ("king_stream_kernel" if args.kernel == "stream" else
                       "king_mfma_kernel" if ctx.get_option("variant") == MFMA_VARIANT else
                       "king_tiled_kernel")
        variant = ctx.get_option("variant")
        form = ("full" if args.counts_mode == 1 else "lean" if args.counts_mode == 0 else
                "lean" if thr > 0 and thr * thr * 32 * wps >= (
                    1.9 ** 2 if args.kernel == "tiled" and variant == MFMA_VARIANT else 1.6 ** 2)
                else "full")
        hbm_view = {
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "algorithmic_bytes_per_pair": bpp,
            "note": "algorithmic bytes (2 samples x words_per_sample x 8 B per pair, "
                    "SURVEY.md 8d) / measured kernel time; operands are re-used from LDS and "
                    "registers, so this exceeds 1.0 of HBM peak by design",
        }
        common = {
            "traffic": load_traffic(key, kernel_name) if not use_dist else None,
            "kernel": kernel_name, "kernel_ms": king_ms, "launches": timing.king_launches,
            "prepare_ms": timing.prepare_ms / max(timing.prepare_launches, 1),
        }
        if args.kernel == "tiled" and variant == MFMA_VARIANT:
            # Matrix-core kernel: algorithmic FLOP = pairs x sites x plane
            # products x 2, against the dense FP4 MFMA peak.
            macs = MFMA_MACS_PER_PAIR_SITE[form]
            tflops = (launch_pairs * m * macs * 2 / (king_ms * 1e-3) / 1e12) if king_ms > 0 else 0.0
            roofline = {
                "bound": "mfma", "achieved": tflops, "peak": MFMA_FP4_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": tflops / MFMA_FP4_PEAK_TFLOPS, **common,
                "form": form, "macs_per_pair_site": macs,
                "note": "fp4 (E2M1) v_mfma_f32_32x32x64_f8f6f4, exact integer sums in f32; "
                        "algorithmic FLOP = pairs x sites x plane products x 2 (padding of "
                        "tiles and of the last k-step not counted); the chip holds ~2.1 GHz "
                        "under this load, where MFMAs alone reach ~9.3 PF "
                        "(profiles/r01_mfma_microbench.txt)",
                "hbm": hbm_view,
            }
        else:
            roofline = {"bound": "hbm", **hbm_view, **common}
            # VALU view: wave64 issue cycles one SIMD spends per pair and 32-site
            # word (at the nominal clock) against the measured floor for this mix.
            cyc = (king_ms * 1e-3 * NOMINAL_CLOCK_HZ * NUM_SIMDS * 64 /
                   (launch_pairs * wps)) if king_ms > 0 else 0.0
            floor = VALU_FLOOR_CYCLES_PER_PAIR_WORD[form]
            roofline["valu"] = {
                "form": form,
                "ops_per_pair_word": VALU_OPS_PER_PAIR_WORD[form],
                "achieved_cycles_per_pair_word": cyc,
                "floor_cycles_per_pair_word": floor,
                "frac": floor / cyc if cyc else 0.0,
            }
        # the arithmetic the sums are computed in
        dtype = ("fp4 products, f32 accumulate (exact integers)"
                 if roofline["bound"] == "mfma" else "u32")
        out = {
            "metric": "sample-pairs/s (all-pairs KING)", "value": value,
            "unit": "sample-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": workload, "samples": n, "sites": m,
                       "pairs": pairs, "kin_threshold": thr,
                       "results_per_step": int(len(recs)),
                       "kernel": args.kernel,
                       "parallelism": f"pair-space tiles over {world} GPU(s)"
                                      + ((", bitset resident on every GPU, records gathered"
                                          if args.dist_mode == "resident" else
                                          f", {args.dist_mode} bitset broadcast") if use_dist else "")},
            "roofline": roofline,
        }
        if not use_dist and args.cpu_seconds > 0:
            def host_bits(s):
                return np.ascontiguousarray(bits[:s].cpu().numpy().view(np.uint64))
            out["cpu_baseline"] = cpu_baseline(host_bits, wps, recs, thr, args.cpu_seconds,
                                               n, args.cpu_threads)
        else:
            out["cpu_baseline"] = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
