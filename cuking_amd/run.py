"""Multi-GPU driver with the reference's CLI surface (cuking.cu:27-52):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \\
        --master-addr 127.0.0.1 --master-port 29500 -m cuking_amd.run \\
        --input-uri in/ --output-uri out/ --kin-threshold 0.05

One process per GPU.  Rank 0 reads `metadata.json` + `*.parquet`, packs the
bitset through the C ABI (`cuking_pack_host`, reader threads like
cuking.cu:550-553), the bitset goes to the other GPUs by the staged RCCL
broadcast of `cuking_amd.dist`, every rank evaluates its band of the pair
space, rank 0 gathers, sorts and writes `part-<shard>.snappy.parquet` with the
reference's schema (cuking.cu:767-870).  `--split-factor/--shard-index` select a
block exactly like the reference; the GPUs of the node share that block's work
(this replaces the one-VM-per-shard fan-out of cloud_batch_submit.py:45,73).
With a single process (no torchrun) it is the one-GPU path.

The single-GPU drop-in without Python is the C++ binary `cuking_amd/bin/cuking`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np


def parse_args(argv=None):
    ap = argparse.ArgumentParser(prog="cuking_amd.run", allow_abbrev=False)

    def flag(name, **kw):  # accept --kin-threshold and --kin_threshold
        ap.add_argument(f"--{name}", f"--{name.replace('-', '_')}",
                        dest=name.replace("-", "_"), **kw)

    flag("input-uri", default="")
    flag("output-uri", default="")
    flag("requester-pays-project", default="")
    flag("num-reader-threads", type=int, default=36)
    flag("max-results", type=int, default=10 << 20)
    flag("kin-threshold", type=float, default=0.0884)
    flag("split-factor", type=int, default=1)
    flag("shard-index", type=int, default=0)
    flag("chunks", type=int, default=8)
    flag("variant", type=int, default=-1,
         help="tiled kernel variant (default: the library's, 7 = one-product filter + exact "
              "recount; 6 = four products for every pair, the choice when more than ~10 %% of "
              "the calls are missing; same records either way)")
    flag("synthetic", default="",
         help="N,M[,seed]: instead of reading --input-uri, generate the synthetic cohort of "
              "cuking_amd.synth on the GPU (BASELINE configs without their 10^9..10^11-row "
              "Parquet form)")
    return ap.parse_args(argv)


class UsageError(Exception):
    pass


def resolve_uri(uri: str) -> Path:
    if uri.startswith("gs://"):
        raise UsageError(f"Unsupported URI: {uri} (no GCS client in this build; "
                         "pass a local directory or file:// URI)")
    return Path(uri[7:] if uri.startswith("file://") else uri)


def validate(args):  # cuking.cu:437-462
    if not args.input_uri and not args.synthetic:
        raise UsageError("No input URI specified")
    if not args.output_uri:
        raise UsageError("No output URI specified")
    if args.num_reader_threads <= 0:
        raise UsageError("Invalid number of reader threads")
    if args.split_factor <= 0:
        raise UsageError("Invalid split factor")
    if not 0 <= args.shard_index < args.split_factor * (args.split_factor + 1) // 2:
        raise UsageError("Invalid shard index")


def read_and_pack(in_dir: Path, sm, num_sites: int, threads: int) -> np.ndarray:
    """cuking.cu:529-711: list, decode, pack (host)."""
    import pyarrow.parquet as pq
    import cuking_amd
    files = sorted(p for p in in_dir.iterdir()
                   if p.is_file() and p.name.endswith(".parquet"))  # non-recursive
    if not files:
        raise RuntimeError("No input files found")
    bits = cuking_amd.new_host_bitset(sm, num_sites)

    def one(path):
        pf = pq.ParquetFile(path)
        if pf.metadata.num_columns != 3:
            raise RuntimeError(f"Expected 3 columns, found {pf.metadata.num_columns} in {path}")
        t = pf.read()
        cols = [t.column(k) for k in range(3)]  # by position (cuking.cu:585-597)
        if cols[0].null_count or cols[1].null_count:
            raise RuntimeError(f"null values are not allowed in row_idx/col_idx in {path}")
        alt = cols[2]
        if alt.null_count:  # null genotype = missing: drop the entry
            keep = alt.is_valid().to_numpy(zero_copy_only=False)
        else:
            keep = None
        row = cols[0].to_numpy().astype(np.int64, copy=False)
        col = cols[1].to_numpy().astype(np.int64, copy=False)
        a = alt.fill_null(0).to_numpy().astype(np.int32, copy=False)
        if keep is not None:
            row, col, a = row[keep], col[keep], a[keep]
        cuking_amd.pack_host(sm, bits, row, col, a)  # thread-safe atomics

    with ThreadPoolExecutor(max(1, threads)) as ex:
        list(ex.map(one, files))
    return bits


def write_results(path: Path, recs: np.ndarray, sample_ids) -> None:
    """cuking.cu:767-863: REQUIRED columns, SNAPPY, one row group."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    ids = np.asarray(sample_ids, dtype=object)
    schema = pa.schema([pa.field("i", pa.string(), nullable=False),
                        pa.field("j", pa.string(), nullable=False),
                        pa.field("kin", pa.float32(), nullable=False),
                        pa.field("ibs0", pa.int32(), nullable=False),
                        pa.field("ibs1", pa.int32(), nullable=False),
                        pa.field("ibs2", pa.int32(), nullable=False)])
    table = pa.table({
        "i": pa.array(ids[recs["sample_i"]], pa.string()),
        "j": pa.array(ids[recs["sample_j"]], pa.string()),
        "kin": pa.array(recs["kin"], pa.float32()),
        "ibs0": pa.array(recs["ibs0"].astype(np.int32), pa.int32()),
        "ibs1": pa.array(recs["ibs1"].astype(np.int32), pa.int32()),
        "ibs2": pa.array(recs["ibs2"].astype(np.int32), pa.int32()),
    }, schema=schema)
    path.parent.mkdir(parents=True, exist_ok=True)
    pq.write_table(table, path, compression="snappy", use_dictionary=False,
                   row_group_size=max(len(recs), 1))


def main(argv=None) -> int:
    args = parse_args(argv)
    import torch
    import torch.distributed as dist
    import cuking_amd
    from cuking_amd.dist import GpuStagedOps, all_pairs_king_staged

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    try:
        validate(args)
        in_dir = resolve_uri(args.input_uri) if args.input_uri else None
        out_dir = resolve_uri(args.output_uri)
        synthetic = None
        if args.synthetic:
            parts = [int(x) for x in args.synthetic.split(",")]
            if len(parts) not in (2, 3) or min(parts[:2]) <= 0:
                raise UsageError("--synthetic expects N,M[,seed]")
            synthetic = (parts[0], parts[1], parts[2] if len(parts) == 3 else 20240229)
    except ValueError:
        if rank == 0:
            print("\nError: INVALID_ARGUMENT: --synthetic expects N,M[,seed]", file=sys.stderr)
        return 1
    except UsageError as e:
        if rank == 0:
            print(f"\nError: INVALID_ARGUMENT: {e}", file=sys.stderr)
        return 1
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(dev))
    try:
        t0 = time.perf_counter()
        if synthetic:
            sample_ids = [f"S{k:07d}" for k in range(synthetic[0])]
            num_sites = synthetic[1]
        else:
            meta = json.loads((in_dir / "metadata.json").read_text())
            sample_ids, num_sites = list(meta["samples"]), int(meta["num_sites"])
        sm = cuking_amd.Submatrix(len(sample_ids), args.split_factor, args.shard_index)
        wps = cuking_amd.words_per_sample(num_sites)
        ctx = cuking_amd.KingContext(local_rank)
        if args.variant >= 0:   # (every geometry call below goes through this context)
            ctx.set_option("variant", args.variant)
        stored = sm.NumSamples()
        bits = torch.zeros((max(stored, 1), wps), dtype=torch.int64, device=dev)
        # Rank 0 reads and packs; its outcome is agreed on before anybody
        # enters the data broadcast (a failed read must not hang the others).
        pack_error = None
        if rank == 0:
            try:
                if synthetic:
                    from cuking_amd.synth import cohort_to_device, plan_cohort
                    kind, pa, pb = cohort_to_device(plan_cohort(synthetic[0], synthetic[2]),
                                                    local_rank)
                    # the block's samples, rows first then columns (cuking.cu:171-175)
                    ctx.synth_bitset(synthetic[2], kind, pa, pb, sm.i_begin, sm.i_end,
                                     num_sites, out=bits[:sm.NumRows()])
                    if sm.i_begin != sm.j_begin:
                        ctx.synth_bitset(synthetic[2], kind, pa, pb, sm.j_begin, sm.j_end,
                                         num_sites, out=bits[sm.NumRows():stored])
                    torch.cuda.synchronize()
                else:
                    host = read_and_pack(in_dir, sm, num_sites, args.num_reader_threads)
                    if stored:
                        bits[:stored].copy_(torch.from_numpy(host.view(np.int64)))
                print(f"[cuking_amd.run] packed {stored} samples x {num_sites} sites "
                      f"({time.perf_counter() - t0:.2f}s)", flush=True)
            except Exception as e:  # noqa: BLE001 - reported below on every rank
                pack_error = e
        if world > 1:
            ok = torch.tensor([0 if pack_error is None else 1], dtype=torch.int32, device=dev)
            dist.broadcast(ok, src=0)
            if int(ok) != 0 and pack_error is None:
                pack_error = RuntimeError("rank 0 failed to read the input")
        if pack_error is not None:
            raise RuntimeError(str(pack_error))
        t1 = time.perf_counter()
        after_reserve = lambda: (0, 0)    # noqa: E731 - (allocations, host waits) since the reservation
        if sm.i_begin == sm.j_begin:
            # Diagonal block (the whole cohort when split_factor = 1): row bands
            # per rank, chunked broadcast overlapped with the kernel.
            local = cuking_amd.Submatrix.from_ranges(0, stored, 0, stored)
            ops = GpuStagedOps(ctx, local, wps, bits, args.kin_threshold, args.max_results)
            recs, _ = all_pairs_king_staged(ops, stored, ctx.tile_samples(), bits,
                                            num_chunks=args.chunks)
            after_reserve = ops.after_reserve
            if recs is not None:  # local -> global sample indices
                recs["sample_i"] += sm.i_begin
                recs["sample_j"] += sm.j_begin
        else:
            # Off-diagonal block: broadcast, then equal tile ranges per rank.
            from cuking_amd.dist import all_pairs_king
            results = torch.zeros((max(args.max_results, 1), 6), dtype=torch.int32, device=dev)
            index_flag = torch.zeros(2, dtype=torch.int32, device=dev)

            def compute_tiles(b, begin, end):
                index_flag.zero_()
                ctx.compute_king(sm, wps, b, args.kin_threshold, args.max_results, results,
                                 index_flag[0:1], index_flag[1:2], tile_range=(begin, end))
                count, ovf = (int(x) & 0xFFFFFFFF for x in index_flag.tolist())
                return results, min(count, args.max_results), int(ovf)

            if world > 1:
                # workspace before the first collective (cuking_ctx_reserve), as in the
                # staged pass
                ctx.reserve(sm, wps, [torch.cuda.current_stream()])
                at = (ctx.get_option("workspace_allocations"), ctx.get_option("host_syncs"))
                after_reserve = lambda: (ctx.get_option("workspace_allocations") - at[0],  # noqa: E731
                                         ctx.get_option("host_syncs") - at[1])
                recs, _ = all_pairs_king(compute_tiles, ctx.num_tiles(sm), bits)
            else:
                recs = ctx.run(sm, wps, bits, args.kin_threshold, args.max_results)
        # every rank's library-side allocations / host waits after its reservation
        # (must be 0: nothing blocks beside in-flight collectives)
        mine = torch.tensor(after_reserve(), dtype=torch.int64, device=dev)
        per_rank = [mine]
        if world > 1:
            per_rank = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(per_rank, mine)
        per_rank = [t.tolist() for t in per_rank]
        if rank == 0:
            dt = time.perf_counter() - t1
            out = out_dir / f"part-{args.shard_index:05d}.snappy.parquet"
            write_results(out, recs, sample_ids)
            pairs = sm.NumPairs()
            print(json.dumps({"pairs": pairs, "results": int(len(recs)), "gpus": world,
                              "compute_seconds": dt,
                              "allocations_after_reserve": [int(x[0]) for x in per_rank],
                              "host_syncs_after_reserve": [int(x[1]) for x in per_rank],
                              "pairs_per_second": pairs / dt if dt > 0 else 0.0}), flush=True)
        rc = 0
    except cuking_amd.ResourceExhaustedError as e:
        if rank == 0:
            print(f"\nError: RESOURCE_EXHAUSTED: {e}", file=sys.stderr)
        rc = 1
    except (RuntimeError, cuking_amd.CukingError, OSError, KeyError, ValueError) as e:
        if rank == 0:
            print(f"\nError: FAILED_PRECONDITION: {e}", file=sys.stderr)
        rc = 1
    if world > 1 and dist.is_initialized():
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
