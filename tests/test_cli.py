"""The `cuking` binary (C++ host): flags, metadata, Parquet decode + pack on CPU;
the full Parquet-in -> kernel -> Parquet-out path on the GPU (BASELINE.json
configs[0]: 1k samples x 10k sites through real Parquet files)."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

import cuking_amd
from cuking_amd import build as cbuild
from cuking_amd.inputs import read_results, write_input_tables
from conftest import random_genotypes

CLI = cbuild.CLI_PATH


def run_cli(*args, check=False):
    p = subprocess.run([str(CLI), *map(str, args)], capture_output=True, text=True,
                       timeout=600)
    if check and p.returncode != 0:
        raise AssertionError(f"cuking failed ({p.returncode}):\n{p.stdout}\n{p.stderr}")
    return p


@pytest.fixture(scope="module", autouse=True)
def built_cli():
    cbuild.build_library()
    cbuild.build_cli()
    assert CLI.exists()


# ---------------------------------------------------------------- flags ----
def test_flag_validation_messages(tmp_path):
    """cuking.cu:437-462 (messages) and :889-892 (stderr + exit code 1)."""
    cases = [
        ([], "No input URI specified"),
        (["--input_uri", tmp_path], "No output URI specified"),
        (["--input_uri", tmp_path, "--output_uri", tmp_path, "--num_reader_threads=0"],
         "Invalid number of reader threads"),
        (["--input_uri", tmp_path, "--output_uri", tmp_path, "--split_factor=0"],
         "Invalid split factor"),
        (["--input-uri", tmp_path, "--output-uri", tmp_path, "--split-factor=4",
          "--shard-index=10"], "Invalid shard index"),
        (["--input_uri=gs://bucket/in", "--output_uri", tmp_path], "Unsupported URI"),
    ]
    for args, msg in cases:
        p = run_cli(*args)
        assert p.returncode == 1, args
        assert "Error: INVALID_ARGUMENT: " in p.stderr and msg in p.stderr, p.stderr
    for bad in ("--synthetic=abc", "--synthetic=10", "--synthetic=0,5", "--synthetic=4,5,6,7"):
        p = run_cli(bad, "--output_uri", tmp_path)
        assert p.returncode == 1 and "Illegal value" in p.stderr and "synthetic" in p.stderr, bad
    p = run_cli("--synthetic=10,20", "--output_uri", tmp_path, "--dump_bitset", tmp_path / "b")
    assert p.returncode == 1 and "--dump_bitset needs input tables" in p.stderr
    p = run_cli("--no_such_flag=1")
    assert p.returncode == 1 and "Unknown command line flag" in p.stderr
    p = run_cli("--kin_threshold=abc")
    assert p.returncode == 1 and "Illegal value" in p.stderr
    assert run_cli("--help").returncode == 0


def test_missing_inputs(tmp_path):
    out = tmp_path / "out"
    p = run_cli("--input_uri", tmp_path / "nope", "--output_uri", out,
                "--dump_bitset", tmp_path / "b.bin")
    assert p.returncode == 1 and "Failed to read metadata" in p.stderr
    empty = tmp_path / "empty"
    empty.mkdir()
    (empty / "metadata.json").write_text('{"num_sites": 3, "samples": ["a"]}')
    p = run_cli("--input_uri", empty, "--output_uri", out, "--dump_bitset", tmp_path / "b.bin")
    assert p.returncode == 1 and "No input files found" in p.stderr  # :542-544
    (empty / "metadata.json").write_text('{"num_sites": 3, "samples": ["a"')
    p = run_cli("--input_uri", empty, "--output_uri", out, "--dump_bitset", tmp_path / "b.bin")
    assert p.returncode == 1 and "Failed to parse metadata JSON" in p.stderr  # :485-487


def test_metadata_json_variants(tmp_path, oracle):
    """metadata.json as other writers may produce it: extra members, nesting,
    escapes, \\u sequences, whitespace, any member order."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    d = tmp_path / "in"
    d.mkdir()
    ids = ["a\"b", "tab\there", "é", "\U0001F9EC", "plain"]
    text = ('{ "extra": {"nested": [1, 2.5e3, true, null, {"x": "}"}]},\n'
            '  "samples" : ["a\\"b", "tab\\there", "\\u00e9", "\\ud83e\\uddec", "plain"],\n'
            '  "num_sites":70 , "z": -1.5 }')
    assert json.loads(text)["samples"] == ids
    (d / "metadata.json").write_text(text)
    pq.write_table(pa.table({"row_idx": pa.array([69, 0], pa.int64()),
                             "col_idx": pa.array([4, 1], pa.int64()),
                             "n_alt_alleles": pa.array([2, 1], pa.int32())}), d / "t.parquet")
    got = dump_bits(d, tmp_path, 5, cuking_amd.words_per_sample(70))
    geno = np.full((5, 70), -1, dtype=np.int8)
    geno[4, 69], geno[1, 0] = 2, 1
    assert np.array_equal(got, oracle.bitset_from_genotypes(geno))
    for bad in ('{"samples": ["a"]}', '{"num_sites": 3}', '{"num_sites": -1, "samples": []}',
                '{"num_sites": 1.5, "samples": []}', '{"num_sites": 3, "samples": [1]}',
                '{"num_sites": 3, "samples": ["a"]} x', '[]', ''):
        (d / "metadata.json").write_text(bad)
        p = run_cli("--input_uri", d, "--output_uri", tmp_path / "o", "--dump_bitset",
                    tmp_path / "x.bin")
        assert p.returncode == 1 and "metadata" in p.stderr, (bad, p.stderr)


def test_host_code_under_asan_ubsan(tmp_path, oracle):
    """The C++ host (flags, JSON, Parquet decode, pack driver) rebuilt with
    AddressSanitizer + UBSan and run through the no-GPU --dump_bitset path."""
    from cuking_amd import build as b
    inc, libdir, libs = b.arrow_flags()
    exe = tmp_path / "cuking_asan"
    cmd = ["g++", "-O1", "-g", "-std=c++20", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-pthread", f"-I{b.INCLUDE}", f"-I{b.HOST}",
           f"-isystem{inc}", *b.ROCM_HOST_FLAGS, *map(str, sorted(b.HOST.glob("*.cc"))),
           "-o", str(exe), f"-L{b.PKG}", "-l:libcuking_amd.so", f"-L{libdir}", *libs,
           *b.ROCM_HOST_LIBS,
           f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{b.PKG}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    rng = np.random.default_rng(9)
    geno = random_genotypes(rng, 23, 300, missing=0.2)
    write_input_tables(tmp_path / "in", geno, num_files=3, nullable=True, spark_layout=True)
    dump = tmp_path / "bits.bin"
    env = dict(**__import__("os").environ, ASAN_OPTIONS="detect_leaks=0")
    p = subprocess.run([str(exe), "--input_uri", str(tmp_path / "in"), "--output_uri",
                        str(tmp_path / "o"), "--dump_bitset", str(dump),
                        "--num_reader_threads=3"], capture_output=True, text=True, env=env,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    got = np.fromfile(dump, dtype=np.uint64).reshape(23, -1)
    assert np.array_equal(got, oracle.bitset_from_genotypes(geno))
    p = subprocess.run([str(exe), "--input_uri", str(tmp_path / "none"), "--output_uri", "x",
                        "--dump_bitset", str(dump)], capture_output=True, text=True, env=env)
    assert p.returncode == 1 and "ERROR: AddressSanitizer" not in p.stderr


# ------------------------------------------- multi-GPU scheduling (CPU) ----
@pytest.mark.parametrize("n,k,shard,world,chunks", [
    (1000, 1, 0, 3, 4), (10_000, 1, 0, 8, 8), (734_000, 1, 0, 8, 8), (300, 1, 0, 8, 3),
    (129, 1, 0, 2, 8), (5000, 2, 0, 4, 5), (5000, 2, 1, 4, 5), (5000, 3, 5, 1, 2)])
def test_multi_gpu_schedule_matches_python_driver(tmp_path, n, k, shard, world, chunks):
    """`cuking --print_schedule` (host/schedule.h, what --num_gpus runs) ==
    cuking_amd.dist's schedule, whose coverage of every pair exactly once is
    brute-forced in test_dist_cpu.py; tile ranges partition the enumeration."""
    from cuking_amd.dist import chunk_ranges, staged_schedule, tile_partition
    d = tmp_path / "in"
    d.mkdir()
    (d / "metadata.json").write_text(json.dumps(
        {"num_sites": 64, "samples": [f"s{i}" for i in range(n)]}))
    p = run_cli("--input_uri", d, "--output_uri", tmp_path / "o", "--print_schedule",
                f"--num_gpus={world}", f"--bcast_chunks={chunks}", f"--split_factor={k}",
                f"--shard_index={shard}", check=True)
    got = json.loads(p.stdout.strip().splitlines()[-1])
    sm = cuking_amd.Submatrix(n, k, shard)
    stored, tile = sm.NumSamples(), got["tile"]
    assert got["world"] == world and got["stored_samples"] == stored
    lib = cuking_amd._lib.load()
    import ctypes as C
    assert tile == lib.cuking_tile_samples(None)
    assert got["num_tiles"] == lib.cuking_num_tiles(None, C.byref(sm.c))
    assert [tuple(r) for r in got["tile_ranges"]] == tile_partition(got["num_tiles"], world)
    assert got["tile_ranges"][0][0] == 0 and got["tile_ranges"][-1][1] == got["num_tiles"]
    assert [tuple(c) for c in got["chunks"]] == chunk_ranges(stored, tile, chunks)
    assert got["mode"] == ("staged" if sm.i_begin == sm.j_begin else "simple")
    for r in range(world):
        want = staged_schedule(stored, tile, world, r, chunks)
        assert len(got["staged"][r]) == len(want)
        for step, ((c0, c1), rect) in zip(got["staged"][r], want):
            assert tuple(step["chunk"]) == (c0, c1)
            if rect is None:
                assert step["rows"] is None
            else:
                assert tuple(step["rows"]) == rect[0] and rect[1] == (c0, c1)


def test_variant_flag_sets_the_geometry_of_the_schedules(tmp_path):
    """`--variant=N` (the four-product kernel for callsets the filter's bound cannot thin
    out) reaches the library's default before anything is cut: the
    schedule of a --num_gpus run is in that variant's tiles; out of range is an error."""
    d = tmp_path / "in"
    d.mkdir()
    (d / "metadata.json").write_text(json.dumps(
        {"num_sites": 64, "samples": [f"s{i}" for i in range(1000)]}))
    lib = cuking_amd._lib.load()
    tiles = {}
    for args in ((), ("--variant=7",), ("--variant=6",), ("--variant", "0")):
        p = run_cli("--input_uri", d, "--output_uri", tmp_path / "o", "--print_schedule",
                    "--num_gpus=2", *args, check=True)
        got = json.loads(p.stdout.strip().splitlines()[-1])
        tiles[" ".join(args)] = (got["tile"], got["num_tiles"])
    assert tiles[""] == tiles["--variant=7"]            # the default is the filter variant
    assert tiles["--variant=7"][0] == 256 and tiles["--variant=6"][0] == 128
    assert tiles["--variant=6"][1] > tiles["--variant=7"][1]
    assert tiles["--variant 0"][0] in (64, 128)
    n_variants = lib.cuking_num_variants()
    for bad in (str(n_variants), "-1", "x"):
        p = run_cli("--input_uri", d, "--output_uri", tmp_path / "o", "--print_schedule",
                    f"--variant={bad}")
        assert p.returncode == 1 and "flag 'variant'" in p.stderr, bad
    assert "--variant=N" in run_cli("--help", check=True).stdout


def test_weighted_tile_ranges_and_calibration_plan(tmp_path):
    """`--rank_weights` / calibration of the simple schedule (host/schedule.h)
    against cuking_amd.dist.weighted_tile_partition, and the new flags' errors."""
    from cuking_amd.dist import weighted_tile_partition
    d = tmp_path / "in"
    d.mkdir()
    n = 300_000
    (d / "metadata.json").write_text(json.dumps(
        {"num_sites": 64, "samples": [f"s{i}" for i in range(n)]}))

    def schedule(*extra):
        p = run_cli("--input_uri", d, "--output_uri", tmp_path / "o", "--print_schedule",
                    "--num_gpus=8", "--multi_gpu_mode=simple", *extra, check=True)
        return json.loads(p.stdout.strip().splitlines()[-1])

    weights = [1.0, 1.03, 0.97, 1.01, 1.0, 0.95, 1.06, 1.0]
    got = schedule("--rank_weights=" + ",".join(map(str, weights)))
    assert got["weighted"] is True and got["calibration_tiles"] == 0
    want = weighted_tile_partition(got["num_tiles"], weights)
    assert [tuple(r) for r in got["tile_ranges"]] == want
    sizes = [b - a for a, b in want]
    assert sizes[6] > sizes[0] > sizes[5] and sum(sizes) == got["num_tiles"]
    # no weights: equal ranges as the fall-back, and a calibration launch of about
    # 2 % of a rank's share (whole rounds of 256 workgroups)
    got = schedule()
    share = got["num_tiles"] // 8
    assert got["weighted"] is False
    assert got["calibration_tiles"] % 256 == 0
    # (... at least 8 rounds: the floor at 256-sample tiles for this cohort)
    assert min(0.015 * share, 2048) <= got["calibration_tiles"] <= max(0.021 * share, 2048)
    assert schedule("--calibrate=false")["calibration_tiles"] == 0
    assert schedule("--calibration_tiles=512")["calibration_tiles"] == 512
    # small jobs do not calibrate
    (d / "metadata.json").write_text(json.dumps(
        {"num_sites": 64, "samples": [f"s{i}" for i in range(5000)]}))
    assert schedule()["calibration_tiles"] == 0
    for bad, needle in (("--rank_weights=1,0", "rank_weights"), ("--rank_weights=a", "rank_weights"),
                        ("--collectives=mpi", "collectives"), ("--inject_failure=1", "inject_failure"),
                        ("--inject_failure=1:later", "inject_failure"), ("--pack=gpu", "pack"),
                        ("--calibrate=maybe", "calibrate")):
        p = run_cli("--input_uri", d, "--output_uri", tmp_path / "o", bad)
        assert p.returncode == 1 and needle in p.stderr, (bad, p.stderr)


def test_weighted_row_deal_of_the_staged_schedule(tmp_path):
    """`--rank_weights` in the staged schedule: tile rows dealt in proportion to
    the weights, every row to exactly one rank, each rank's rows spread evenly
    over the deal's period (host/schedule.h MakeRowDeal)."""
    d = tmp_path / "in"
    d.mkdir()
    n, world = 40_000, 4
    (d / "metadata.json").write_text(json.dumps(
        {"num_sites": 64, "samples": [f"s{i}" for i in range(n)]}))
    weights = [1.0, 1.10, 0.95, 1.02]
    p = run_cli("--input_uri", d, "--output_uri", tmp_path / "o", "--print_schedule",
                f"--num_gpus={world}", "--multi_gpu_mode=staged", "--bcast_chunks=5",
                "--rank_weights=" + ",".join(map(str, weights)), check=True)
    got = json.loads(p.stdout.strip().splitlines()[-1])
    tile, period = got["tile"], got["row_deal_period"]
    assert got["mode"] == "staged" and period == 16 * world
    deal = got["row_deal"]
    assert sorted(x for r in deal for x in r) == list(range(period))      # a partition
    share = [len(r) / period for r in deal]
    for s_, w in zip(share, weights):
        assert abs(s_ - w / sum(weights)) <= 1.0 / period
    assert share[1] > share[0] > share[2]
    for r in deal:                                                       # evenly spread
        gaps = np.diff(sorted(r) + [sorted(r)[0] + period])
        assert gaps.max() < 2 * -(-period // len(r)), r           # no long stretch without the rank
    # every (tile row, chunk) rectangle row belongs to exactly one rank
    tile_rows = (n + tile - 1) // tile
    for k, (c0, c1) in enumerate(got["chunks"]):
        owners = np.zeros(tile_rows, dtype=np.int32)
        for r in range(world):
            step = got["staged"][r][k]
            assert tuple(step["chunk"]) == (c0, c1)
            for b, e, st in step["rects"]:
                assert e == c1 and st == period * tile and b % tile == 0
                owners[np.arange(b // tile, (e + tile - 1) // tile, st // tile)] += 1
        below = (c1 + tile - 1) // tile
        assert np.all(owners[:below] == 1) and np.all(owners[below:] == 0)


def test_host_threads_under_tsan(tmp_path, oracle):
    """ThreadSanitizer over the threaded host code: the reader pool (ParallelFor,
    one task per row group) decoding on 8 threads while cuking_pack_host's relaxed
    atomic ANDs (cuking.cu:317-323, :550-553) land in ONE shared bitset.  The
    host-only half of the ABI (csrc/king_host.cc, plain C++) is compiled INTO the
    instrumented binary, in front of the copy inside libcuking_amd.so, so the
    pack itself is instrumented.  CPU only (--dump_bitset touches no GPU)."""
    from cuking_amd import build as b
    inc, libdir, libs = b.arrow_flags()
    exe = tmp_path / "cuking_tsan"
    cmd = ["g++", "-O1", "-g", "-std=c++20", "-fsanitize=thread", "-pthread", f"-I{b.INCLUDE}",
           f"-I{b.HOST}", f"-I{b.CSRC}", f"-isystem{inc}", *b.ROCM_HOST_FLAGS,
           *map(str, sorted(b.HOST.glob("*.cc"))),
           *[str(b.CSRC / f) for f in b.HOST_ABI_SOURCES],
           "-o", str(exe), f"-L{b.PKG}", "-l:libcuking_amd.so", f"-L{libdir}", *libs,
           *b.ROCM_HOST_LIBS,
           f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{b.PKG}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    rng = np.random.default_rng(12)
    geno = random_genotypes(rng, 61, 900, missing=0.1)
    write_input_tables(tmp_path / "in", geno, num_files=2, row_group_size=4000,
                       spark_layout=False)
    dump = tmp_path / "bits.bin"
    supp = Path(__file__).parent / "tsan.supp"      # prebuilt libarrow / libparquet only
    env = dict(**__import__("os").environ,
               TSAN_OPTIONS=f"halt_on_error=1 exitcode=66 suppressions={supp}")
    p = subprocess.run([str(exe), "--input_uri", str(tmp_path / "in"), "--output_uri",
                        str(tmp_path / "o"), "--dump_bitset", str(dump),
                        "--num_reader_threads=8"], capture_output=True, text=True, env=env,
                       timeout=600)
    assert p.returncode == 0 and "ThreadSanitizer" not in p.stderr, p.stderr[-4000:]
    tasks = int(p.stdout.split(" decode tasks")[0].split()[-1])
    assert tasks > 8                                   # row groups, not files, were the unit
    got = np.fromfile(dump, dtype=np.uint64).reshape(61, -1)
    assert np.array_equal(got, oracle.bitset_from_genotypes(geno))


# ------------------------------------------------- decode + pack (CPU) ----
def dump_bits(in_dir, tmp_path, n_stored, wps, *extra):
    """The packed bitset of the host path -- through both forms of the decode (whole
    tables; batches packed as they are decoded: the default, here also with batches of
    97 triples so that they end inside pages, row groups and runs of nulls), which
    must agree bit for bit."""
    dump = tmp_path / "bits.bin"
    got = {}
    for mode in (("--decode=table",), ("--decode=stream",), ("--decode=stream", "--decode_batch=97"),
                 ()):
        run_cli("--input_uri", in_dir, "--output_uri", tmp_path / "unused",
                "--dump_bitset", dump, "--num_reader_threads=4", *mode, *extra, check=True)
        got[mode] = np.fromfile(dump, dtype=np.uint64).reshape(n_stored, wps)
    first = got[("--decode=table",)]
    for mode, bits in got.items():
        assert np.array_equal(bits, first), mode
    return first


@pytest.mark.parametrize("layout", [
    dict(compression="zstd", nullable=True, spark_layout=True),      # Spark / Hail
    dict(compression="snappy", nullable=False, spark_layout=False),
    dict(compression=None, nullable=True, spark_layout=False, use_dictionary=False),
    dict(compression="zstd", nullable=False, spark_layout=True, row_group_size=1000,
         shuffle_seed=3),
    dict(compression="gzip", nullable=True, spark_layout=False, row_group_size=777),
])
def test_parquet_decode_and_pack_match_oracle(tmp_path, oracle, layout):
    rng = np.random.default_rng(17)
    n, m = 37, 523
    geno = random_genotypes(rng, n, m, missing=0.15)
    ids = [f"sample \"{k}\" é中" if k % 5 == 0 else f"NA{k:05d}" for k in range(n)]
    in_dir = tmp_path / "in"
    write_input_tables(in_dir, geno, ids, num_files=5, **layout)
    wps = cuking_amd.words_per_sample(m)
    got = dump_bits(in_dir, tmp_path, n, wps)
    assert np.array_equal(got, oracle.bitset_from_genotypes(geno))
    # a shard only keeps its own samples, in Submatrix storage order
    for k, shard in ((2, 1), (3, 4)):
        osm = oracle.submatrix(n, k, shard)
        got = dump_bits(in_dir, tmp_path, osm.i_end - osm.i_begin +
                        (0 if osm.i_begin == osm.j_begin else osm.j_end - osm.j_begin),
                        wps, f"--split_factor={k}", f"--shard_index={shard}")
        assert np.array_equal(got, oracle.bitset_from_genotypes(geno, osm))


def test_row_groups_are_the_unit_of_the_decode(tmp_path, oracle):
    """Fewer files than reader threads: one decode task per (file, row group)
    (the reference hands out whole files, cuking.cu:550-553); same bitset."""
    import pyarrow.parquet as pq
    rng = np.random.default_rng(23)
    n, m = 41, 1200
    geno = random_genotypes(rng, n, m, missing=0.1)
    in_dir = tmp_path / "in"
    paths = write_input_tables(in_dir, geno, num_files=2, row_group_size=5000, nullable=True,
                               spark_layout=True, shuffle_seed=5)
    groups = sum(pq.ParquetFile(p).metadata.num_row_groups for p in paths)
    assert groups > 6
    dump = tmp_path / "bits.bin"
    p = run_cli("--input_uri", in_dir, "--output_uri", tmp_path / "unused", "--dump_bitset", dump,
                "--num_reader_threads=6", check=True)
    assert f"{groups} decode tasks" in p.stdout
    got = np.fromfile(dump, dtype=np.uint64).reshape(n, -1)
    assert np.array_equal(got, oracle.bitset_from_genotypes(geno))
    # enough files for the threads: whole files, as before
    p = run_cli("--input_uri", in_dir, "--output_uri", tmp_path / "unused", "--dump_bitset", dump,
                "--num_reader_threads=2", check=True)
    assert "2 decode tasks" in p.stdout
    assert np.array_equal(np.fromfile(dump, dtype=np.uint64).reshape(n, -1), got)


def test_null_genotypes_are_missing_and_bad_inputs_fail(tmp_path, oracle):
    import pyarrow as pa
    import pyarrow.parquet as pq
    d = tmp_path / "in"
    d.mkdir()
    (d / "metadata.json").write_text(json.dumps({"num_sites": 40, "samples": ["a", "b"]}))
    tbl = pa.table({"row_idx": pa.array([0, 1, 2, 3], pa.int64()),
                    "col_idx": pa.array([0, 0, 1, 1], pa.int64()),
                    "n_alt_alleles": pa.array([1, None, 2, 0], pa.int32())})
    pq.write_table(tbl, d / "t.parquet")
    got = dump_bits(d, tmp_path, 2, 2)
    geno = np.full((2, 40), -1, dtype=np.int8)
    geno[0, 0], geno[1, 2], geno[1, 3] = 1, 2, 0
    assert np.array_equal(got, oracle.bitset_from_genotypes(geno))

    def expect_fail(table, needle):
        for f in d.glob("*.parquet"):
            f.unlink()
        pq.write_table(table, d / "t.parquet")
        for mode in ("table", "stream"):
            p = run_cli("--input_uri", d, "--output_uri", tmp_path / "o", "--dump_bitset",
                        tmp_path / "x.bin", f"--decode={mode}")
            assert p.returncode == 1 and needle in p.stderr, (mode, p.stderr)

    expect_fail(pa.table({"row_idx": pa.array([0], pa.int64()),
                          "col_idx": pa.array([0], pa.int64()),
                          "n_alt_alleles": pa.array([3], pa.int32())}),
                "Invalid value for n_alt_alleles (3)")          # cuking.cu:698-702
    expect_fail(pa.table({"row_idx": pa.array([0], pa.int64()),
                          "col_idx": pa.array([0], pa.int64())}),
                "Expected 3 columns, found 2")                  # :586-590
    expect_fail(pa.table({"row_idx": pa.array([0], pa.int32()),
                          "col_idx": pa.array([0], pa.int64()),
                          "n_alt_alleles": pa.array([1], pa.int32())}),
                "Expected INT64 type, found INT32")             # :608-613
    expect_fail(pa.table({"row_idx": pa.array([64], pa.int64()),
                          "col_idx": pa.array([0], pa.int64()),
                          "n_alt_alleles": pa.array([1], pa.int32())}),
                "outside the 64 padded sites")
    expect_fail(pa.table({"row_idx": pa.array([None], pa.int64()),
                          "col_idx": pa.array([0], pa.int64()),
                          "n_alt_alleles": pa.array([1], pa.int32())}),
                "null values are not allowed")


def test_without_gpu_the_cli_fails_loudly(tmp_path):
    if cuking_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    geno = random_genotypes(np.random.default_rng(0), 4, 40)
    write_input_tables(tmp_path / "in", geno, num_files=1)
    p = run_cli("--input_uri", tmp_path / "in", "--output_uri", tmp_path / "out")
    assert p.returncode == 1 and "no CPU path" in p.stderr
    assert not (tmp_path / "out").exists()


# ----------------------------------------------------- end to end (GPU) ----
def c0_genotypes():
    rng = np.random.default_rng(20240229)
    n, m = 1000, 10000
    geno = random_genotypes(rng, n, m, missing=0.01)
    geno[900] = geno[10]                          # duplicate
    for child, (a, b) in {901: (20, 21), 902: (20, 21), 903: (22, 23)}.items():
        for s in range(m):                         # crude Mendelian children
            ga, gb = geno[a, s], geno[b, s]
            if ga < 0 or gb < 0:
                geno[child, s] = -1
                continue
            ta = ga // 2 if ga != 1 else rng.integers(0, 2)
            tb = gb // 2 if gb != 1 else rng.integers(0, 2)
            geno[child, s] = ta + tb
    return geno


@pytest.fixture(scope="module")
def c0(tmp_path_factory):
    d = tmp_path_factory.mktemp("c0")
    geno = c0_genotypes()
    ids = [f"HG{k:05d}" for k in range(geno.shape[0])]
    write_input_tables(d / "in", geno, ids, num_files=8, compression="zstd",
                       nullable=True, spark_layout=True)
    return dict(dir=d, geno=geno, ids=ids)


def expected_table(oracle, geno, ids, thr, k=1, shard=0):
    osm = oracle.submatrix(geno.shape[0], k, shard)
    res, ovf, _ = oracle.compute(osm, oracle.bitset_from_genotypes(geno, osm), thr)
    assert ovf == 0
    return res


def check_output(path, exp, ids):
    import pyarrow.parquet as pq
    meta = pq.ParquetFile(path).metadata
    assert meta.num_row_groups == 1                                # cuking.cu:804
    schema = pq.ParquetFile(path).schema
    names = [schema.column(k).name for k in range(6)]
    assert names == ["i", "j", "kin", "ibs0", "ibs1", "ibs2"]      # :770-788
    phys = [schema.column(k).physical_type for k in range(6)]
    assert phys == ["BYTE_ARRAY", "BYTE_ARRAY", "FLOAT", "INT32", "INT32", "INT32"]
    assert all(schema.column(k).max_definition_level == 0 for k in range(6))  # REQUIRED
    assert str(schema.column(0).logical_type) == "String"
    assert all(meta.row_group(0).column(k).compression == "SNAPPY" for k in range(6))
    t = pq.read_table(path)
    assert t.num_rows == len(exp)
    assert t.column("i").to_pylist() == [ids[x] for x in exp["sample_i"]]
    assert t.column("j").to_pylist() == [ids[x] for x in exp["sample_j"]]
    kin = t.column("kin").to_numpy()
    assert kin.dtype == np.float32
    assert np.array_equal(kin.view(np.uint32), exp["kin"].view(np.uint32))   # bit-exact
    for name in ("ibs0", "ibs1", "ibs2"):
        assert np.array_equal(t.column(name).to_numpy().astype(np.uint32), exp[name])


def test_c0_cpu_plumbing_and_oracle_vs_naive(c0, oracle, naive, tmp_path):
    """BASELINE.json configs[0] without a GPU: the binary's Parquet decode + pack
    of the 1k x 10k cohort (8 zstd files, OPTIONAL columns, Spark layout) gives
    the oracle's bitset, and on it the bit-plane oracle equals the naive
    per-genotype oracle for all 499,500 pairs (counts exact, kin bit-exact)."""
    geno = c0["geno"]
    n, m = geno.shape
    got = dump_bits(c0["dir"] / "in", tmp_path, n, cuking_amd.words_per_sample(m))
    bits = oracle.bitset_from_genotypes(geno)
    assert np.array_equal(got, bits)
    oi, oj, counts, kin = oracle.all_pairs(oracle.submatrix(n), got)
    ni, nj, nc = naive.all_pairs_matmul(geno)
    assert len(oi) == n * (n - 1) // 2
    assert np.array_equal(oi, ni) and np.array_equal(oj, nj)
    for k, name in enumerate(counts.dtype.names):
        assert np.array_equal(counts[name].astype(np.int64), nc[:, k]), name
    nk = naive.kin_f32(nc[:, 0], nc[:, 1], nc[:, 2], nc[:, 3])
    assert np.array_equal(kin.view(np.uint32), nk.view(np.uint32))
    res, ovf, _ = oracle.compute(oracle.submatrix(n), got, 0.05, threads=4)
    assert ovf == 0 and res.tobytes() == naive.king(geno, 0.05).tobytes()
    related = {(int(r["sample_i"]), int(r["sample_j"])) for r in res}
    assert {(10, 900), (20, 901), (21, 901), (901, 902), (22, 903)} <= related


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--kernel=stream"], ["--pack=device"]])
def test_c0_end_to_end(c0, oracle, extra):
    """1k x 10k through real Parquet (zstd, OPTIONAL, Spark layout incl. the
    _temporary junk) -> bit-exact records vs the CPU path."""
    out = c0["dir"] / ("out_" + "_".join(a.strip("-").replace("=", "_") for a in extra))
    p = run_cli("--input-uri", c0["dir"] / "in", "--output-uri", out,
                "--kin-threshold=0.05", "--num_reader_threads=8", *extra, check=True)
    assert "Found 8 input files." in p.stdout
    exp = expected_table(oracle, c0["geno"], c0["ids"], 0.05)
    assert len(exp) >= 5
    files = sorted(f.name for f in out.iterdir())
    assert files == ["part-00000.snappy.parquet"]                   # :868-870
    check_output(out / files[0], exp, c0["ids"])
    summary = json.loads(p.stdout.strip().splitlines()[-1])
    assert summary["pairs"] == 1000 * 999 // 2 and summary["results"] == len(exp)


@pytest.mark.gpu
def test_device_pack_multichunk_tables(c0, oracle):
    """Tables larger than the packer's 2M-triple staging chunk (two files of
    ~5M rows), two reader threads with their own streams."""
    d = c0["dir"] / "in_two_files"
    write_input_tables(d, c0["geno"], c0["ids"], num_files=2, compression="snappy",
                       nullable=False, spark_layout=False)
    out = c0["dir"] / "out_two_files"
    run_cli("--input_uri", d, "--output_uri", out, "--kin_threshold=0.05", "--pack=device",
            "--num_reader_threads=2", check=True)
    exp = expected_table(oracle, c0["geno"], c0["ids"], 0.05)
    check_output(out / "part-00000.snappy.parquet", exp, c0["ids"])


@pytest.mark.gpu
def test_device_pack_large_bitset_tiny_files(tmp_path, oracle):
    """--pack=device with a 6 GB bitset and sixteen tiny part files: the all-ones
    fill of the bitset (milliseconds at this size, on the null stream) must be
    complete before the first pack kernel clears bits on a reader thread's own
    non-blocking stream -- otherwise the fill overwrites them and those
    genotypes silently turn missing."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    n, m, m_used, thr = 60_000, 400_000, 20_000, 0.05
    who = np.array(sorted({0, 1, 2, 3, 777, 778, 20_000, 20_001, 31_415, 31_416, 44_444,
                           44_445, 59_000, 59_001, 59_996, 59_997, 59_998, 59_999,
                           128, 129, 255, 256, 10_000, 10_001}))
    rng = np.random.default_rng(11)
    sub = random_genotypes(rng, len(who), m_used, missing=0.02)
    sub[1], sub[5], sub[-1] = sub[0], sub[4], sub[-2]          # duplicates
    d = tmp_path / "in"
    d.mkdir()
    ids = [f"S{k:06d}" for k in range(n)]
    (d / "metadata.json").write_text(json.dumps({"num_sites": m, "samples": ids}))
    bounds = np.linspace(0, m_used, 17).astype(int)
    for f in range(16):
        block = sub[:, bounds[f]:bounds[f + 1]].T
        row, col = np.nonzero(block >= 0)
        t = pa.table({"row_idx": (row + bounds[f]).astype(np.int64),
                      "col_idx": who[col].astype(np.int64),
                      "n_alt_alleles": block[row, col].astype(np.int32)})
        pq.write_table(t, d / f"part-{f:05d}.parquet", compression="zstd")
    out = tmp_path / "out"
    for attempt in range(2):
        p = run_cli("--input_uri", d, "--output_uri", out, f"--kin_threshold={thr}",
                    "--pack=device", "--num_reader_threads=16", check=True)
        full = np.full((len(who), m), -1, dtype=np.int8)
        full[:, :m_used] = sub
        exp, ovf, _ = oracle.compute(oracle.submatrix(len(who)),
                                     oracle.bitset_from_genotypes(full), thr)
        assert ovf == 0 and len(exp) >= 3
        exp = exp.copy()
        exp["sample_i"], exp["sample_j"] = who[exp["sample_i"]], who[exp["sample_j"]]
        check_output(out / "part-00000.snappy.parquet", exp, ids)


@pytest.mark.gpu
def test_c0_split_factor_shards(c0, oracle):
    """README.md:94-102: k = 3 => 6 shards, each its own part file."""
    out = c0["dir"] / "out_split"
    thr = 0.0884
    parts = []
    for shard in range(6):
        run_cli("--input_uri", c0["dir"] / "in", "--output_uri", f"file://{out}",
                "--split_factor=3", f"--shard_index={shard}", check=True)
        exp = expected_table(oracle, c0["geno"], c0["ids"], thr, 3, shard)
        check_output(out / f"part-{shard:05d}.snappy.parquet", exp, c0["ids"])
        parts.append(exp)
    whole = expected_table(oracle, c0["geno"], c0["ids"], thr)
    assert sum(len(p) for p in parts) == len(whole)
    from cuking_amd.merge import merge
    assert merge(out, 3, merged=out / "all.parquet").num_rows == len(whole)
    assert (out / "_SUCCESS").exists()
    (out / "all.parquet").unlink()
    with pytest.raises(FileNotFoundError):
        merge(out, 4)
    t = read_results(out)
    assert t.num_rows == len(whole)
    got = sorted(zip(t.column("i").to_pylist(), t.column("j").to_pylist()))
    assert got == sorted((c0["ids"][a], c0["ids"][b])
                         for a, b in zip(whole["sample_i"], whole["sample_j"]))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--num_gpus=1"], ["--num_gpus=1", "--multi_gpu_mode=simple"],
                                   ["--num_gpus=1", "--pack=device", "--bcast_chunks=3"],
                                   ["--num-gpus", "1", "--kernel=stream"]])
def test_c0_through_the_rccl_path(c0, oracle, extra):
    """`--num_gpus=1`: the multi-GPU host path with a one-rank RCCL communicator
    -- per-rank thread, ncclCommInitAll, chunked ncclBroadcast, staged rectangle
    launches (or a tile range), ncclAllGather of the counts, gather -- must write
    the same file as the classic path."""
    out = c0["dir"] / ("out_mg_" + "_".join(a.strip("-").replace("=", "_") for a in extra))
    p = run_cli("--input-uri", c0["dir"] / "in", "--output-uri", out,
                "--kin-threshold=0.05", "--num_reader_threads=8", *extra, check=True)
    exp = expected_table(oracle, c0["geno"], c0["ids"], 0.05)
    check_output(out / "part-00000.snappy.parquet", exp, c0["ids"])
    summary = json.loads(p.stdout.strip().splitlines()[-1])
    assert summary["gpus"] == 1 and summary["results"] == len(exp)
    assert summary["rank_results"] == [len(exp)]
    staged = "--multi_gpu_mode=simple" not in extra and "--kernel=stream" not in extra
    assert summary["multi_gpu_mode"] == ("staged" if staged else "simple")
    assert summary["bytes_broadcast"] == 1000 * cuking_amd.words_per_sample(10000) * 8


@pytest.mark.gpu
def test_rccl_path_shards_overflow_and_bad_gpu_count(c0, oracle):
    # an off-diagonal shard takes the simple schedule
    out = c0["dir"] / "out_mg_split"
    p = run_cli("--input_uri", c0["dir"] / "in", "--output_uri", out, "--split_factor=2",
                "--shard_index=1", "--num_gpus=1", "--kin_threshold=0.05", check=True)
    exp = expected_table(oracle, c0["geno"], c0["ids"], 0.05, 2, 1)
    check_output(out / "part-00001.snappy.parquet", exp, c0["ids"])
    assert json.loads(p.stdout.strip().splitlines()[-1])["multi_gpu_mode"] == "simple"
    # overflow is the reference's error, whatever the number of GPUs
    p = run_cli("--input_uri", c0["dir"] / "in", "--output_uri", c0["dir"] / "out_mg_ovf",
                "--kin_threshold=-10", "--max_results=1000", "--num_gpus=1")
    assert p.returncode == 1
    assert ("Error: RESOURCE_EXHAUSTED: Could not store all results: try increasing "
            "the --max_results parameter.") in p.stderr
    # more GPUs than the box has
    p = run_cli("--input_uri", c0["dir"] / "in", "--output_uri", c0["dir"] / "out_mg_bad",
                "--num_gpus=64")
    assert p.returncode == 1 and "INVALID_ARGUMENT" in p.stderr and "GPU(s) are visible" in p.stderr


# The N > 1 branch of the C++ multi-GPU host on ONE GPU: `--collectives=loopback`
# swaps RCCL for device-to-device copies + host rendezvous (host/collectives.h),
# so that three rank threads run RankMain's real schedule, gather offsets, cap on
# the total and agreement on failures (RCCL itself refuses two ranks per device).
@pytest.mark.gpu
@pytest.mark.parametrize("extra", [
    ["--multi_gpu_mode=staged", "--bcast_chunks=3"],
    ["--multi_gpu_mode=simple"],
    ["--multi_gpu_mode=simple", "--calibration_tiles=2"],
    ["--multi_gpu_mode=simple", "--rank_weights=1,3,2"],
    ["--multi_gpu_mode=staged", "--pack=device", "--bcast_chunks=5"],
    ["--multi_gpu_mode=staged", "--rank_weights=1,3,2", "--bcast_chunks=2"],
    ["--kernel=stream"],
])
def test_three_ranks_on_one_gpu_through_the_loopback(c0, oracle, extra):
    out = c0["dir"] / ("out_lb_" + "_".join(a.strip("-").replace("=", "_").replace(",", "_")
                                            for a in extra))
    p = run_cli("--input-uri", c0["dir"] / "in", "--output-uri", out, "--kin-threshold=0.05",
                "--num_reader_threads=8", "--num_gpus=3", "--collectives=loopback", "--pack=host",
                *extra, check=True)
    exp = expected_table(oracle, c0["geno"], c0["ids"], 0.05)
    check_output(out / "part-00000.snappy.parquet", exp, c0["ids"])
    s = json.loads(p.stdout.strip().splitlines()[-1])
    assert s["gpus"] == 3 and s["collectives"] == "loopback" and s["results"] == len(exp)
    assert sum(s["rank_results"]) == len(exp) and len(s["rank_results"]) == 3
    # nothing allocated, nothing waited for, once the first broadcast was enqueued
    assert s["allocations_after_reserve"] == [0, 0, 0], s
    assert s["host_syncs_after_reserve"] == [0, 0, 0], s
    if "--kernel=stream" in extra:
        assert s["rank_results"][1:] == [0, 0]
        return
    if "--multi_gpu_mode=simple" in extra:
        ranges = s["rank_tile_ranges"]
        num_tiles = cuking_amd._lib.load().cuking_num_tiles(
            None, __import__("ctypes").byref(cuking_amd.Submatrix(1000).c))
        if "--calibration_tiles=2" in extra:
            assert s["calibration_tiles"] == 2 and len(s["rank_rates_tiles_per_ms"]) == 3
            assert all(r > 0 for r in s["rank_rates_tiles_per_ms"])
            assert ranges[0][0] == 6                      # behind the three calibration ranges
        else:
            assert s["calibration_tiles"] == 0 and ranges[0][0] == 0
        assert ranges[-1][1] == num_tiles
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        if "--rank_weights=1,3,2" in extra:
            sizes = [b - a for a, b in ranges]
            assert sizes[1] > sizes[2] > sizes[0]
    # every rank found something in this cohort's staged / ranged share, or at least ran
    assert len(s["rank_kernel_ms"]) == 3 and all(ms > 0 for ms in s["rank_kernel_ms"])


@pytest.mark.gpu
def test_loopback_ranks_overflow_failures_and_shards(c0, oracle):
    base = ["--input_uri", c0["dir"] / "in", "--num_gpus=3", "--collectives=loopback",
            "--num_reader_threads=8"]
    # The total over the ranks exceeds --max_results although every rank stays
    # under it: the reference's error (cuking.cu:747-751), whatever the GPU count.
    # (threshold 0: about half of the 499,500 pairs pass, spread over all ranks)
    total = len(expected_table(oracle, c0["geno"], c0["ids"], 0.0))
    assert total > 100_000
    ok = run_cli(*base, "--output_uri", c0["dir"] / "out_lb_cap2", "--kin_threshold=0.0",
                 f"--max_results={total}", "--multi_gpu_mode=simple", check=True)
    s2 = json.loads(ok.stdout.strip().splitlines()[-1])
    assert s2["results"] == total and sum(s2["rank_results"]) == total
    assert max(s2["rank_results"]) < total - 1 and min(s2["rank_results"]) > 0, s2["rank_results"]
    p = run_cli(*base, "--output_uri", c0["dir"] / "out_lb_cap3", "--kin_threshold=0.0",
                f"--max_results={total - 1}", "--multi_gpu_mode=simple")
    assert p.returncode == 1
    assert ("Error: RESOURCE_EXHAUSTED: Could not store all results: try increasing "
            "the --max_results parameter.") in p.stderr
    assert not (c0["dir"] / "out_lb_cap3" / "part-00000.snappy.parquet").exists()
    # A failure on ONE rank ends the job on all of them, in every phase and in
    # both schedules: exit 1 with that rank's message, no hang, no output file.
    for mode in ("staged", "simple"):
        for phase in ("setup", "compute", "gather"):
            out = c0["dir"] / f"out_lb_fail_{mode}_{phase}"
            p = run_cli(*base, "--output_uri", out, "--kin_threshold=0.05",
                        f"--multi_gpu_mode={mode}", f"--inject_failure=1:{phase}")
            assert p.returncode == 1, (mode, phase, p.stdout, p.stderr)
            assert f"rank 1: injected failure in phase {phase}" in p.stderr
            assert not (out / "part-00000.snappy.parquet").exists()
    # an off-diagonal shard over three ranks (simple schedule)
    out = c0["dir"] / "out_lb_split"
    p = run_cli(*base, "--output_uri", out, "--split_factor=2", "--shard_index=1",
                "--kin_threshold=0.05", check=True)
    check_output(out / "part-00001.snappy.parquet",
                 expected_table(oracle, c0["geno"], c0["ids"], 0.05, 2, 1), c0["ids"])
    assert json.loads(p.stdout.strip().splitlines()[-1])["multi_gpu_mode"] == "simple"


@pytest.mark.gpu
def test_hundred_million_triples_through_both_packs(tmp_path_factory):
    """A 1e8-genotype input (2000 samples x 50,000 sites, 8 zstd files of several
    row groups) through the host pack and the pipelined device pack: same output
    file, the decode handed out per row group (8 files, 16 reader threads), and
    the device pack not slower than the host pack beyond the noise of a shared
    box (archive/profiles/r02_pack_pipeline.txt: equal at this size, +37 % at 1e9)."""
    import sys
    from concurrent.futures import ProcessPoolExecutor
    tools = str(Path(__file__).resolve().parent.parent / "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)       # (the workers import the generator by module name)
    import cli_timing
    d = tmp_path_factory.mktemp("pack1e8")
    n, m, files = 2000, 50_000, 8
    (d / "in").mkdir()
    (d / "in" / "metadata.json").write_text(json.dumps(
        {"num_sites": m, "samples": [f"S{k:07d}" for k in range(n)]}))
    bounds = np.linspace(0, m, files + 1).astype(int)
    jobs = [(str(d / "in"), f, int(bounds[f]), int(bounds[f + 1]), n, 1, 2_000_000)
            for f in range(files)]
    with ProcessPoolExecutor(8) as ex:
        triples = sum(ex.map(cli_timing.write_part, jobs))
    assert 0.98e8 < triples < 1.0e8
    best, outputs = {}, {}
    for rep in range(2):
        for pack in ("host", "device"):
            # (both packs with the default decode: batches packed as they are decoded)
            p = run_cli("--input_uri", d / "in", "--output_uri", d / f"out_{pack}", f"--pack={pack}",
                        "--num_reader_threads=16", "--kin_threshold=0.05", check=True)
            s_ = json.loads(p.stdout.strip().splitlines()[-1])
            assert s_["triples"] == triples and s_["pack"] == pack and s_["decode"] == "stream"
            assert s_["decode_tasks"] > files            # row groups, not files
            best[pack] = min(best.get(pack, 1e9), s_["read_pack_seconds"])
            if rep == 1:
                print(pack, {k: s_[k] for k in ("read_pack_seconds", "decode_thread_seconds",
                                                "pack_thread_seconds", "device_pack_thread_seconds")})
            outputs[pack] = (d / f"out_{pack}" / "part-00000.snappy.parquet").read_bytes()
    assert outputs["host"] == outputs["device"] and len(outputs["host"]) > 0
    print(f"1e8 triples: host pack {best['host']:.3f} s, device pack {best['device']:.3f} s")
    # At this size the device pack's set-up (page-locked rings, streams: ~0.12 s) is
    # not amortised; it must stay within that -- plus the noise of a shared box -- of the
    # host pack (profiles/r04_pack_pipeline.txt: 0.175 s against 0.069 s here with 16 reader
    # threads; it overtakes the host pack near 1e9 triples).
    assert best["device"] <= best["host"] + 0.20, best
    # --pack=auto: the host pack for an input this small (0.2 GB of Parquet), at any
    # thread count
    for threads in (16, 48):
        p = run_cli("--input_uri", d / "in", "--output_uri", d / f"out_auto{threads}",
                    f"--num_reader_threads={threads}", "--kin_threshold=0.05", check=True)
        assert json.loads(p.stdout.strip().splitlines()[-1])["pack"] == "host"
        assert (d / f"out_auto{threads}" / "part-00000.snappy.parquet").read_bytes() == outputs["host"]


@pytest.mark.gpu
def test_cli_result_overflow(c0):
    p = run_cli("--input_uri", c0["dir"] / "in", "--output_uri", c0["dir"] / "out_ovf",
                "--kin_threshold=-10", "--max_results=1000")
    assert p.returncode == 1
    assert ("Error: RESOURCE_EXHAUSTED: Could not store all results: try increasing "
            "the --max_results parameter.") in p.stderr              # :747-751
    assert not (c0["dir"] / "out_ovf" / "part-00000.snappy.parquet").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("extra,k,shard", [([], 1, 0), (["--split_factor=2", "--shard_index=0"], 2, 0),
                                           (["--split-factor", "2", "--shard-index", "1"], 2, 1)])
def test_python_driver_matches_oracle(c0, oracle, extra, k, shard):
    """`python -m cuking_amd.run` (the multi-GPU driver, here with one rank):
    same files, same schema, same records as the C++ binary."""
    import subprocess
    import sys
    out = c0["dir"] / f"out_py_{k}_{shard}"
    p = subprocess.run([sys.executable, "-m", "cuking_amd.run", "--input-uri",
                        str(c0["dir"] / "in"), "--output_uri", str(out),
                        "--kin-threshold=0.05", "--num_reader_threads=8", *extra],
                       capture_output=True, text=True, timeout=600,
                       cwd=str(Path(__file__).resolve().parent.parent))
    assert p.returncode == 0, p.stderr
    exp = expected_table(oracle, c0["geno"], c0["ids"], 0.05, k, shard)
    check_output(out / f"part-{shard:05d}.snappy.parquet", exp, c0["ids"])


def test_python_driver_flag_errors(tmp_path):
    import subprocess
    import sys
    root = str(Path(__file__).resolve().parent.parent)
    for argv, msg in ((["--output-uri", "x"], "No input URI specified"),
                      (["--input-uri", "x", "--output-uri", "y", "--split-factor", "2",
                        "--shard-index", "3"], "Invalid shard index"),
                      (["--input-uri", "gs://b/x", "--output-uri", "y"], "Unsupported URI")):
        p = subprocess.run([sys.executable, "-m", "cuking_amd.run", *argv],
                           capture_output=True, text=True, timeout=300, cwd=root)
        assert p.returncode == 1 and f"Error: INVALID_ARGUMENT: {msg}" in p.stderr, p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("k,shard", [(1, 0), (2, 1)])
def test_python_driver_synthetic_mode(tmp_path, oracle, k, shard):
    """`--synthetic N,M,seed` (BASELINE configs without their Parquet form):
    device generator -> kernel -> Parquet, checked against the oracle twin."""
    import subprocess
    import sys
    import pyarrow.parquet as pq
    from cuking_amd.synth import plan_cohort
    n, m, seed, thr = 600, 3000, 5, 0.06
    out = tmp_path / "out"
    p = subprocess.run([sys.executable, "-m", "cuking_amd.run", "--output-uri", str(out),
                        "--synthetic", f"{n},{m},{seed}", f"--kin-threshold={thr}",
                        f"--split-factor={k}", f"--shard-index={shard}"],
                       capture_output=True, text=True, timeout=600,
                       cwd=str(Path(__file__).resolve().parent.parent))
    assert p.returncode == 0, p.stderr
    cohort = plan_cohort(n, seed)
    bits = oracle.synth_bitset(seed, cohort.kind, cohort.pa, cohort.pb, 0, n, m)
    osm = oracle.submatrix(n, k, shard)
    idx = list(range(osm.i_begin, osm.i_end))
    if osm.i_begin != osm.j_begin:
        idx += list(range(osm.j_begin, osm.j_end))
    exp, _, _ = oracle.compute(osm, np.ascontiguousarray(bits[idx]), thr)
    t = pq.read_table(out / f"part-{shard:05d}.snappy.parquet")
    assert t.num_rows == len(exp) > 0
    assert t.column("i").to_pylist() == [f"S{x:07d}" for x in exp["sample_i"]]
    assert t.column("j").to_pylist() == [f"S{x:07d}" for x in exp["sample_j"]]
    assert np.array_equal(t.column("kin").to_numpy().view(np.uint32), exp["kin"].view(np.uint32))
    for name in ("ibs0", "ibs1", "ibs2"):
        assert np.array_equal(t.column(name).to_numpy().astype(np.uint32), exp[name])


@pytest.mark.gpu
@pytest.mark.parametrize("extra,k,shard", [([], 1, 0), (["--num_gpus=1"], 1, 0),
                                           (["--split_factor=2", "--shard_index=1"], 2, 1),
                                           (["--split_factor=2", "--shard_index=2", "--num_gpus=1",
                                             "--multi_gpu_mode=simple"], 2, 2)])
def test_cli_synthetic_mode(tmp_path, oracle, extra, k, shard):
    """`cuking --synthetic=N,M,seed`: the C++ host plans the same cohort as
    cuking_amd.synth.plan_cohort (host/synth_plan.h), the device generator fills
    the shard's bitset, and the records are the oracle twin's -- through the
    classic path and through the RCCL path."""
    import pyarrow.parquet as pq
    from cuking_amd.synth import plan_cohort
    n, m, seed, thr = 700, 3000, 5, 0.06
    out = tmp_path / "out"
    p = run_cli("--output_uri", out, f"--synthetic={n},{m},{seed}", f"--kin_threshold={thr}",
                *extra, check=True)
    assert "Synthesising genotypes on the GPU" in p.stdout
    cohort = plan_cohort(n, seed)
    bits = oracle.synth_bitset(seed, cohort.kind, cohort.pa, cohort.pb, 0, n, m)
    osm = oracle.submatrix(n, k, shard)
    idx = list(range(osm.i_begin, osm.i_end))
    if osm.i_begin != osm.j_begin:
        idx += list(range(osm.j_begin, osm.j_end))
    exp, _, _ = oracle.compute(osm, np.ascontiguousarray(bits[idx]), thr)
    assert len(exp) > 0
    check_output(out / f"part-{shard:05d}.snappy.parquet", exp, [f"S{x:07d}" for x in range(n)])
    summary = json.loads(p.stdout.strip().splitlines()[-1])
    assert summary["pack"] == "synthetic" and summary["results"] == len(exp)
