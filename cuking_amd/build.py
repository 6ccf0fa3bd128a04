"""In-tree build recipes (hipcc / g++), no JIT cache: the built files travel to
the GPU box with the repository snapshot.

    python -m cuking_amd.build            # library + CLI
    python -m cuking_amd.build --lib      # libcuking_amd.so only
"""
from __future__ import annotations

import argparse
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
HOST = PKG / "host"
INCLUDE = ROOT / "include"

LIB_PATH = PKG / "libcuking_amd.so"
CLI_PATH = PKG / "bin" / "cuking"
# The non-default flags the library next to it was built with ("" = the shipped
# configuration).  Experiment scripts rebuild the library in place with extra
# -D flags; the next default build must not mistake that file for its own.
LIB_FLAGS_PATH = PKG / "libcuking_amd.flags"

HIP_SOURCES = ["king_abi.hip", "king_kernels.hip", "king_mfma.hip", "king_filter.hip",
               "king_sort.hip", "synth.hip"]
# Host-only half of the ABI: plain C++, also compiled by the sanitizer tests.
HOST_ABI_SOURCES = ["king_host.cc"]
# IEEE-correct fp32 divide (kinship must match the reference bit for bit):
# no fast-math, no contraction, correctly rounded divide/sqrt stays on.
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall",
    "-fno-fast-math", "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
]


# host/multi_gpu.cc: HIP runtime API (streams, events, copies) and RCCL from a
# plain g++ translation unit -- no device code outside csrc/.
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
ROCM_HOST_FLAGS = ["-D__HIP_PLATFORM_AMD__=1", f"-isystem{ROCM / 'include'}"]
ROCM_HOST_LIBS = [f"-L{ROCM / 'lib'}", "-lrccl", "-lamdhip64"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def _newer(target: Path, deps) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(Path(d).stat().st_mtime <= t for d in deps)


def build_library(force: bool = False, save_temps: bool = False,
                  tuning: bool = False) -> Path:
    """tuning=True adds timing-only experiment kernels (-DCUKING_TUNING); never
    the shipped configuration (build() and the tests use the default)."""
    srcs = [CSRC / s for s in HIP_SOURCES + HOST_ABI_SOURCES]
    deps = srcs + [CSRC / "king_common.h", CSRC / "king_device.h", CSRC / "king_host.h",
                   CSRC / "king_submatrix.h", INCLUDE / "cuking_amd.h",
                   Path(__file__)]
    extra = (["-DCUKING_TUNING"] if tuning else []) + \
        os.environ.get("CUKING_EXTRA_HIPFLAGS", "").split()
    wanted = " ".join(extra)
    # (no stamp = a library from before stamps existed, or a box the stamp did not
    #  travel to: trusted; a stamp that says something else = somebody's experiment)
    stamped = LIB_FLAGS_PATH.read_text().strip() if LIB_FLAGS_PATH.exists() else wanted
    if not force and _newer(LIB_PATH, deps) and stamped == wanted:
        return LIB_PATH
    cmd = [_hipcc(), *HIP_FLAGS, "-shared", f"-I{INCLUDE}", f"-I{CSRC}",
           *map(str, srcs), "-o", str(LIB_PATH)]
    # experiments: extra -D flags, e.g. CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_STAGES=3"
    for flag in extra:
        cmd.insert(1, flag)
    # Always with -save-temps (in build_tmp/): the assembly of the matrix-core
    # kernel is checked below.  --save-temps adds the resource-usage remarks on the
    # console; otherwise the compiler's output goes to build_tmp/build.log.
    cwd = PKG / "build_tmp"
    cwd.mkdir(exist_ok=True)
    cmd += ["-save-temps"]
    if save_temps:
        cmd += ["-Rpass-analysis=kernel-resource-usage"]
        subprocess.run(cmd, check=True, cwd=str(cwd))
    else:
        with open(cwd / "build.log", "w") as log:
            r = subprocess.run(cmd, cwd=str(cwd), stdout=log, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            sys.stderr.write((cwd / "build.log").read_text()[-8000:])
            raise subprocess.CalledProcessError(r.returncode, cmd)
    problems = check_mfma_loops(cwd / "king_mfma-hip-amdgcn-amd-amdhsa-gfx950.s")
    problems += check_filter_loop(cwd / "king_filter-hip-amdgcn-amd-amdhsa-gfx950.s")
    if problems and not (tuning or os.environ.get("CUKING_EXTRA_HIPFLAGS")):
        LIB_PATH.unlink(missing_ok=True)
        raise RuntimeError("matrix-core kernel: the compiler put vector-memory waits or scratch "
                           "accesses inside an LDS-DMA loop:\n  " + "\n  ".join(problems))
    for line in problems:
        print("warning (experiment build):", line, file=sys.stderr)
    LIB_FLAGS_PATH.write_text(wanted + "\n")
    return LIB_PATH


def check_filter_loop(asm_path: Path, verbose: bool = False):
    """The same check for king_filter_kernel (king_filter.hip): its k-loops hold LDS-DMA
    requests and hand-counted waits (22 in flight at a hand-over; 30 with
    -DCUKING_FILTER_FINE=1) and must hold no scratch access and no other vmcnt wait."""
    import re
    text = Path(asm_path).read_text()
    bodies = [m.group(1) for m in re.finditer(
        r"\n_ZN6cuking12_GLOBAL__N_1\d+king_filter(?:_persistent)?_kernelE\w+:(.*?)\.Lfunc_end", text, re.S)]
    if len(bodies) < 2:
        return [f"king_filter_kernel / king_filter_persistent_kernel not both in {asm_path}"]
    problems, seen = [], 0
    for block in (b for body in bodies for b in re.split(r"\n(?=\.LBB\d+_\d+:)", body)):
        lines = block.split("\n")
        head = lines[0].split(":")[0]
        end = next((i for i, l in enumerate(lines)
                    if re.search(r"s_cbranch_\w+ " + re.escape(head) + r"\b", l)), None)
        if end is None:
            continue
        loop = lines[:end + 1]
        mfma = sum("v_mfma" in l for l in loop)
        if mfma < 16 or not any("global_load_lds" in l for l in loop):
            continue
        seen += 1
        scratch = [l.strip() for l in loop if "scratch_" in l]
        waits = [l.strip() for l in loop if re.search(r"s_waitcnt.*vmcnt\(\d+\)", l)]
        foreign = [w for w in waits if not any(f"vmcnt({n})" in w for n in (22, 30))]
        if verbose:
            print(f"king_filter_kernel: loop {head} ({mfma} MFMAs): scratch {len(scratch)}, "
                  f"vmcnt waits {waits}")
        if scratch or foreign:
            problems.append(f"king_filter_kernel, loop {head}: scratch {scratch[:2]}, waits {foreign}")
    if seen < 2:
        problems.append("king_filter kernels: an LDS-DMA loop is missing (listing format changed?)")
    return problems


def check_mfma_loops(asm_path: Path, verbose: bool = False):
    """The LDS-DMA requests of king_mfma.hip are inline asm the compiler's wait-count
    pass does not see, so ANY wait it inserts on the vector-memory counter inside
    such a loop (for a spill reload or a load of its own still in flight at loop
    entry) drains the whole prefetch pipeline every k-step -- 10 % of a launch the
    one time it happened (round 2, full form).  Returns the offending loops: blocks
    with LDS-DMA and >= 16 MFMAs that hold a scratch instruction or a vmcnt wait
    other than the hand-counted ones."""
    import re
    text = Path(asm_path).read_text()
    funcs = re.split(r"\n(?=_ZN6cuking12_GLOBAL__N_116king_mfma_kernel\w+:)", text)[1:]
    if not funcs:
        return [f"no king_mfma_kernel in {asm_path}"]
    problems = []
    for f in funcs:
        m = re.match(r"_ZN6cuking12_GLOBAL__N_116king_mfma_kernelILb(\d)ELb(\d)ELi(\d)ELb(\d)", f)
        label = (f"king_mfma_kernel<FULL={m.group(1)}, SPLIT={m.group(2)}, ABLATE={m.group(3)}, "
                 f"N4={m.group(4)}>")
        body = f.split(".Lfunc_end")[0]
        seen = 0
        for block in re.split(r"\n(?=\.LBB\d+_\d+:)", body):
            lines = block.split("\n")
            head = lines[0].split(":")[0]
            # the loop proper ends at its back branch; what follows in the listing
            # is the fall-through block
            end = next((i for i, l in enumerate(lines)
                        if re.search(r"s_cbranch_\w+ " + re.escape(head) + r"\b", l)), None)
            if end is None:
                continue
            loop = lines[:end + 1]
            mfma = sum("v_mfma" in l for l in loop)
            if mfma < 16 or not any("global_load_lds" in l for l in loop):
                continue  # (the five-product full form's pass in front: compiler-counted loads, no LDS-DMA)
            seen += 1
            scratch = [l.strip() for l in loop if "scratch_" in l]
            waits = [l.strip() for l in loop if re.search(r"s_waitcnt.*vmcnt\(\d+\)", l)]
            # (hand-counted: 16 = four stages in flight; 24 with CUKING_MFMA_PAIRED_STAGES=10;
            #  20 = the four-product form's hand-over)
            foreign = [w for w in waits
                       if not any(f"vmcnt({n})" in w for n in (16, 20, 24))]
            if verbose:
                print(f"{label}: loop {head} ({mfma} MFMAs): scratch {len(scratch)}, "
                      f"vmcnt waits {waits}")
            if scratch or foreign:
                problems.append(f"{label}, loop {head}: scratch {scratch[:2]}, waits {foreign}")
        if seen == 0:
            problems.append(f"{label}: no LDS-DMA loop found (listing format changed?)")
    return problems


def arrow_flags():
    """Arrow/Parquet C++ headers and libraries bundled with the pyarrow wheel
    (the reference links Arrow 8.0.0, Dockerfile:116)."""
    import pyarrow
    inc = pyarrow.get_include()
    libdir = pyarrow.get_library_dirs()[0]
    libs = []
    for stem in ("parquet", "arrow"):
        cands = sorted(Path(libdir).glob(f"lib{stem}.so.*"))
        if not cands:
            raise RuntimeError(f"lib{stem}.so not found in {libdir}")
        libs.append(f"-l:{cands[0].name}")
    return inc, libdir, libs


def build_cli(force: bool = False) -> Path:
    srcs = sorted(HOST.glob("*.cc"))
    if not srcs:
        raise RuntimeError("no host sources")
    deps = srcs + sorted(HOST.glob("*.h")) + [INCLUDE / "cuking_amd.h",
                                              LIB_PATH, Path(__file__)]
    if not force and _newer(CLI_PATH, deps):
        return CLI_PATH
    inc, libdir, libs = arrow_flags()
    CLI_PATH.parent.mkdir(exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++20", "-Wall", "-pthread",
           f"-I{INCLUDE}", f"-I{HOST}", f"-isystem{inc}", *ROCM_HOST_FLAGS,
           *map(str, srcs), "-o", str(CLI_PATH),
           f"-L{PKG}", "-l:libcuking_amd.so", f"-L{libdir}", *libs, *ROCM_HOST_LIBS,
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,$ORIGIN/..",
           "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return CLI_PATH


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", action="store_true", help="library only")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--save-temps", action="store_true")
    ap.add_argument("--tuning", action="store_true")
    args = ap.parse_args(argv)
    print(build_library(force=args.force or args.tuning, save_temps=args.save_temps,
                        tuning=args.tuning))
    if not args.lib:
        print(build_cli(force=args.force))
    return 0


if __name__ == "__main__":
    sys.exit(main())
