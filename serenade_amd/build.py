"""Builds libserenade_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container; the resulting .so
travels to the GPU box with the source snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libserenade_hip.so")
SYNTH_LIB = os.path.join(HERE, "libsrn_synth.so")
SOURCES = ["srn_index.cpp", "srn_capi.cpp", "srn_batcher.cpp", "srn_combine.cpp", "srn_session_store.cpp", "srn_avro.cpp", "srn_kernels.hip", "srn_fast.hip", "srn_shard.hip", "srn_sback.hip", "srn_runtime.hip", "srn_hostpipe.hip", "srn_group.hip", "srn_build_gpu.hip"]
HEADERS = ["srn_internal.h", "srn_kernels.h", "srn_device.h", "srn_prep.h", "srn_runtime.h", os.path.join("..", "..", "include", "serenade_hip.h")]


EXPERIMENT_ONLY_FLAGS = ("SRN_FAST_EXP_", "SRN_FAST_STOP", "SRN_ABLATE")   # "timing only, wrong results" macros of the kernels


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _includes(src, seen=None):
    """The quoted #include closure of a source file: an object is rebuilt when ITS headers change, not when any header does."""
    seen = set() if seen is None else seen
    import re
    try:
        text = open(src).read()
    except OSError:
        return []
    for inc in re.findall(r'^\s*#include\s+"([^"]+)"', text, flags=re.M):
        path = os.path.normpath(os.path.join(os.path.dirname(src), inc))
        if path not in seen and os.path.exists(path):
            seen.add(path)
            _includes(path, seen)
    return sorted(seen)


def build_hip(force=False, verbose=False):
    """One object per source (rebuilt only when that source or a header changed), then one link."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("SRN_CFLAGS", "").split()          # experiments only (e.g. -DSRN_ABLATE=1)
    # libserenade_hip.so is THE library -- what ships to the GPU box and what every test loads.  Flags that make a kernel compute something else ("timing only, wrong
    # results": SRN_FAST_EXP_*, SRN_FAST_STOP, SRN_FAST_EARLY_CLEAR ...) may only ever produce a variant file (build_variant -> serenade_amd/variants/, loaded through
    # SRN_LIB_PATH); and no SRN_CFLAGS build may replace the default library in place at all unless SRN_CFLAGS_INPLACE=1 says the caller knows
    banned = [f for f in extra if any(t in f for t in EXPERIMENT_ONLY_FLAGS)]
    if banned:
        raise RuntimeError("SRN_CFLAGS holds experiment-only flags %s: they change what the kernels compute and may not be built into libserenade_hip.so -- "
                           "use build.build_variant(name, cflags) and SRN_LIB_PATH" % banned)
    if extra and os.environ.get("SRN_CFLAGS_INPLACE") != "1":
        raise RuntimeError("SRN_CFLAGS=%r would rebuild libserenade_hip.so in place with non-default flags: use build.build_variant(name, cflags) + SRN_LIB_PATH, "
                           "or set SRN_CFLAGS_INPLACE=1 if that is really what you want" % " ".join(extra))
    objdir = os.path.join(CSRC, "_obj")
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    if not force and not extra and not _stale(LIB, [os.path.join(CSRC, n) for n in SOURCES] + headers):
        return LIB                                              # (the objects do not travel to the GPU box; the library does)
    os.makedirs(objdir, exist_ok=True)
    objs, relink, jobs = [], force or not os.path.exists(LIB), []
    for name in SOURCES:
        src, obj = os.path.join(CSRC, name), os.path.join(objdir, name + ".o")
        if force or extra or _stale(obj, [src] + _includes(src)):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-value"]
            if name == "srn_kernels.hip":
                # the predict kernel is ~130 KB of code against a 64 KB instruction cache shared by two CUs: optimising for size
                # measures +1.3 % queries/s over -O3 (-O2 / -O1 -1 %, -Oz -33 %)
                cmd[2] = "-Os"
                # hoisting loop invariants out of the per-query loop keeps them live for ~100 K cycles: a third fewer SGPR spill
                # moves and half the scratch accesses without it, +5..6 % queries/s
                cmd += ["-mllvm", "-disable-machine-licm"]
            if name == "srn_fast.hip":
                # same reason: invariants hoisted out of the per-query loop get spilled (11 -> 4 VGPR spills at the 80-register cap of three workgroups per CU);
                # -Os: 25.77 / 25.82 ms against 25.88 / 25.86 with -O3 (same box, alternating)
                cmd[2] = "-Os"
                cmd += ["-mllvm", "-disable-machine-licm"]
            cmd += extra + ["-c", "-o", obj, src]
            jobs.append(cmd)
            relink = True
        objs.append(obj)
    if jobs:   # the translation units are independent: compile them side by side (srn_build_gpu.hip -- rocPRIM -- and srn_kernels.hip take ~100 s each alone)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), (os.cpu_count() or 2) - 1))) as ex:
            list(ex.map(run, jobs))
    if relink or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_variant(name, cflags, sources=("srn_fast.hip",)):
    """Experiments: the library with `sources` recompiled under extra flags (e.g. -DSRN_FAST_REPLICAS=0), linked with the standard objects of everything else,
    as serenade_amd/variants/libserenade_hip_<name>.so; SRN_LIB_PATH=<that file> makes capi load it.  A/B runs of kernel variants in ONE GPU call."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    build_hip()
    objdir, vdir = os.path.join(CSRC, "_obj"), os.path.join(HERE, "variants")
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for n in SOURCES:
        if n in sources:
            obj = os.path.join(vdir, "%s.%s.o" % (n, name))
            cmd = [hipcc, "--offload-arch=gfx950", "-Os" if n in ("srn_fast.hip", "srn_kernels.hip") else "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-value"]
            if n in ("srn_fast.hip", "srn_kernels.hip"):
                cmd += ["-mllvm", "-disable-machine-licm"]
            subprocess.check_call(cmd + list(cflags) + ["-c", "-o", obj, os.path.join(CSRC, n)])
            objs.append(obj)
        else:
            objs.append(os.path.join(objdir, n + ".o"))
    out = os.path.join(vdir, "libserenade_hip_%s.so" % name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs)
    return out


def build_synth(force=False):
    src = os.path.join(CSRC, "srn_synth.cpp")
    if force or _stale(SYNTH_LIB, [src]):
        subprocess.check_call(["g++", "-O3", "-march=x86-64-v3", "-std=c++17", "-fPIC", "-shared", "-pthread",
                               "-o", SYNTH_LIB, src])
    return SYNTH_LIB


EVALUATOR = os.path.join(HERE, "bin", "evaluator")


def build_evaluator(force=False):
    """Host program over the C ABI (mirror of the reference's evaluator / evaluate_file binaries)."""
    src = os.path.join(CSRC, "host", "evaluator.cpp")
    if force or _stale(EVALUATOR, [src, LIB]):
        os.makedirs(os.path.dirname(EVALUATOR), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", EVALUATOR, src, "-L" + HERE, "-lserenade_hip",
                               "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"])
    return EVALUATOR


SERVE_BENCH = os.path.join(HERE, "bin", "serve_bench")


def build_serve_bench(force=False):
    """Closed-loop load generator for the dynamic batcher (host program over the C ABI)."""
    src = os.path.join(CSRC, "host", "serve_bench.cpp")
    if force or _stale(SERVE_BENCH, [src, LIB]):
        os.makedirs(os.path.dirname(SERVE_BENCH), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", SERVE_BENCH, src, "-L" + HERE, "-lserenade_hip",
                               "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"])
    return SERVE_BENCH


ROW_FETCH_BENCH = os.path.join(HERE, "bin", "row_fetch_bench")


def build_row_fetch_bench(force=False):
    """The random-row gather ceiling of the GPU in the fast kernel's own access shape (tools/row_fetch_bench.hip): bench.py runs it for roofline_secondary's row_gather entry."""
    src = os.path.join(os.path.dirname(HERE), "tools", "row_fetch_bench.hip")
    if force or _stale(ROW_FETCH_BENCH, [src]):
        os.makedirs(os.path.dirname(ROW_FETCH_BENCH), exist_ok=True)
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-Wno-unused-result", "-o", ROW_FETCH_BENCH, src])
    return ROW_FETCH_BENCH


def build_all(force=False, verbose=False):
    build_hip(force, verbose)
    build_synth(force)
    build_evaluator(force)
    build_serve_bench(force)
    build_row_fetch_bench(force)


if __name__ == "__main__":
    build_all(force=True, verbose=True)
