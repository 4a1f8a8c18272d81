"""CPU checks of the C++ drop-in surface (cp-cals_amd/cals/: the reference's header set over the C ABI).

  * the reference's OWN CLI driver (src/examples/driver.cpp) compiles and links, unmodified, against
    this header set and libcals.so -- where /root/reference exists (this container; nothing of the
    reference is copied, the compiler reads it where it lies and the outputs go to a temp directory);
  * tests/cpp/ref_style_caller.cpp -- an own caller with the same includes / symbols / call pattern as
    the reference's driver, MEX glue and experiment CSV writers -- compiles (it RUNS in the -m gpu suite);
  * tests/cpp/test_host_api: value classes, MultiKtensor packing, tensor file reader, jackknife helpers,
    report writers -- plain and under AddressSanitizer + UBSan;
  * the engine's host side (cals_hip_engine.cpp) + the C++ layer under ASan/UBSan on a fake device;
  * the oracle's own suite re-run on an ASan/UBSan build of oracle/cals_oracle.c.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALS = os.path.join(ROOT, "cp-cals_amd", "cals")
INC = ["-I" + CALS, "-I" + os.path.join(CALS, "utils")]
REF_DRIVER = "/root/reference/src/examples/driver.cpp"


def test_header_set_has_the_reference_names():
    for h in ("cals.h", "als.h", "tensor.h", "matrix.h", "ktensor.h", "multi_ktensor.h", "timer.h", "cals_blas.h",
              "definitions.h", "utils/utils.h", "utils/mttkrp.h", "utils/update.h", "utils/line_search.h",
              "utils/error.h", "rectangular_lsap/rectangular_lsap.h"):
        assert os.path.exists(os.path.join(CALS, h)), h


@pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="the reference tree is not present on this machine")
def test_reference_driver_compiles_and_links_unmodified(tmp_path):
    exe = str(tmp_path / "reference_driver")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall"] + INC + [REF_DRIVER, "-o", exe, "-L" + os.path.join(ROOT, "cp-cals_amd"),
                                                        "-lcals", "-lcals_hip",
                                                        "-Wl,-rpath," + os.path.join(ROOT, "cp-cals_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    # it runs up to the point where it needs the GPU: usage text and argument errors are the reference's own
    r = subprocess.run([exe, "--help"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "--components MIN:MAX:COPIES" in r.stdout
    r = subprocess.run([exe, "-c", "1:2"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "MIN:MAX:COPIES" in r.stderr


REF = "/root/reference"


@pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="the reference tree is not present on this machine")
@pytest.mark.parametrize("main_src", ["experiments.cpp", "experiments_jk.cpp", "experiments_letter.cpp"])
@pytest.mark.parametrize("with_time", [0, 1])
def test_reference_experiment_harness_compiles_and_links_unmodified(tmp_path, main_src, with_time):
    """The reference's paper-experiment programs (src/experiments/experiments{,_jk,_letter}.cpp +
    experiments_utils.cpp: compare_als_cals, run_cals / run_als / run_omp_als / run_jk_*, report CSVs with
    the WITH_TIME timer matrices) against this header set and libcals.so, unmodified.  The one reference
    header they need beyond the boundary's is their own include/experiments/experiments_utils.h, found through
    a trailing -I of the reference's include/ (every boundary header resolves to cp-cals_amd/cals/ first)."""
    exe = str(tmp_path / "prog")
    srcs = [os.path.join(REF, "src", "experiments", main_src), os.path.join(REF, "src", "experiments", "experiments_utils.cpp")]
    cmd = ["g++", "-std=c++17", "-O0", "-fopenmp", '-DSOURCE_DIR="/tmp"', "-DWITH_TIME=%d" % with_time] + INC + [
        "-I" + os.path.join(REF, "include")] + srcs + ["-o", exe, "-L" + os.path.join(ROOT, "cp-cals_amd"), "-lcals",
                                                       "-lcals_hip", "-Wl,-rpath," + os.path.join(ROOT, "cp-cals_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # every header of the boundary came from this repo, not from the reference tree
    deps = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-fopenmp", "-H", '-DSOURCE_DIR="/tmp"'] + INC + [
        "-I" + os.path.join(REF, "include"), srcs[1]], capture_output=True, text=True, timeout=300).stderr
    used = [ln.split()[-1] for ln in deps.splitlines() if ln.startswith(".") and "/root/reference" in ln]
    assert used and all(u.endswith("experiments/experiments_utils.h") for u in used), used


@pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="the reference tree is not present on this machine")
def test_reference_mttkrp_microbenchmark_compiles_and_links_unmodified(tmp_path):
    """src/experiments/benchmark_cals_mttkrp.cpp with include/experiments/bench_mttkrp{,_cals,_ctf,_planc}.h and
    bench_utils.h -- the protocol SURVEY section 8(d) cites (bench_mttkrp_cals.h:49-84) -- against this header
    set: needs the free function mttkrp::mttkrp(const Tensor &, Ktensor &, vector<Matrix> &, dim_t,
    MttkrpParams &) as a linkable symbol (here: one fused MTTKRP launch on a leased engine, cals_hip_mttkrp)."""
    exe = str(tmp_path / "bench_mttkrp")
    cmd = ["g++", "-std=c++17", "-O0", "-fopenmp", '-DSOURCE_DIR="/tmp"'] + INC + ["-I" + os.path.join(REF, "include"),
           os.path.join(REF, "src", "experiments", "benchmark_cals_mttkrp.cpp"), "-o", exe,
           "-L" + os.path.join(ROOT, "cp-cals_amd"), "-lcals", "-lcals_hip", "-Wl,-rpath," + os.path.join(ROOT, "cp-cals_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, "--help"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "USAGE" in r.stdout


@pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="the reference tree is not present on this machine")
def test_reference_mex_argument_parser_compiles_unmodified():
    """matlab/matlab_parsing.cpp (the string-argument parser of the three MEX entry points: update-method,
    mttkrp-method, maxiters, buffer-size, tol, cuda / no-cuda, ls / no-ls, ls-interval, ls-step on CalsParams)
    needs no MATLAB header: it compiles against this header set as it is."""
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall"] + INC + ["-I" + os.path.join(REF, "matlab"),
                       os.path.join(REF, "matlab", "matlab_parsing.cpp")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_reference_style_caller_compiles():
    src = os.path.join(ROOT, "tests", "cpp", "ref_style_caller.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall"] + INC + [src], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def _run(exe, tmp_path, env=None):
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600, env=e)
    print(r.stdout[-3000:], r.stderr[-6000:])
    return r


def test_host_api(tmp_path):
    r = _run(os.path.join(ROOT, "tests", "cpp", "test_host_api"), tmp_path)
    assert r.returncode == 0 and "all checks passed" in r.stdout


def test_host_api_under_asan_ubsan(tmp_path):
    r = _run(os.path.join(ROOT, "tests", "cpp", "test_host_api_asan"), tmp_path,
             {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"})
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_engine_host_side_under_asan_ubsan_on_a_fake_device():
    """cals_hip_engine.cpp's host logic + the C++ layer, life cycles with queueing / eviction / compress /
    rebind / two engines / the CLI driver's call pattern, under ASan + UBSan + LeakSanitizer + libstdc++ assertions on
    the fake device of tests/asan/fake_device.cpp (numeric kernels are no-ops there, so every model must come back
    bit-identical) -- once with a device that completes every stream operation at once and once with one that is as
    late as the API allows -- and the replay of the two round-3 anomalies' call patterns with the eviction schedule
    of the real runs (tests/asan/patterns.txt, from the oracle: tools/make_asan_patterns.py), CALS_HIP_VERIFY on."""
    exe = os.path.join(ROOT, "tests", "asan", "test_engine_host_asan")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "asan", "patterns.txt")], capture_output=True, text=True,
                       timeout=900, env=env)
    print(r.stdout[-2000:], r.stderr[-6000:])
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr


def test_engine_host_side_under_thread_sanitizer():
    """The same program under ThreadSanitizer: CalsParams::devices = {0, 0} runs two engines from two host threads
    that claim models from one queue through a shared atomic counter (cals/cals.cpp: cp_cals_devices)."""
    exe = os.path.join(ROOT, "tests", "asan", "test_engine_host_tsan")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1", CALS_HARNESS_ONLY_LATE="1")
    # (ASLR off for the child: this kernel's 32-bit mmap entropy trips older TSan runtimes, "unexpected memory mapping")
    cmd = [exe, os.path.join(ROOT, "tests", "asan", "patterns.txt")]
    if os.path.exists("/usr/bin/setarch"):
        cmd = ["/usr/bin/setarch", os.uname().machine, "-R"] + cmd
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, env=env)
    print(r.stdout[-2000:], r.stderr[-6000:])
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stderr[-4000:]
    assert "WARNING: ThreadSanitizer" not in r.stderr


ORACLE_UNDER_ASAN = r"""
import ctypes, os, sys
sys.path.insert(0, os.path.join({root!r}, "oracle")); sys.path.insert(0, {root!r})
import numpy as np
import oracle as O
assert "asan" in os.environ["CALS_ORACLE_LIB"]
O.lib()
from cp_cals_amd import inputs
modes = [9, 7, 5]
X = inputs.tensor(modes, 3)
ranks = [1, 4, 2, 5, 3, 2, 4, 1]
def models(jk=None):
    out = []
    for k, (fs, lam) in enumerate(inputs.model_factors(modes, ranks, 2)):
        j = None if jk is None else (0, k % modes[0])
        if j: fs[0][j[1], :] *= 0.0
        out.append(O.Model(fs, lam, jk=j))
    return out
for kw in (dict(), dict(line_search=1, line_search_interval=3), dict(line_search=1, line_search_method=1),
           dict(update_method=1), dict(mttkrp_method=O.TWOSTEP0), dict(mttkrp_method=O.TWOSTEP1)):
    for jk in (None, True):
        ms = models(jk)
        rep = O.cp_cals(X, modes, ms, O.default_params(max_iterations=40, tol=1e-6, buffer_size=7, **kw))
        assert rep.n_ktensors == len(ranks)
m4 = [6, 5, 4, 3]
X4 = inputs.tensor(m4, 1)
ms = [O.Model(fs, lam) for fs, lam in inputs.model_factors(m4, [2, 3], 4)]
O.cp_cals(X4, m4, ms, O.default_params(max_iterations=10, force_max_iter=1, buffer_size=5))
print("oracle under asan ok")
"""


def test_oracle_under_asan_ubsan(tmp_path):
    """queue / eviction / compress / line search / NNLS / jackknife / 4-way life cycles of the oracle on
    its sanitizer build (the library is loaded into python with the ASan runtime preloaded)."""
    lib = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    assert os.path.exists(lib), "run __graft_entry__.build() first"
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan_rt):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="2", CALS_ORACLE_LIB=lib)
    script = tmp_path / "run.py"
    script.write_text(ORACLE_UNDER_ASAN.format(root=ROOT))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-2000:], r.stderr[-6000:])
    assert r.returncode == 0 and "oracle under asan ok" in r.stdout, r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
