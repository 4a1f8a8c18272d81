"""Build recipe of libcals_hip.so (hipcc, gfx950 only).  Used by __graft_entry__.build()."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcals_hip.so")
SOURCES = ["mttkrp_kernel_v3.hip", "ttm_kernel.hip", "model_kernels.hip", "nnls_kernel.hip", "cals_hip_engine.cpp"]
API_SOURCES = ["cals.cpp", "als.cpp", "tensor.cpp", "ktensor.cpp", "multi_ktensor.cpp", "utils.cpp"]
HEADERS = ["cals_hip_internal.h", "mfma_common.h", os.path.join("..", "..", "include", "cals_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wall", "-Wno-unused-result"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    # objects built with other -D switches (CALS_DIAG, ring depths) must not survive into this build
    # CALS_EXTRA_DEFS="-DNAME=1 ...": experiment switches (timing-only variants, see the kernels' #ifndef blocks)
    variant = " ".join("%s=%s" % (k, os.environ[k])
                       for k in ("CALS_V3_RING", "CALS_DIAG", "CALS_TTM_RING", "CALS_EXTRA_DEFS")
                       if os.environ.get(k)) or "production"
    stamp = os.path.join(HERE, "build", "variant.txt")
    if not os.path.exists(stamp) or open(stamp).read() != variant:
        force = True
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(HERE, "build", os.path.splitext(s)[0] + ".o")
        if force or _newer(obj, [src] + hdrs):
            extra = ["-DCALS_V3_RING=%s" % os.environ["CALS_V3_RING"]] if os.environ.get("CALS_V3_RING") else []
            if os.environ.get("CALS_DIAG"):
                extra.append("-DCALS_DIAG=1")
            if os.environ.get("CALS_TTM_RING"):
                extra.append("-DCALS_TTM_RING=%s" % os.environ["CALS_TTM_RING"])
            extra += os.environ.get("CALS_EXTRA_DEFS", "").split()
            cmd = [hipcc] + FLAGS + extra + ["-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    with open(stamp, "w") as f:
        f.write(variant)
    if force or _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # C++ host layer (the reference's header set: cals::cp_cals, Tensor / Matrix / Ktensor / MultiKtensor,
    # cp_als ..., utils) over the C ABI
    api_dir = os.path.join(HERE, "cals")
    api_srcs = [os.path.join(api_dir, f) for f in API_SOURCES]
    api_hdrs = [os.path.join(dp, f) for dp, _, fs in os.walk(api_dir) for f in fs if f.endswith(".h")]
    api_lib = os.path.join(HERE, "libcals.so")
    if force or _newer(api_lib, api_srcs + api_hdrs + [LIB, os.path.join(HERE, "..", "include", "cals_hip.h")]):
        cmd = ["g++", "-std=c++17", "-O2", "-g", "-Wall", "-Wextra", "-fPIC", "-pthread", "-shared", "-I" + api_dir,
               "-I" + os.path.join(api_dir, "utils"), "-o", api_lib] + api_srcs + [
               "-L" + HERE, "-lcals_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_variant(name, defs, sources=None, verbose=False):
    """An experiment build next to the product: build/variants/libcals_hip_<name>.so from the sources listed in
    `sources` (default: all) compiled with the extra -D switches `defs`, the other objects taken from the
    production build.  Selected per process with CALS_HIP_LIB=<path> (cp_cals_amd.load_library; tools/ab_libs.sh);
    the shipped libcals_hip.so is not touched."""
    build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    vdir = os.path.join(HERE, "build", "variants")
    os.makedirs(os.path.join(vdir, name), exist_ok=True)
    objs = []
    for s in SOURCES:
        prod = os.path.join(HERE, "build", os.path.splitext(s)[0] + ".o")
        if sources is not None and s not in sources:
            objs.append(prod)
            continue
        obj = os.path.join(vdir, name, os.path.splitext(s)[0] + ".o")
        cmd = [hipcc] + FLAGS + list(defs) + ["-x", "hip", "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    lib = os.path.join(vdir, "libcals_hip_%s.so" % name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    import sys
    if len(sys.argv) >= 3 and sys.argv[1] == "variant":  # build.py variant <name> "<-D...>" [source ...]
        print(build_variant(sys.argv[2], sys.argv[3].split() if len(sys.argv) > 3 else [], sys.argv[4:] or None, verbose=True))
    else:
        print(build(verbose=True))
