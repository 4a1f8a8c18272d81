#!/usr/bin/env python3
"""Register / scratch / LDS figures of every kernel in a built object or in libcals_hip.so, from the code
object's own metadata (llvm-readelf --notes) -- NOT rocprofv3's halved VGPR column.

  python tools/kernel_resources.py [cp-cals_amd/build/ttm_kernel.o | cp-cals_amd/libcals_hip.so] [name filter]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path):
    d = tempfile.mkdtemp(prefix="kres")
    tmp = os.path.join(d, os.path.basename(path))
    subprocess.check_call(["cp", path, tmp])
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp], check=True, capture_output=True)
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if "amdgcn" in f]


def kernels(co):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True,
                         text=True).stdout
    for blk in txt.split("- .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        get = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        yield {k: get(k) for k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count",
                                   "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")}


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "cp-cals_amd/libcals_hip.so"
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print("%-90s %5s %5s %5s %6s %6s %8s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds"))
    for co in code_objects(path):
        for k in kernels(co):
            name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
            if flt and flt not in name:
                continue
            print("%-90s %5s %5s %5s %6s %6s %8s %8s" % (name[:90], k["vgpr_count"], k["agpr_count"], k["sgpr_count"],
                                                     k["vgpr_spill_count"], k["sgpr_spill_count"],
                                                     k["private_segment_fixed_size"], k["group_segment_fixed_size"]))


if __name__ == "__main__":
    main()
