"""Host side of libfst_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer (race / memory checking).

GPU ASan is not available on this pool; what CAN be sanitized is everything the launchers do on the host — argument and extent
checks, plan walking, pointer tables, launch geometry.  `hipcc --cuda-host-only -fsanitize=address,undefined` builds that half
of every csrc/*.hip (the device code objects are replaced by empty stand-ins: nothing is launched), and
tests/host_sanitize/driver.py drives every entry point up to its launch."""
import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "feature_level_style_transfer_for_tsc_amd", "csrc")
OUT = os.path.join(ROOT, "build", "san")


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _asan_runtime():
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return hits[0] if hits else None


@pytest.mark.skipif(torch.cuda.is_available(), reason="the host-only build must not meet a real device (its code objects are stand-ins)")
@pytest.mark.skipif(not os.path.exists(_hipcc()) or _asan_runtime() is None, reason="needs hipcc and clang's ASan runtime")
def test_host_code_under_asan_and_ubsan():
    os.makedirs(OUT, exist_ok=True)
    flags = ["--cuda-host-only", "-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
             "-fno-omit-frame-pointer", "--offload-arch=gfx950", "-w"]
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))

    def cc(src):
        obj = os.path.join(OUT, os.path.basename(src)[:-4] + ".o")
        subprocess.run([_hipcc(), *flags, "-c", src, "-o", obj], check=True, capture_output=True)
        return obj

    with ThreadPoolExecutor(4) as ex:
        objs = list(ex.map(cc, srcs))
    lib = os.path.join(OUT, "libfst_hip_san.so")
    subprocess.run([_hipcc(), "-shared", "-fPIC", "-fsanitize=address,undefined", *objs, "-o", lib], check=True, capture_output=True)
    # the host-only objects still reference their device code objects (__hip_fatbin_<hash>): empty stand-ins
    undefined = subprocess.run(["nm", "-u", lib], check=True, capture_output=True, text=True).stdout
    syms = [l.split()[-1] for l in undefined.splitlines() if "__hip_fatbin_" in l]
    assert len(syms) == len(srcs)
    stub = os.path.join(OUT, "fatbin_stubs.cpp")
    with open(stub, "w") as f:
        f.write("// empty stand-ins for the device code objects of the host-only sanitizer build: nothing is ever launched from it\n")
        f.writelines(f'extern "C" const char {s}[64] = {{0}};\n' for s in syms)
    subprocess.run(["g++", "-c", "-fPIC", stub, "-o", stub[:-4] + ".o"], check=True)
    subprocess.run([_hipcc(), "-shared", "-fPIC", "-fsanitize=address,undefined", *objs, stub[:-4] + ".o", "-o", lib], check=True,
                   capture_output=True)
    env = dict(os.environ, LD_PRELOAD=_asan_runtime(), FST_HIP_LIB=lib, PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_sanitize", "driver.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-4000:]
    assert "AddressSanitizer" not in log and "runtime error:" not in log, log[-4000:]
    assert "OK:" in r.stdout
