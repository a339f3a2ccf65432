"""AddressSanitizer + UndefinedBehaviorSanitizer over the host-side code (the reference ships no race / memory
checking of its own; this is the product's).  Two instrumented builds, each driven by the CPU test files that exercise it, in a
subprocess with the sanitizer runtime preloaded (the interpreter itself is not instrumented):
  * oracle/srl_oracle.cpp (g++ -fsanitize=address,undefined) under tests/test_oracle.py, test_heap_replay.py, test_eigen_solver.py;
  * every host translation unit of libsrlivo_hip.so -- the C-ABI (srl_capi.cpp), the RCCL table, the host mirror
    (csrc/host/*.cpp) -- under tests/test_host_logic.py, test_tr1_order.py.  Device code is not instrumented.
Any report (heap / stack overflow, use after free, signed overflow, misaligned or null access, out-of-range shift or cast ...)
aborts the subprocess: -fno-sanitize-recover, halt_on_error."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}


def _run_pytest(env_extra, files):
    env = dict(os.environ, **SAN_ENV, **env_extra)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu"] + files, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, out[-4000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-4000:]
    return out


def test_oracle_under_asan_and_ubsan(tmp_path):
    gxx = shutil.which("g++")
    rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE).stdout.decode().strip() if gxx else ""
    if not gxx or not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("g++ / libasan not available")
    out_dir = str(tmp_path / "oracle_san")
    b = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "san", f"SAN_OUT={out_dir}"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert b.returncode == 0, b.stdout.decode(errors="replace")[-3000:]
    env = {"LD_PRELOAD": rt, "ORC_LIB_PLAIN": os.path.join(out_dir, "liboracle.so")}
    if os.path.exists(os.path.join(out_dir, "liboracle_tsl.so")):
        env["ORC_LIB_TSL"] = os.path.join(out_dir, "liboracle_tsl.so")
    else:
        env["ORC_LIB_TSL"] = os.path.join(out_dir, "absent.so")          # never mix the instrumented build with an uninstrumented twin
    out = _run_pytest(env, ["tests/test_oracle.py", "tests/test_heap_replay.py", "tests/test_eigen_solver.py"])
    assert " passed" in out


def test_host_side_of_the_product_under_asan_and_ubsan(tmp_path):
    hipcc = "/opt/rocm/bin/hipcc"
    rts = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not os.path.exists(hipcc) or not rts:
        pytest.skip("hipcc / clang ASan runtime not available")
    out_dir = str(tmp_path / "product_san")
    csrc = os.path.join(ROOT, "sr_livo_amd", "csrc")
    b = subprocess.run(["make", "-C", csrc, "san", f"SAN_OUT={out_dir}"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    assert b.returncode == 0 and os.path.exists(os.path.join(out_dir, "libsrlivo_hip.so")), b.stdout.decode(errors="replace")[-3000:]
    env = {"LD_PRELOAD": rts[0], "SRL_LIB_PATH": os.path.join(out_dir, "libsrlivo_hip.so")}
    out = _run_pytest(env, ["tests/test_host_logic.py", "tests/test_tr1_order.py", "tests/test_capi_symbols.py"])
    assert " passed" in out
