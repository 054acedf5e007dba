"""CPU-side AddressSanitizer build of the library's host-only paths (bincode readers of LeannIndex
and HnswGraph, the storage.rs chunk framing, host CSR accessors, shard-record arithmetic): built
from the library's own sources with the host code instrumented (`make -C islands_amd/csrc asan`)
and driven by tests/cpp/asan_host_paths.cpp over valid images, every truncation, wrapping length
fields and seeded byte flips.  ASan aborts on any access outside a buffer; LeakSanitizer fails the
run on a leak.  (GPU ASan is not available on this pool; the device paths have the parity suite.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_host_paths_under_address_sanitizer():
    csrc = os.path.join(ROOT, "islands_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "asan", "-j", "4", "-s"])
    exe = os.path.join(ROOT, "islands_amd", "lib", "asan", "asan_host_paths")
    env = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:detect_leaks=1:halt_on_error=1")
    pr = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = pr.stdout.decode(errors="replace")
    assert pr.returncode == 0 and "asan host paths: ok" in out and "AddressSanitizer" not in out, out[-3000:]
