"""CPU: the host-side native code under AddressSanitizer + UndefinedBehaviorSanitizer (GPU ASan is not available on the pool, and these
pieces have no device code): the segment-schedule builder every balanced SpMM launch trusts blindly (csrc/segments.h) and the text
readers / writer (csrc/textio.hip).  The drivers (tests/native/*.cpp) assert the invariants; any sanitizer report fails the run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gcn-drug-repurposing_amd", "csrc")
SAN = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def _run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, **kw)
    assert r.returncode == 0, (cmd, r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    return r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_segment_schedule_builder_under_sanitizers(tmp_path):
    exe = str(tmp_path / "segments_check")
    _run(["g++"] + SAN + ["-I" + CSRC, os.path.join(ROOT, "tests", "native", "segments_check.cpp"), "-o", exe])
    out = _run([exe], env=ENV)
    assert "segments ok: 1800 schedules" in out        # 1500 whole-CSR schedules + 300 over the item lists of the giant-row views


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (host-only compile of a .hip file)")
def test_text_io_under_sanitizers(tmp_path):
    exe = str(tmp_path / "textio_check")
    _run(["hipcc", "--offload-host-only", "-Wno-unused-value"] + SAN + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
          os.path.join(CSRC, "textio.hip"), "-x", "c++", os.path.join(ROOT, "tests", "native", "textio_check.cpp"), "-o", exe], cwd=str(tmp_path))
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    out = _run([exe, str(scratch)], env=ENV)
    assert "textio ok" in out
