"""The CPU checker and the splitter of the sharded sort under AddressSanitizer and
UndefinedBehaviorSanitizer (host code only; `make -C oracle sanitize`).

The reference's prefix-sum shaders carry latent races and unchecked indexing
(src/shaders/radix_sort_block_sum_small.wgsl:60-62, radix_sort_block_sum_large.wgsl:155-159); the
restatements of that code used as the checker here, and the host-side cut logic of the sharded
sort, are run once with the sanitizers so that an out-of-bounds index or an overflow there is a
failing test, not a silently wrong expectation."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None, reason="needs g++ and make")
def test_checker_and_splitter_run_clean_under_sanitizers():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert "sanitize ok" in out.stdout
