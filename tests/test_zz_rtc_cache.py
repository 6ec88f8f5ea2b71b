"""Runs LAST in the GPU suite (file name): the run-time compiled variants the suite's custom-drift tests asked for in THIS process must
have come from the on-disk code-object cache that travels with the tree (cd_dynamax_amd/lib/rtc_cache, its MANIFEST tracked) --
fewer than 90 % hits means the cache is stale (a ROCm / hipRTC bump, changed kernel headers or options: the key covers them all) and the
suite took 2-3 x as long as it should: rebuild it with `CDKF_RTC_CACHE_DIR=gpurun_out/rtc_cache python -m pytest tests -m gpu` on a GPU
box and copy the directory to cd_dynamax_amd/lib/rtc_cache."""
import ctypes as C
import glob
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_the_suite_ran_on_cached_code_objects(hip_lib):
    cache = os.path.join(ROOT, "cd_dynamax_amd", "lib", "rtc_cache")
    if os.environ.get("CDKF_RTC_CACHE") == "0" or os.environ.get("CDKF_RTC_CACHE_DIR") or len(glob.glob(os.path.join(cache, "*.co"))) < 20:
        pytest.skip("no in-tree code-object cache in use (a clean checkout, or the cache is being rebuilt)")
    hits, misses = C.c_int64(0), C.c_int64(0)
    hip_lib.cdkf_rtc_cache_stats(C.byref(hits), C.byref(misses))
    total = hits.value + misses.value
    if total < 20:
        pytest.skip(f"only {total} run-time compiled variants were requested in this process (run the whole suite)")
    assert hits.value >= 0.9 * total, (f"{misses.value} of {total} run-time compiled variants missed the in-tree cache: it is stale for this "
                                       "toolchain / these headers -- rebuild it (this file's docstring)")
