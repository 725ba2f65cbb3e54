"""The C-ABI library loads and exports every symbol include/thesia_amd.h declares; without a
GPU every compute entry fails loudly (no CPU fallback).  CPU only."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("thesia_amd.h", "thesia_amd_testing.h")   # the product interface | test and measurement entry points
TESTING_ONLY = {"th_plan_set_kernel", "th_build_ab_variants", "th_plan_mel_moments_info", "th_plan_time_kernel", "th_plan_kernel_ms_history", "th_plan_last_kernel_ms", "th_tm_put_img"}


def header_symbols(headers=HEADERS):
    out = set()
    for h in headers:
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        out |= set(re.findall(r"TH_API\s+[\w\s\*]+?\b(th_\w+)\s*\(", txt))
    return sorted(out)


def test_header_declares_expected_surface():
    syms = header_symbols()
    assert len(syms) >= 45
    for must in ("th_calc_spec_batch_dev", "th_spec_to_img_dev", "th_encode_spectrogram_tile_dev",
                 "th_encode_waveform_tile_dev", "th_tm_apply_track_list_changes", "th_plan_create"):
        assert must in syms


def test_testing_entry_points_are_not_in_the_product_header():
    """kernel selectors, kernel timing and th_tm_put_img have no reference counterpart — they live in
    include/thesia_amd_testing.h, and the reference-side binding of INTEGRATION.md names none of them."""
    product, testing = set(header_symbols(HEADERS[:1])), set(header_symbols(HEADERS[1:]))
    assert testing == TESTING_ONLY and not (product & TESTING_ONLY)
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    binding = "\n".join(re.findall(r"```rust(.*?)```", integ, flags=re.S))
    assert binding and not any(sym in binding for sym in TESTING_ONLY)


def test_library_exports_every_header_symbol():
    import thesia_amd
    lib = C.CDLL(thesia_amd.LIB_PATH)
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_python_binding_declares_every_header_symbol():
    from thesia_amd import _ffi
    declared = set(_ffi._SIGS) | {"th_last_error"}
    assert set(header_symbols()) <= declared, set(header_symbols()) - declared


def test_compute_entries_fail_loudly_without_gpu():
    import thesia_amd as ta
    if ta.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(ta.ThError) as e:
        ta.Context(0)
    assert e.value.code == -4 and "no CPU fallback" in str(e.value)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under thesia_amd/ may import, link or call it."""
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b)|libthesia_oracle|\borc_\w+\s*\(|oracle[/\\]_build|oracle\.oracle",
                     re.M)
    for dp, _, fns in os.walk(os.path.join(ROOT, "thesia_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert not bad.search(txt), (dp, fn, bad.search(txt).group(0))
