"""CPU: the C-ABI shared library loads and exports every symbol include/ccal.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

from camera_intrinsic_calibration_rs_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ccal.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ccal_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    declared = _declared()
    bound = sorted(n for n, _, _ in _ffi.SYMBOLS)
    assert declared == bound, (set(declared) ^ set(bound))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_ffi.LIB_PATH), "libccal_hip.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"missing export {name}"


def test_library_exports_nothing_beyond_the_header():
    """Every ccal_* symbol the product build exports is declared in include/ccal.h (developer hooks live behind build
    macros such as -DCCAL_STAMPS and are not in the shipped library)."""
    import subprocess
    nm = "/opt/rocm/lib/llvm/bin/llvm-nm"
    if not os.path.exists(nm):
        nm = "nm"
    out = subprocess.run([nm, "-D", "--defined-only", _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and re.fullmatch(r"ccal_[a-z0-9_]+", ln.split()[-1])})
    assert exported, "no ccal_* exports found"
    extra = sorted(set(exported) - set(_declared()))
    assert not extra, f"exported but not declared in include/ccal.h: {extra}"


def test_no_test_hooks_in_the_product_library():
    """Fault injection (CCAL_TEST_FAIL_SHARD) is compiled only into the second library the tests load (-DCCAL_TEST_HOOKS);
    the product .so does not even contain the string."""
    blob = open(_ffi.LIB_PATH, "rb").read()
    assert b"CCAL_TEST_" not in blob
    assert b"CCAL_TEST_FAIL_SHARD" in open(_ffi.LEGACY_LIB_PATH, "rb").read()


def test_host_only_entry_points():
    lib = _ffi.load()
    assert lib.ccal_version().startswith(b"ccal-mi355x")
    assert [lib.ccal_model_num_params(m) for m in range(6)] == [5, 6, 8, 9, 8, -1]      # 4 = the EUCMT parameter container
    o = _ffi.SolverOpts()
    assert lib.ccal_set_defaults(ctypes.byref(o)) == 0
    # tiny-solver OptimizerOptions::default() as restated in SURVEY 3.3
    assert (o.method, o.max_iterations) == (0, 100)
    assert (o.min_abs_error_decrease, o.min_rel_error_decrease, o.min_error) == (1e-5, 1e-5, 1e-10)
    assert (o.error_metric, o.reserved_) == (_ffi.ERROR_SQUARED_NORM, 0)


def test_solver_opts_layout_is_the_headers():
    """ccal_solver_opts in the header, the ctypes binding and the Rust stub: the same fields in the same order (the struct
    grew by error_metric + reserved_ in round 6: a stale binding would hand the library a short struct)."""
    src = open(os.path.join(ROOT, "include", "ccal.h")).read()
    body = re.search(r"typedef struct \{([^}]*)\} ccal_solver_opts;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:int32_t|double)\s+([a-z_0-9]+)\s*;", body)
    assert fields == [f for f, _ in _ffi.SolverOpts._fields_]
    assert ctypes.sizeof(_ffi.SolverOpts) == 72
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rust = re.search(r"pub struct ccal_solver_opts \{(.*?)\n\}", doc, flags=re.S).group(1)
    assert re.findall(r"pub ([a-z_0-9]+):", rust) == fields


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _ffi.load()
    except _ffi.CcalLibraryMissing as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must raise when the HIP library is missing")


def test_integration_stub_lists_every_entry_point():
    """INTEGRATION.md's Rust `extern "C"` block (what a maintainer of the reference pastes into src/gpu/ffi.rs) names every
    function include/ccal.h declares."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [d for d in _declared() if f"fn {d}(" not in doc]
    assert not missing, missing
