"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/rkh.h declares.  No compute call is made here (no GPU in this container)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from reak_amd import lib as L

    L.build()
    return L.load()


def _declared():
    hdr = open(os.path.join(ROOT, "include", "rkh.h")).read() + open(os.path.join(ROOT, "include", "rkh_diag.h")).read()
    return set(re.findall(r"\b(rkh_[a-z0-9_]+)\s*\(", hdr)) - {"rkh_status"}


def test_header_symbols_are_exported(lib):
    declared = _declared()
    assert len(declared) >= 30
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"declared in include/rkh.h but not exported by librkh.so: {missing}"


def test_python_binding_covers_the_header(lib):
    from reak_amd import lib as L

    assert set(L.EXPORTS) == _declared()


def test_diagnostics_stay_out_of_the_boundary_header():
    """Profiling / diagnostic entry points live in rkh_diag.h; rkh.h declares only what replaces a reference interface."""
    hdr = open(os.path.join(ROOT, "include", "rkh.h")).read()
    for name in ("rkh_diag_feval_cycles", "rkh_nn_set_events", "rkh_nn_kernel_name", "rkh_planner_nn_profile",
                 "rkh_planner_nn_pairs", "rkh_planner_steer_profile"):
        assert name + "(" not in hdr


def test_pod_layouts_match_the_header():
    """sizeof of the ctypes mirrors == sizeof of the C structs (compiled with gcc from include/rkh_types.h)."""
    import subprocess
    import tempfile

    from reak_amd import types as T

    src = '#include <stdio.h>\n#include "rkh_types.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n",sizeof(rkh_pose),' \
          'sizeof(rkh_kte_op),sizeof(rkh_chain_base),sizeof(rkh_shape),sizeof(rkh_dyn_space),sizeof(rkh_rrt_params),' \
          'sizeof(rkh_kte_op)-sizeof(double)*7);return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(td, "t.c"), "-o", os.path.join(td, "t")], check=True)
        out = subprocess.run([os.path.join(td, "t")], check=True, capture_output=True, text=True).stdout.split()
    sizes = [int(v) for v in out[:6]]
    assert sizes == [C.sizeof(T.Pose), C.sizeof(T.KteOp), C.sizeof(T.ChainBase), C.sizeof(T.Shape), C.sizeof(T.DynSpace),
                     C.sizeof(T.RrtParams)]


def test_version_and_error_strings(lib):
    assert b"gfx950" in lib.rkh_version() and b"ABI 3" in lib.rkh_version()
    assert isinstance(lib.rkh_last_error(), bytes)


def test_abi_handshake(lib):
    """rkh_abi_check refuses a caller whose view of the public PODs differs from the library's (another header version):
    the Python binding passes with its mirrors, a wrong ABI number or one wrong struct size is an error, not a later
    overrun of a stats array."""
    from reak_amd import lib as L

    assert lib.rkh_abi_version() == L.ABI_VERSION == 3
    sizes = L.abi_sizes()
    assert lib.rkh_abi_check(L.ABI_VERSION, *sizes) == 0
    assert lib.rkh_abi_check(L.ABI_VERSION - 1, *sizes) != 0
    for k in range(len(sizes)):
        bad = list(sizes)
        bad[k] -= 8
        assert lib.rkh_abi_check(L.ABI_VERSION, *bad) != 0, k
    assert b"another version" in lib.rkh_last_error()


def test_no_gpu_fails_loudly(lib):
    """Without a GPU the product path must fail, not fall back to anything on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from reak_amd import lib as L

    with pytest.raises(L.RkhError):
        L.Context(0)
