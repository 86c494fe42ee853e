"""Report text formats against a real iostream in the oracle library."""
import ctypes as C

import numpy as np

from reak_amd import reports
import oracle_lib


def _orc_vlist(pos, da=None, de=None):
    lib = oracle_lib.load()
    lib.orc_format_vlist.restype = C.c_int64
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    n, D = pos.shape
    buf = C.create_string_buffer(64 * (D + 2) * max(n, 1) + 16)
    name = C.create_string_buffer(64)
    dp = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))
    sz = lib.orc_format_vlist(pos.ctypes.data_as(C.POINTER(C.c_double)), C.c_uint64(n), D, dp(da), dp(de), buf,
                              C.c_uint64(len(buf)), name, C.c_uint64(64))
    return buf.raw[:sz].decode(), name.value.decode()


def test_vlist_text_matches_iostream(tmp_path):
    rng = np.random.default_rng(5)
    awkward = np.array([0.0, -0.0, 1.0, -1.0, 0.1 + 0.2, 1e-5, 9.9999995e-5, 123456.5, 1234567.0, 999999.5, 1e10, 1e100,
                        -2.5e-310, np.inf, -np.inf, np.pi, 1.0 / 3.0, 100000.0, 1e6, 0.00012345678])
    pos = np.concatenate([awkward, rng.uniform(-np.pi, np.pi, 37), rng.standard_normal(15) * 10.0 ** rng.integers(-12, 12, 15)])
    pos = pos.reshape(-1, 3)
    da = rng.uniform(0, 30, len(pos))
    de = rng.uniform(0, 1, len(pos))
    for a, d in ((None, None), (da, None), (da, de), (None, de)):
        text, name = _orc_vlist(pos, a, d)
        assert reports.vlist_text(pos, a, d) == text
        assert name == "vlist_%06d" % len(pos)
    written = reports.write_vlist(str(tmp_path) + "/run_", pos, da)
    assert written.endswith("run_vlist_%06d" % len(pos))
    assert open(written).read() == _orc_vlist(pos, da)[0]


def test_progress_files_and_cost_lines(tmp_path):
    rng = np.random.default_rng(6)
    pos = rng.uniform(-1, 1, (25, 4))
    names = reports.write_rrt_progress(str(tmp_path) + "/p_", pos, 10)
    assert [n.rsplit("_", 1)[1] for n in names] == ["000010", "000020"]
    assert open(names[1]).read() == _orc_vlist(pos[:20])[0]
    lib = oracle_lib.load()
    lib.orc_format_cost_line.restype = C.c_int64
    rep = reports.LeastCostReport()
    rep.progress(10)
    rep.solution(12.3456789)
    rep.progress(20)
    rep.solution(15.0)
    rep.solution(1e-7)
    expect = []
    for n, c in ((10, 1e10), (20, 12.3456789)):
        buf = C.create_string_buffer(64)
        sz = lib.orc_format_cost_line(C.c_uint64(n), C.c_double(c), buf, C.c_uint64(64))
        expect.append(buf.raw[:sz].decode())
    assert rep.out == expect
    assert rep.sol == ["10 12.3457\n", "20 12.3457\n", "20 1e-07\n"]
    assert reports.timing_line(5000, 123456) == "5000 123456\n"
