"""The LOG_ADD coefficient-table offset of the fb kernel (muscle_amd/csrc/device_math.h: mpc_coef_offset) restated in
numpy and checked against the definition it replaces — the interval tests of LOGEXP1 (scoretype.h:100-109: d <= 1,
d <= 2.5, d <= 4.5, else) — on every interval bound, its float neighbours, denormals, zero, and a dense random sample.
Integer/bit arithmetic only, so numpy reproduces the device computation exactly."""
import numpy as np


def coef_offset(d):
    d = np.asarray(d, np.float32)
    u = (d.view(np.uint32).astype(np.uint64) + 0x027FFFFF).astype(np.uint32)
    p = u.view(np.float32).astype(np.float64)
    t = np.where(p >= 4294967296.0, 4294967295.0, np.floor(np.maximum(p, 0.0)))  # v_cvt_u32_f32: truncate, saturate
    return t.astype(np.uint64).astype(np.uint32) & np.uint32(0xF0)


def interval_of_entry(q):  # mpc_coef_table_init
    return 0 if q <= 1 else 1 if q <= 4 else 2 if q <= 8 else 3


def interval_reference(d):  # scoretype.h:104-108
    return np.where(d <= 1.0, 0, np.where(d <= 2.5, 1, np.where(d <= 4.5, 2, 3)))


def test_offsets_select_the_reference_interval():
    rng = np.random.default_rng(7)
    vals = [np.float32(0.0), np.float32(1e-45), np.float32(1e-39), np.float32(1.17549435e-38)]
    for k in range(0, 16):
        b = np.float32(k * 0.5)
        vals += [b, np.nextafter(b, np.float32(100)), np.nextafter(b, np.float32(-1))] if k else [b]
        for _ in range(3):
            vals.append(np.nextafter(vals[-1], np.float32(100)))
    d = np.concatenate([np.array(vals, np.float32), rng.uniform(0, 7.5, 2_000_000).astype(np.float32),
                        (rng.uniform(0, 1, 200_000) ** 8 * 7.5).astype(np.float32)])
    d = d[(d >= 0) & (d < 7.5)]  # d >= 7.5 returns hi: the table entry is not used
    off = coef_offset(d)
    assert np.all(off % 16 == 0) and np.all(off <= 0xF0)
    got = np.array([interval_of_entry(q) for q in range(16)])[off // 16]
    assert np.array_equal(got, interval_reference(d))


def test_offsets_stay_in_range_for_huge_d():
    d = np.array([7.5, 8.0, 100.0, 2e20, 4e20, 3.0e38], np.float32)  # LOG_ZERO operands: d up to 4e20
    assert np.all(coef_offset(d) <= 0xF0)
