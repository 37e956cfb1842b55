"""Checks behind two claims the kernel source makes without the compiler being able to prove them.

(a) csrc/salp_device.h step_head(): `p = timer / duration` of the breathing cycle (legacy:212-213, :240-246) is computed as
    y = RN(1/den), q0 = RN(t y), p = RN(q0 + RN_fma(t - den q0) y) instead of the IEEE division sequence, and is claimed
    to equal the correctly rounded quotient for every duration 1..255 and timer 1..257.  Checked here in exact rationals.
(b) csrc/salp_vec.hip: the observation tile (96-B rows, float4 column XOR bit 2 of the row) is claimed conflict-free for
    both the row writes (ds_write_b128) and the flush reads (ds_read_b128) under the banking table of
    MI355X_MICROARCH.md §LDS; profiles/isa_lds_model.py is that table as code.
"""
import importlib.util
import os
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rn(x: Fraction) -> float:
    """Round-to-nearest-even of an exact rational to binary64 (CPython's int / int is correctly rounded)."""
    return x.numerator / x.denominator


def test_reciprocal_fma_quotient_is_the_ieee_quotient():
    bad = []
    for den in range(1, 256):
        y = 1.0 / den                                    # RN(1 / den): IEEE division
        fy, fden = Fraction(y), Fraction(den)
        for t in range(1, 258):
            ft = Fraction(t)
            q0 = rn(ft * fy)                              # v_mul_f64
            r = rn(ft - fden * Fraction(q0))              # v_fma_f64(-den, q0, t)
            p = rn(Fraction(q0) + Fraction(r) * fy)       # v_fma_f64(r, y, q0)
            if p != t / den:
                bad.append((den, t, p, t / den))
    assert not bad, bad[:5]


def test_the_thrust_window_of_forced_breathing_is_an_integer_test():
    """0.1 <= t/150 <= 0.5 exactly for t = 15..75 (61 steps), evaluated on the rounded quotient."""
    on = [t for t in range(1, 152) if 0.1 <= t / 150 <= 0.5]
    assert on == list(range(15, 76))


def test_swizzled_observation_tile_is_conflict_free():
    spec = importlib.util.spec_from_file_location("isa_lds_model", os.path.join(ROOT, "profiles", "isa_lds_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    w, rd = m.model(m.layouts[m.SHIPPED])
    assert (w, rd) == (8 * m.Q, 4 * m.Q)                 # one LDS cycle per lane group: no conflict anywhere
    w1, rd1 = m.model(m.layouts['112-B padded pitch (round 1)'])
    assert w1 == 8 * m.Q and rd1 == 2 * 4 * m.Q          # round 1: writes clean, flush reads 2-way (63.5 M conflict cycles measured)
    # the kernel's own formulas (salp_vec.hip: lds_off, myrow_even / myrow_odd) are this layout
    f = m.layouts[m.SHIPPED]
    for lane in range(64):
        swz = (lane >> 2) & 1
        for q in range(6):
            mine = lane * 96 + 16 * ((q - swz) if (q & 1) else (q + swz))
            assert mine == f(lane, q)


def test_ellipse_a_is_the_larger_axis_for_the_reference_radii():
    """csrc/salp_device.h step_head(): the literal-constant kernels take r = max(ellipse_a, ellipse_b) = ellipse_a
    (legacy:190-246: a = 1.3R .. 1.1R .. 1.3R, b = 0.8R .. 1.1R .. 0.8R).  Every (phase, timer, duration) the state machine
    can reach, in the kernel's own fp64 expressions (R = 30: a_rest 39, b_rest 24, full 33, slopes -6 / +9 / +6 / -9)."""
    for t in range(0, 121):                       # inhaling, and the unchanged ellipse at release (water = t / 120)
        p = t / 120
        a, b = 39.0 + (-6.0) * p, 24.0 + 9.0 * p
        assert a >= b, (t, a, b)
    for dur in range(1, 256):                     # exhaling with any duration the packed word can hold
        for t in range(1, dur + 1):
            p = t / dur
            a, b = 33.0 + 6.0 * p, 33.0 + (-9.0) * p
            assert a >= b, (dur, t, a, b)
    assert 33.0 + 6.0 * 1.0 >= 33.0 + (-9.0) * 1.0 and 39.0 >= 24.0     # last exhale step, rest


def test_fp32_food_key_error_stays_inside_the_tie_tolerance():
    """csrc/salp_food_reg.h: the fp32 ordering pass may decide the order of two foods only when their keys are farther
    apart than `tie_tolerance(d2) = 1.4e-7 L^2 + 8e-6 d2`; the claim behind it is that a key (fp32 squared distance from
    fp32-rounded positions, low 4 bits replaced by the slot) is within HALF of that of the exact squared distance.
    Emulated here with numpy float32 arithmetic for random geometry in tanks of several sizes; the GPU-side check of the
    same property is tests/test_gpu_parity.py::test_near_tie_food_order_matches_oracle_across_scales."""
    import numpy as np
    rng = np.random.default_rng(0)
    for L, H in ((800, 600), (1200, 900), (500, 450), (4000, 3000)):
        n = 400000
        x, y = rng.uniform(0, L, n), rng.uniform(0, H, n)
        # foods at every distance scale from 1 px to the tank's diagonal
        r = 10.0 ** rng.uniform(0, np.log10(np.hypot(L, H)), n)
        a = rng.uniform(0, 2 * np.pi, n)
        fx, fy = x + r * np.cos(a), y + r * np.sin(a)
        ok = (fx >= 0) & (fx <= L) & (fy >= 0) & (fy <= H)
        x, y, fx, fy = x[ok], y[ok], fx[ok], fy[ok]
        exact = (fx - x) ** 2 + (fy - y) ** 2
        dx = (fx.astype(np.float32) - x.astype(np.float32)).astype(np.float32)
        dy = (fy.astype(np.float32) - y.astype(np.float32)).astype(np.float32)
        d2 = (dy.astype(np.float64) ** 2 + dx.astype(np.float64) ** 2).astype(np.float32)      # fma(dy, dy, dx dx): one rounding
        for slot in (0, 15):
            key = ((d2.view(np.uint32) & np.uint32(0x7FFFFFF0)) | np.uint32(slot)).view(np.float32).astype(np.float64)
            err = np.abs(key - exact)
            tol = 1.4e-7 * max(L, H) ** 2 + 8e-6 * exact
            worst = float((err / tol).max())
            assert worst <= 0.5, (L, slot, worst)
