"""CPU: csrc/bev_exact.h evaluates the reference's mixed float/double expressions with float-only
forms (fewer, cheaper device instructions).  Each form is compared with the literal expression for
ALL 2^32 float inputs."""
import ctypes as C

import hostcheck_lib as hc


def test_float_only_forms_equal_the_literal_expressions_for_every_float():
    out = (C.c_uint64 * 7)()
    hc.lib().hc_exhaustive_exact_forms(out)
    names = ["x + 75.0f / y + 50.0f", "floor(n / 2.0)", "round(v + 0.5) bin", "(z + 2) * 4 height", "d > 0.30",
             "bin_in_range (six image sizes)", "bin_of_shifted (seven range / interval pairs)"]
    assert {n: int(v) for n, v in zip(names, out)} == {n: 0 for n in names}


def test_exact_reciprocal_is_a_division():
    """bev_exact.h replaces `/ interval` and `/ HEIGHT_RES` (BatchMultiBevGen.cpp:279-281) by a multiplication when the
    divisor is a power of two: the same correctly rounded operation for every dividend, subnormal and overflowing
    results included; any other divisor keeps the division."""
    import hostcheck_lib as hc
    assert hc.lib().hc_exact_reciprocal_check(200000) == 0


def test_angle_predicate_without_the_division():
    """bev_exact.h angle_is_ground_nodiv (what the column walk evaluates: a double multiply and compare against the
    midpoint between the threshold and its successor) against the division form, BatchMultiBevGen.cpp:173-179: random
    bit patterns, the walk's magnitudes, pairs straddling the cut by up to four ulps over sixty binades, special values."""
    assert hc.lib().hc_angle_nodiv_check(40_000_000) == 0


def test_count_of_a_cell_in_one_step():
    """bev_exact.h count_advance: n of the reference's `cnt = cnt + 1` float steps (BatchMultiBevGen.cpp:205-206, from
    0.01f, :135-136) at once — one exact addition per binade plus the rounding steps — against the step-by-step loop,
    for every n up to 2^21 (more than the slots of the largest range image) and for chains of random run lengths."""
    assert hc.lib().hc_count_advance_check(1 << 21) == 0


def test_raster_band_of_an_x_bin_without_a_division():
    """bev_exact.h small_div / raster_band_of_nodiv (every workgroup of the walk and of phase C fills a table of M bands:
    two multiplications instead of three integer divisions per entry) against the divisions: every 0 <= x < 512 and
    1 <= d <= 512, and every image size and band layout fill_geometry can produce."""
    assert hc.lib().hc_small_div_check() == 0
