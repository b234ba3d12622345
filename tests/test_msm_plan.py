"""The merged-window MSM plan (csrc/msm.hip: make_plan_merged) on the CPU: the top window of a scalar holds only its leading
bits and its digit is scaled by 2^top_shift against a table entry built with that many fewer doublings.  The recoding relies
on the scaled digit never exceeding 2^(c-1) (it is never recoded to a negative digit, so no carry leaves the top window)."""
import ctypes

from zkevm_circuits_amd import binding

R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def test_top_window_shift_keeps_the_digit_in_range():
    lib = binding.lib()
    seen = set()
    for k in range(0, 29):
        c, w, sh = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert lib.zk_host_msm_plan(ctypes.c_uint32(k), ctypes.byref(c), ctypes.byref(w), ctypes.byref(sh)) == 0
        c, w, sh = c.value, w.value, sh.value
        assert 8 <= c <= 22 and c == max(8, min(k, 22))
        assert w == (256 + c - 1) // c and w * c >= 254
        low = c * (w - 1)
        top_max = ((R_MOD - 1) >> low) + 1                       # largest top digit of a canonical scalar, plus the carry of the window below
        assert (top_max << sh) <= 1 << (c - 1), (k, c, w, sh)
        assert sh == c - 1 or (top_max << (sh + 1)) > 1 << (c - 1), "the shift is the largest one that fits"
        seen.add((c, w, sh))
    assert (20, 13, 5) in seen                                   # the 2^20 plan of the bench: 13 windows, top digit x 32
    assert lib.zk_host_msm_plan(ctypes.c_uint32(29), None, None, None) != 0
