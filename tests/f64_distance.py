"""TEST INFRASTRUCTURE -- the one bound every float32 gradient check of this repository can be reduced to.

A float32 result is a sum of rounded terms in SOME order; two correct float32 evaluations of the same sum (the
reference's loop, this repository's kernel, the reference's own CUDA kernel) differ from each other by their two
rounding errors, which grow with what was summed, not with the result.  Comparing the kernel with the float32 oracle
under a flat 1e-5 therefore fails now and then for reasons that are nobody's defect (round 3 loosened three such
checks, each after a fuzzer went red, each justified by hand with exactly the comparison below).  The comparison that
means something is against the same computation carried in DOUBLE:

    |kernel_f32 - oracle_f64|  <=  max( rel * max|oracle_f64| ,  k * |oracle_f32 - oracle_f64| ,  ulps * u * A )

  rel = 1e-5   the north-star bar (BASELINE.json), on the magnitude of the output;
  k   = 3      "as near to the exact result as the reference's own float32 evaluation, up to a factor": both errors are
               maxima over the output of rounding noise of the same size, so their ratio concentrates near 1 as soon as
               the output has more than a handful of elements;
  A            (optional) the magnitude that was ACCUMULATED into the worst element -- the same operator applied to
               absolute values -- for outputs with so few elements that the maximum of the oracle's own error is one
               lucky draw (the 1 x 1 level of a mip pyramid collecting every tap of every pixel: three numbers);
               u = 2^-24, ulps = 8.
Used by tests/test_gpu_f64_distance.py (every backward operator and sampler mode) and by the fuzzers
(fuzz_mipmap_snapped.py, fuzz_next_ops.py) in place of their hand-tuned bounds."""
import torch as th

U32 = 2.0 ** -24


def f64_distance_bound(oracle_f32, oracle_f64, k=3.0, rel=1e-5, acc_magnitude=None, acc_ulps=8.0):
    o32, o64 = oracle_f32.detach().cpu().double(), oracle_f64.detach().cpu().double()
    assert o32.shape == o64.shape
    if o64.numel() == 0:
        return 0.0, 0.0
    own = float((o32 - o64).abs().max())
    bound = max(rel * float(o64.abs().max()), k * own)
    if acc_magnitude is not None:
        bound = max(bound, acc_ulps * U32 * float(acc_magnitude))
    return bound, own


def assert_within_f64_distance(got_f32, oracle_f32, oracle_f64, what, k=3.0, rel=1e-5, acc_magnitude=None, acc_ulps=8.0):
    """got_f32: the kernel's float32 output; oracle_f32 / oracle_f64: the CPU oracle on the same inputs in float32 / with
    every floating-point input cast to double.  Returns (error of the kernel, error of the float32 oracle), both against
    the double result, for reporting."""
    got = got_f32.detach().cpu().double()
    o64 = oracle_f64.detach().cpu().double()
    assert got.shape == o64.shape, (what, got.shape, o64.shape)
    assert bool(th.isfinite(got).all()) or not bool(th.isfinite(o64).all()), f"{what}: non-finite output"
    bound, own = f64_distance_bound(oracle_f32, oracle_f64, k, rel, acc_magnitude, acc_ulps)
    err = float((got - o64).abs().max()) if got.numel() else 0.0
    assert err <= bound, (f"{what}: |kernel - f64| = {err:.3e} > {bound:.3e} = max({rel:g} * max|f64| = {rel * float(o64.abs().max()) if o64.numel() else 0:.3e}, "
                          f"{k:g} * |oracle_f32 - f64| = {k * own:.3e}" + (f", {acc_ulps:g} ulp of the accumulated {float(acc_magnitude):.3e}" if acc_magnitude is not None else "") + ")")
    return err, own


# ---- per element (round 6) -------------------------------------------------------------------------------------------------
# The bound above is a MAX-NORM bound: with max|ref| ~ 40 it lets a vertex whose whole gradient is 1e-4 be 100 % wrong.
# What a float32 sum can be held to per element is the magnitude that was accumulated INTO THAT ELEMENT,
#     A_i = sum_j |term_ij|         (oracle.accumulated_magnitudes(): the same operator summing absolute values),
# because that -- not the result, which may have cancelled -- is what the rounding errors of the terms and of their sum
# scale with.  Element i passes iff
#     |kernel_i - f64_i|  <=  max( k * |oracle_f32_i - f64_i| ,  ulps * u * A_i ,  floor_rel * max|f64| )
# (floor_rel: what is not resolved per element -- the tests use 1e-7, the north star's 1e-5 of the output's scale tightened 100 x)
# `ulps` is per operator (the terms of render backward are themselves differences of rounded quantities: b0 = 1 - b1 - b2,
# -dL_b0 + dL_b1 ... -- the float32 ORACLE is up to ~300 u A_i from the double result there; interpolate's terms are single
# products: 3-4 u A_i); the values used by the tests are stated there, each a small multiple of what the float32 oracle
# itself needs.  An element nothing was accumulated into (A_i = 0) must be exactly what the oracle has (0).
def elementwise_excess(got_f32, oracle_f32, oracle_f64, magnitudes, ulps, k=3.0, floor_rel=0.0):
    """max_i |got_i - f64_i| / bound_i  (<= 1 passes), the index of the worst element, and the same ratio for the float32
    oracle against ulps * u * A_i alone (how much of the allowance the reference's own arithmetic uses)."""
    got, o32, o64 = (t.detach().cpu().double() for t in (got_f32, oracle_f32, oracle_f64))
    A = magnitudes.detach().cpu().double()
    assert got.shape == o64.shape == o32.shape == A.shape
    if o64.numel() == 0:
        return 0.0, -1, 0.0
    err = (got - o64).abs()
    own = (o32 - o64).abs()
    bound = th.maximum(k * own, ulps * U32 * A).clamp(min=floor_rel * float(o64.abs().max()))  # floor_rel: what is not resolved per element
    dead = bound == 0
    ratio = th.where(dead, th.where(err == 0, th.zeros_like(err), th.full_like(err, float("inf"))), err / bound.clamp(min=1e-300))
    worst = int(ratio.flatten().argmax())
    live = A > 0
    own_ratio = float((own[live] / (ulps * U32 * A[live])).max()) if bool(live.any()) else 0.0
    return float(ratio.flatten()[worst]), worst, own_ratio


def assert_elementwise_within(got_f32, oracle_f32, oracle_f64, magnitudes, ulps, what, k=3.0, floor_rel=0.0):
    assert bool(th.isfinite(got_f32).all()) or not bool(th.isfinite(oracle_f64).all()), f"{what}: non-finite output"
    excess, worst, own_ratio = elementwise_excess(got_f32, oracle_f32, oracle_f64, magnitudes, ulps, k, floor_rel)
    if excess > 1.0:
        g, o32, o64, A = (t.detach().cpu().double().flatten()[worst] for t in (got_f32, oracle_f32, oracle_f64, magnitudes))
        raise AssertionError(f"{what}: element {worst}: kernel {float(g):.9e}, f64 {float(o64):.9e}, oracle f32 {float(o32):.9e}, accumulated magnitude "
                             f"{float(A):.3e}: |kernel - f64| = {abs(float(g - o64)):.3e} is {excess:.2f} x max({k:g} |oracle_f32 - f64|, {ulps:g} u A, {floor_rel:g} max|f64|)")
    return excess, own_ratio
