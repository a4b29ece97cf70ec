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
