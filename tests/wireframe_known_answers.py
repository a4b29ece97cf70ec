"""TEST INFRASTRUCTURE -- hand-derived known answers for wireframe rasterization (rasterize_kernel.cu:170-400).

The reference has no CPU twin and no test for `wireframe=True`, so these are the only pin that does not come from
reading the same CUDA source into code twice: each case is small enough to be worked out on paper from the RULES the
source states, and every coordinate is a small dyadic number, so that every product, sum and quotient the kernel
forms (line coefficients `get_line` :171-180, Cramer's rule `get_cross_point` :193-218) is exact in float32 -- the
answer does not depend on rounding, on the evaluation order or on the compiler.

The rules (pixel (x, y) = its centre; a triangle's edges are e0 = p0p1, e1 = p1p2, e2 = p0p2, :350-352):
  R1  a pixel is an EDGE pixel of a visible edge iff the closed edge segment meets one of the four closed sides of the
      pixel's diamond |dx| + |dy| = 1/2 in a point, the side not being parallel to the edge (:220-262; parallel
      lines have no cross point :198-200; `is_point_in_segment` is inclusive :182-190 -- there is NO diamond-exit
      refinement: touching a diamond corner counts, for both pixels that share the corner)
  R2  edge k is visible iff bit k of the top nibble of vi[...,0] is set; bit 3 is ignored (:293-303)
  R3  edge pixels write (depth, id); pixels inside the triangle by the top-left rule that are not edge pixels of
      one of its visible edges write (depth, 0xFFFFFFFF), i.e. depth with index -1: the surface occludes what is
      behind it without being drawn (:354-389); 64-bit atomicMin => nearer wins, at equal depth the drawn id wins
  R4  the loop runs over x in [max(1, int(min_x) - 2), min(W - 2, int(max_x) + 2)] and likewise y (:320-325): the
      one-pixel canvas border is never written
  R5  culls as in triangle mode: all-equal indices, any z <= 1e-8, bbox off canvas, zero area (:305-317)

Each case: name, v [V,3], triangles [(i0, i1, i2, flag)], H, W, expected {id: set of (x, y)} for drawn pixels,
expected set of filled-but-not-drawn pixels (index -1, depth > 0) or None when the case does not state it."""
import torch as th


def _tri(v, tris, H, W):
    vv = th.tensor(v, dtype=th.float64)[None]
    vi = th.tensor([[a | (f << 28), b, c] for a, b, c, f in tris], dtype=th.int64).to(th.int32)
    return vv, vi, H, W


def row(y, x0, x1):
    return {(x, y) for x in range(x0, x1 + 1)}


def col(x, y0, y1):
    return {(x, y) for y in range(y0, y1 + 1)}


CASES = []


def case(name, v, tris, H, W, drawn, filled=None, why=""):
    CASES.append(dict(name=name, scene=_tri(v, tris, H, W), drawn=drawn, filled=filled, why=why))


# 1. horizontal edge THROUGH pixel centres, y = 5, x in [2, 8].  The line meets the diamond of (x, 5) in its left and
#    right corners (x -+ 1/2, 5); one of them lies in [2, 8] for x = 2..8 (x = 2: 2.5; x = 8: 7.5; x = 1: 1.5 and 0.5
#    are outside; x = 9: 8.5, 9.5 outside).  Rows 4 and 6: their diamonds reach y = 4.5 / 5.5 only.
case("horizontal_through_centres", [[2, 5, 2], [8, 5, 2], [5, 9, 2]], [(0, 1, 2, 1)], 12, 12, {0: row(5, 2, 8)})
# 2. the same edge a quarter pixel lower, y = 5.25: crosses the diamond of (x, 5) at (x -+ 1/4, 5.25); x -+ 1/4 in
#    [2, 8] for x = 2..8.  |dy| = 3/4 for row 6: outside its diamonds.
case("horizontal_quarter_offset", [[2, 5.25, 2], [8, 5.25, 2], [5, 9, 2]], [(0, 1, 2, 1)], 12, 12, {0: row(5, 2, 8)})
# 3. ... and exactly between two rows, y = 5.5: it touches the bottom corner (x, 5.5) of the diamonds of row 5 AND the
#    top corner of those of row 6; corners count (R1), so both rows are drawn for x = 2..8 (the corner itself must
#    lie in [2, 8]).  This is the case that shows there is no diamond-EXIT rule in this code.
case("horizontal_between_rows_touches_both", [[2, 5.5, 2], [8, 5.5, 2], [5, 9, 2]], [(0, 1, 2, 1)], 12, 12,
     {0: row(5, 2, 8) | row(6, 2, 8)})
# 4. vertical edge through centres, x = 5, y in [2, 8]: the transpose of case 1.
case("vertical_through_centres", [[5, 2, 2], [5, 8, 2], [9, 5, 2]], [(0, 1, 2, 1)], 12, 12, {0: col(5, 2, 8)})
# 5. 45-degree edge through centres, (2,2) -> (8,8).  It is parallel to two sides of every diamond (no cross point)
#    and crosses the other two sides of the diamond of (k, k) at (k -+ 1/4, k -+ 1/4); in the segment for k = 2..8.
#    For a pixel off the diagonal, (k+1, k): on y = x, |x - k - 1| + |x - k| >= 1 > 1/2 -- never inside a diamond.
case("diagonal_through_centres", [[2, 2, 2], [8, 8, 2], [2, 8, 2]], [(0, 1, 2, 1)], 12, 12, {0: {(k, k) for k in range(2, 9)}})
# 6. anti-diagonal through centres, (9,2) -> (2,9), as edge e1 (bit 1) of a right triangle.
case("antidiagonal_as_edge1", [[2, 2, 2], [9, 2, 2], [2, 9, 2]], [(0, 1, 2, 2)], 14, 14, {0: {(k, 11 - k) for k in range(2, 10)}})
# 7. slope-2 edge (3,2) -> (6,8), x = 3 + (y - 2) / 2.  Rows with an integer x (y even) cross one diamond through its
#    centre; rows with a half-integer x (y odd) pass exactly through the shared corner (x, y) of two diamonds, which
#    counts for both.  Nothing else: e.g. for (3, 4), |x - 3| + |y - 4| on the line is >= 1.
case("slope2_hits_shared_corners", [[3, 2, 2], [6, 8, 2], [9, 2, 2]], [(0, 1, 2, 1)], 12, 12,
     {0: {(3, 2), (3, 3), (4, 3), (4, 4), (4, 5), (5, 5), (5, 6), (5, 7), (6, 7), (6, 8)}})
# 8-15. the visibility nibble (R2) on the right triangle (2,2) (9,2) (2,9): e0 = row 2, x = 2..9 (case 1's argument);
#    e1 = the anti-diagonal of case 6; e2 = column 2, y = 2..9.  flag f draws the union of the edges whose bit is
#    set; bit 3 draws nothing.  The triangle's own interior is FILLED (R3): denominator = 49 > 0, e0 (v01 = (7,0))
#    and e2 (v02 = (0,7)) are top/left edges, e1 (v12 = (-7,7)) is not, so the inside set is x >= 2, y >= 2,
#    x + y <= 10 -- 28 pixels; whichever of them are not drawn are filled.  Pixels on x + y = 11 are not inside
#    (non-top-left edge), so they carry depth only when e1 is drawn.
_E0, _E1, _E2 = row(2, 2, 9), {(k, 11 - k) for k in range(2, 10)}, col(2, 2, 9)
_INSIDE = {(x, y) for x in range(2, 9) for y in range(2, 9) if x + y <= 10}
for f in range(16):
    drawn = (_E0 if f & 1 else set()) | (_E1 if f & 2 else set()) | (_E2 if f & 4 else set())
    case(f"nibble_{f:04b}", [[2, 2, 2], [9, 2, 2], [2, 9, 2]], [(0, 1, 2, f)], 14, 14, {0: drawn}, filled=_INSIDE - drawn)
# 16. occluding fill (R3): a far wire triangle (z = 4, all edges, id 0) behind a near UNDRAWN triangle (z = 2, flag 0,
#    id 1) with corners (2,2) (6,2) (2,6): by the inside rule above (top/left edges e0, e2; the hypotenuse x + y = 8 is
#    not one) it covers x >= 2, y >= 2, x + y <= 7 -- those pixels hold the near depth with index -1, so the far
#    triangle's edge pixels among them vanish: of its row 4 (x = 1..10) the pixels x = 2, 3 are hidden (x + 4 <= 7;
#    (4,4) lies ON the near hypotenuse, is not inside, and stays drawn).  Far triangle: (1,4) (10,4) (5,10):
#    e0 = row 4, x = 1..10.
_NEAR = {(x, y) for x in range(2, 7) for y in range(2, 7) if x + y <= 7}
case("near_fill_hides_far_wire", [[1, 4, 4], [10, 4, 4], [5, 10, 4], [2, 2, 2], [6, 2, 2], [2, 6, 2]],
     [(0, 1, 2, 1), (3, 4, 5, 0)], 14, 14, {0: row(4, 1, 10) - _NEAR, 1: set()}, filled=None)
# 17. the converse: the near triangle drawn (flag 7), the far one undrawn: the near wire is unaffected by the far fill.
case("near_wire_over_far_fill", [[1, 4, 4], [10, 4, 4], [5, 10, 4], [2, 2, 2], [6, 2, 2], [2, 6, 2]],
     [(0, 1, 2, 0), (3, 4, 5, 7)], 14, 14, {0: set(), 1: row(2, 2, 6) | col(2, 2, 6) | {(k, 8 - k) for k in range(2, 7)}})
# 18. equal depth, drawn beats filled (R3): the SAME triangle twice, id 0 undrawn, id 1 drawn -- identical arithmetic,
#    identical depth bits, and (depth, 1) < (depth, 0xFFFFFFFF) on every edge pixel.
case("drawn_beats_fill_at_equal_depth", [[2, 2, 2], [9, 2, 2], [2, 9, 2]], [(0, 1, 2, 0), (0, 1, 2, 7)], 14, 14,
     {0: set(), 1: _E0 | _E1 | _E2}, filled=_INSIDE - (_E0 | _E2))
# 19. the canvas border is never written (R4): a triangle larger than the 8x8 canvas with every edge drawn; its
#    edges x = 0 and y = 0 pass through border pixels only, its hypotenuse x + y = 9 through (2,7) .. (7,2) --
#    of which (7,2) and (2,7) lie on the border column / row 7.
case("border_never_written", [[0, 0, 2], [9, 0, 2], [0, 9, 2]], [(0, 1, 2, 7)], 8, 8,
     {0: {(k, 9 - k) for k in range(3, 7)}}, filled={(x, y) for x in range(1, 7) for y in range(1, 7) if x + y <= 8})
# 20. culls (R5): (a) all-equal indices, (b) a vertex at z = 1e-9 <= 1e-8, (c) zero area with every edge "visible",
#    (d) bbox entirely right of the canvas.  Nothing is written at all.
case("culled_triangles", [[2, 2, 2], [9, 2, 2], [2, 9, 2], [5, 5, 1e-9], [3, 3, 2], [6, 6, 2], [9, 9, 2], [20, 3, 2], [25, 3, 2], [20, 8, 2]],
     [(0, 0, 0, 7), (0, 1, 3, 7), (4, 5, 6, 7), (7, 8, 9, 7)], 12, 12, {0: set(), 1: set(), 2: set(), 3: set()}, filled=set())


def check(rasterize_lines, dtype=th.float32):
    """`rasterize_lines(v [1,V,3], vi [F,3] int32, H, W) -> (depth_img [1,H,W] f32, index_img [1,H,W] i32)` on CPU tensors."""
    for c in CASES:
        v, vi, H, W = c["scene"]
        depth, index = rasterize_lines(v.to(dtype), vi, H, W)
        depth, index = depth[0], index[0]
        for tid, want in c["drawn"].items():
            got = {(int(x), int(y)) for y, x in (index == tid).nonzero().tolist()}
            assert got == want, f"{c['name']}: triangle {tid} drawn at {sorted(got ^ want)} differently (got - want, want - got = {sorted(got - want)}, {sorted(want - got)})"
        ids = set(index.unique().tolist()) - {-1}
        assert ids <= set(c["drawn"]), (c["name"], ids)
        assert (depth[index >= 0] > 0).all(), c["name"]
        if c["filled"] is not None:
            got = {(int(x), int(y)) for y, x in ((index == -1) & (depth > 0)).nonzero().tolist()}
            assert got == c["filled"], f"{c['name']}: filled set differs by {sorted(got ^ c['filled'])}"
            # nothing else is touched: depth 0 <=> neither drawn nor filled
            assert int((depth > 0).sum()) == len(c["filled"]) + sum(len(s) for s in c["drawn"].values()), c["name"]
        # every vertex of these cases is at z = 2 or z = 4: written depths are one of the two (to a few ulp)
        d = depth[depth > 0].double()
        assert bool((((d - 2).abs() < 1e-5) | ((d - 4).abs() < 1e-5)).all()), c["name"]
        # R4
        assert int((depth[0] > 0).sum() + (depth[-1] > 0).sum() + (depth[:, 0] > 0).sum() + (depth[:, -1] > 0).sum()) == 0, c["name"]


# ---- the same rules in exact rational arithmetic (an independent evaluation, not a third reading of the source's
# ---- floating-point code: no get_line / Cramer, no rounding) -------------------------------------------------------
def exact_model(v, vi, H, W):
    """(index image, written mask) [H][W] for ONE view by rules R1-R5 in fractions.Fraction.  Coordinates must be
    dyadic (they are converted exactly); every triangle must have one constant z, distinct from the others', so that
    depth order is decided by z alone."""
    from fractions import Fraction as Fr

    def seg_hits(p, q, a, b):
        """closed segments pq and ab, non-parallel: do their LINES cross in a point that lies in both bounding boxes
        (for a point on both lines that is 'on both segments')?"""
        d1, d2 = (q[0] - p[0], q[1] - p[1]), (b[0] - a[0], b[1] - a[1])
        den = d1[0] * d2[1] - d1[1] * d2[0]
        if den == 0:
            return False
        t = ((a[0] - p[0]) * d2[1] - (a[1] - p[1]) * d2[0]) / den
        c = (p[0] + t * d1[0], p[1] + t * d1[1])
        inside = lambda s0, s1: min(s0[0], s1[0]) <= c[0] <= max(s0[0], s1[0]) and min(s0[1], s1[1]) <= c[1] <= max(s0[1], s1[1])  # noqa: E731
        return inside(p, q) and inside(a, b)

    h = Fr(1, 2)
    best = [[None] * W for _ in range(H)]
    for tid, (i0, i1, i2) in enumerate(vi.tolist()):
        flag, i0 = (i0 >> 28) & 0xF, i0 & 0x0FFFFFFF
        if i0 == i1 == i2:
            continue
        P = [tuple(Fr(float(c)) for c in v[i]) for i in (i0, i1, i2)]
        z = P[0][2]
        assert P[1][2] == z and P[2][2] == z
        if not all(p[2] > Fr(1, 10**8) for p in P):
            continue
        xs, ys = [p[0] for p in P], [p[1] for p in P]
        if not (min(xs) <= W - 1 and min(ys) <= H - 1 and max(xs) > 0 and max(ys) > 0):
            continue
        den = (P[1][0] - P[0][0]) * (P[2][1] - P[0][1]) - (P[1][1] - P[0][1]) * (P[2][0] - P[0][0])
        if den == 0:
            continue
        sg = 1 if den > 0 else -1
        # signed area of (a, b, p), positive on the triangle's side once multiplied by sg (rasterize_kernel.cu:19-27)
        e = lambda a, b, p: ((p[1] - a[1]) * (b[0] - a[0]) - (p[0] - a[0]) * (b[1] - a[1])) * sg  # noqa: E731
        edges = [(P[1], P[2]), (P[2], P[0]), (P[0], P[1])]  # barycentric k belongs to the edge opposite vertex k

        def top_or_left(a, b):
            """edge a -> b taken in the triangle's positive sense: going up the image (dy < 0), or level and going right"""
            dx, dy = (b[0] - a[0]) * sg, (b[1] - a[1]) * sg
            return dy < 0 or (dy == 0 and dx > 0)

        tl = [top_or_left(a, b) for a, b in edges]  # :369-377 (its v_02 is edge 1 reversed, hence the flipped signs there)
        vis = [(P[0], P[1], flag & 1), (P[1], P[2], flag & 2), (P[0], P[2], flag & 4)]
        for y in range(max(1, int(min(ys)) - 2), min(H - 2, int(max(ys)) + 2) + 1):
            for x in range(max(1, int(min(xs)) - 2), min(W - 2, int(max(xs)) + 2) + 1):
                c = (Fr(x), Fr(y))
                corners = [(c[0], c[1] - h), (c[0] + h, c[1]), (c[0], c[1] + h), (c[0] - h, c[1])]
                hit = any(f and any(seg_hits(a, b, corners[k], corners[(k + 1) % 4]) for k in range(4)) for a, b, f in vis)
                b = [e(a2, b2, c) for a2, b2 in edges]
                inside = all(t >= 0 for t in b) and not any(b[k] == 0 and not tl[k] for k in range(3))
                if hit or inside:
                    key = (z, tid if hit else 0xFFFFFFFF)
                    if best[y][x] is None or key < best[y][x]:
                        best[y][x] = key
    return [[-1 if k is None or k[1] == 0xFFFFFFFF else k[1] for k in r] for r in best], [[k is not None for k in r] for r in best]


def random_dyadic_scene(seed, H=20, W=24, ntri=6):
    """Triangles with vertices on the quarter-pixel grid (a third of them on pixel centres, a third on diamond
    corners), random nibbles, one distinct dyadic z per triangle."""
    g = th.Generator().manual_seed(seed)
    xy = th.rand(ntri, 3, 2, generator=g, dtype=th.float64) * th.tensor([W + 4.0, H + 4.0], dtype=th.float64) - 2.0
    xy = (xy * 4).round() / 4
    xy[::3] = xy[::3].round()
    xy[1::3] = (xy[1::3] * 2).round() / 2
    z = (1 + th.randperm(ntri, generator=g).double() * 0.25)[:, None, None].expand(ntri, 3, 1)
    v = th.cat([xy, z], -1).reshape(1, ntri * 3, 3).contiguous()
    vi = th.arange(ntri * 3, dtype=th.int32).view(ntri, 3).clone()
    vi[:, 0] |= th.randint(0, 16, (ntri,), generator=g, dtype=th.int32) << 28
    return v, vi, H, W


def check_against_exact_model(rasterize_lines, seeds, dtype=th.float32):
    for seed in seeds:
        v, vi, H, W = random_dyadic_scene(seed)
        depth, index = rasterize_lines(v.to(dtype), vi, H, W)
        want_i, want_w = exact_model(v[0].tolist(), vi, H, W)
        assert index[0].tolist() == want_i, f"seed {seed}: index image differs from the exact-arithmetic model"
        assert (depth[0] > 0).tolist() == want_w, f"seed {seed}: written-pixel set differs from the exact-arithmetic model"
