// s2p_amd/csrc/mgm_geom.hpp -- lattice decomposition of the MGM recursion (host + device, no HIP types).
//
// MGM (oracle/census_oracle.c, recursion = 1) gives every direction r = (dx, dy) two predecessors per pixel p:
// p - r and p - r_perp with r_perp = (-dy, dx).  For each of the 8 directions the dependency graph is a union of
// "quadrant" lattices: integer coordinates (u, v) in [0, U) x [0, V) whose point (u, v) depends on (u - 1, v) and
// (u, v - 1) only, with an affine map (u, v) -> pixel:
//   * the 4 axis directions are ONE lattice each, the image itself up to flips (U = w, V = h);
//   * a diagonal direction (after a flip / transpose that makes its predecessors (X - 1, Y - 1) and (X + 1, Y - 1))
//     splits into the TWO parity classes of a = X + Y: with b = Y - X + WX - 1 the predecessors are (a - 2, b) and
//     (a, b - 2), so u = a >> 1, v = b >> 1 inside a class; the image is a rotated rectangle inside that lattice and
//     the points outside it are "not in the image": they send no message, exactly like a predecessor beyond the border.
// 12 lattices in total; every (direction, pixel) pair is a point of exactly one of them (tests/test_mgm_geom.py
// compiles tools/probes/mgm_geom_check.cpp against this header and checks the cover, the predecessor maps and the
// row intervals on the host).
//
// 16 directions (cfg['mgm_nb_directions'] = 16, round 4): the 8 knight's moves r = (+-2, +-1), (+-1, +-2) come on top, directions
// 8..15 of the table below, with the same rule r_perp = (-dy, dx).  The vectors r and r_perp span a sublattice of index
// dx^2 + dy^2 = 5 of the pixel grid, so a knight direction splits into the FIVE residue classes of a = dx x + dy y mod 5 (inside a
// class b = -dy x + dx y is fixed mod 5 too): u = (a - a0) / 5, v = (b - b0) / 5, pixel = origin + u r + v r_perp -- the image is a
// rectangle rotated by atan(1/2) inside the lattice.  40 more lattices, numbered 12..51: 52 in all.
#pragma once

#if defined(__HIPCC__)
#define S2P_HD __host__ __device__ __forceinline__
#else
#define S2P_HD inline
#endif

namespace s2p {

struct MgmLattice {
    int r;                       // direction index (same table as the path kernel: 0..3 axis, 4..7 diagonal; 8..15 knight's moves)
    int U, V;                    // lattice extent (<= 0: empty)
    int x0, xu, xv, y0, yu, yv;  // pixel of (u, v): x = x0 + u xu + v xv, y = y0 + u yu + v yv
};

enum { MGM_LATTICES = 12, MGM_LATTICES_16 = 52 };
// lattices swept for nb_dir directions: 4 = the axis ones, 8 = 12 lattices, 16 = 52
S2P_HD int mgm_nlat(int nb_dir) { return nb_dir >= 16 ? MGM_LATTICES_16 : nb_dir >= 8 ? MGM_LATTICES : 4; }
// direction r of the table: 0..3 axis, 4..7 diagonal (the path kernel's order), 8..15 knight's moves
S2P_HD void mgm_direction(int r, int* dx, int* dy)
{
    const int DX[16] = {1, -1, 0, 0, 1, -1, -1, 1, 2, -1, -2, 1, 1, -2, -1, 2};
    const int DY[16] = {0, 0, 1, -1, 1, 1, -1, -1, 1, 2, -1, -2, 2, 1, -2, -1};
    *dx = DX[r & 15]; *dy = DY[r & 15];
}
S2P_HD int mgm_mod5(int a) { const int m = a % 5; return m < 0 ? m + 5 : m; }
S2P_HD int mgm_ceil_div(int a, int b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }        // b > 0
S2P_HD int mgm_floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

S2P_HD MgmLattice mgm_lattice(int q, int w, int h)
{
    MgmLattice l;
    if (q < 4) {
        l.r = q; l.U = w; l.V = h; l.xv = 0; l.yu = 0;
        const bool fx = (q == 1 || q == 2), fy = (q == 1 || q == 3);     // predecessors to the right / below
        l.x0 = fx ? w - 1 : 0; l.xu = fx ? -1 : 1;
        l.y0 = fy ? h - 1 : 0; l.yv = fy ? -1 : 1;
        return l;
    }
    if (q >= MGM_LATTICES) {                                              // knight's move, residue class k
        const int r = 8 + (q - MGM_LATTICES) / 5, k = (q - MGM_LATTICES) % 5;
        int dx, dy;
        mgm_direction(r, &dx, &dy);
        const int ax = dx * (w - 1), ay = dy * (h - 1), bx = -dy * (w - 1), by = dx * (h - 1);
        const int amin = (ax < 0 ? ax : 0) + (ay < 0 ? ay : 0), amax = (ax > 0 ? ax : 0) + (ay > 0 ? ay : 0);
        const int bmin = (bx < 0 ? bx : 0) + (by < 0 ? by : 0), bmax = (bx > 0 ? bx : 0) + (by > 0 ? by : 0);
        const int a0 = amin + k;
        // the class of b that goes with a0: the point (t, 0) with dx t = a0 mod 5 has b = -dy t
        const int inv = dx == 1 ? 1 : dx == 2 ? 3 : dx == -1 ? 4 : 2;     // dx inv = 1 mod 5
        const int t = mgm_mod5(mgm_mod5(a0) * inv);
        const int b0 = bmin + mgm_mod5(mgm_mod5(-dy * t) - bmin);
        l.r = r;
        l.U = a0 <= amax ? (amax - a0) / 5 + 1 : 0;
        l.V = b0 <= bmax ? (bmax - b0) / 5 + 1 : 0;
        l.x0 = (dx * a0 - dy * b0) / 5; l.y0 = (dy * a0 + dx * b0) / 5;  // both numerators are multiples of 5
        l.xu = dx; l.xv = -dy; l.yu = dy; l.yv = dx;
        return l;
    }
    const int r = 4 + ((q - 4) >> 1), p = (q - 4) & 1;
    const bool transposed = (r == 5 || r == 7);                           // X runs along y
    const int WX = transposed ? h : w, HY = transposed ? w : h;
    const int pb = (WX - 1 - p) & 1;
    const int top = WX + HY - 2;                                          // largest a and largest b
    l.r = r;
    l.U = top - p >= 0 ? ((top - p) >> 1) + 1 : 0;
    l.V = top - pb >= 0 ? ((top - pb) >> 1) + 1 : 0;
    const int cX = (p - pb + WX - 1) >> 1, cY = (p + pb - WX + 1) >> 1;   // both numerators are even
    // X = u - v + cX, Y = u + v + cY
    switch (r) {
        case 4:  l.x0 = cX; l.xu = 1; l.xv = -1;  l.y0 = cY; l.yu = 1; l.yv = 1; break;                    // x = X, y = Y
        case 6:  l.x0 = cX; l.xu = 1; l.xv = -1;  l.y0 = h - 1 - cY; l.yu = -1; l.yv = -1; break;          // x = X, y = h-1-Y
        case 5:  l.y0 = cX; l.yu = 1; l.yv = -1;  l.x0 = w - 1 - cY; l.xu = -1; l.xv = -1; break;          // y = X, x = w-1-Y
        default: l.y0 = cX; l.yu = 1; l.yv = -1;  l.x0 = cY; l.xu = 1; l.xv = 1; break;                    // y = X, x = Y
    }
    return l;
}

S2P_HD bool mgm_lattice_pixel(const MgmLattice& l, int w, int h, int u, int v, int* x, int* y)
{
    *x = l.x0 + u * l.xu + v * l.xv;
    *y = l.y0 + u * l.yu + v * l.yv;
    return u >= 0 && u < l.U && v >= 0 && v < l.V && *x >= 0 && *x < w && *y >= 0 && *y < h;
}

// The points of lattice row v that lie in the image form ONE interval of u (x and y are affine in u, slopes in
// {-2, ..., 2}): [*lo, *lo + *span); span = 0 for a row outside the lattice.
S2P_HD void mgm_row_interval(const MgmLattice& l, int w, int h, int v, int* lo, int* span)
{
    int a0 = 0, a1 = (v >= 0 && v < l.V) ? l.U : 0;
    const int cx = l.x0 + v * l.xv, cy = l.y0 + v * l.yv;
    // 0 <= c + u s <= n - 1 with s > 0: ceil(-c / s) <= u <= floor((n - 1 - c) / s); with s < 0: ceil((c - n + 1) / -s) <= u <= floor(c / -s)
    if (l.xu > 0) { const int lo_ = mgm_ceil_div(-cx, l.xu), hi_ = mgm_floor_div(w - 1 - cx, l.xu) + 1; a0 = a0 > lo_ ? a0 : lo_; a1 = a1 < hi_ ? a1 : hi_; }
    else if (l.xu < 0) { const int lo_ = mgm_ceil_div(cx - w + 1, -l.xu), hi_ = mgm_floor_div(cx, -l.xu) + 1; a0 = a0 > lo_ ? a0 : lo_; a1 = a1 < hi_ ? a1 : hi_; }
    else if ((unsigned)cx >= (unsigned)w) a1 = 0;
    if (l.yu > 0) { const int lo_ = mgm_ceil_div(-cy, l.yu), hi_ = mgm_floor_div(h - 1 - cy, l.yu) + 1; a0 = a0 > lo_ ? a0 : lo_; a1 = a1 < hi_ ? a1 : hi_; }
    else if (l.yu < 0) { const int lo_ = mgm_ceil_div(cy - h + 1, -l.yu), hi_ = mgm_floor_div(cy, -l.yu) + 1; a0 = a0 > lo_ ? a0 : lo_; a1 = a1 < hi_ ? a1 : hi_; }
    else if ((unsigned)cy >= (unsigned)h) a1 = 0;
    *lo = a0; *span = a1 > a0 ? a1 - a0 : 0;
}

}  // namespace s2p
