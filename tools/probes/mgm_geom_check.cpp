// g++ -O2 -o build/mgm_geom_check tools/probes/mgm_geom_check.cpp && build/mgm_geom_check
// Host-only check of s2p_amd/csrc/mgm_geom.hpp: for every direction, the lattices cover each pixel exactly once,
// and the lattice predecessors (u-1, v), (u, v-1) are the MGM predecessors p - r and p - r_perp (or both outside);
// 4, 8 (12 lattices) and 16 directions (52 lattices: the knight's moves split into 5 residue classes each).
#include <cstdio>
#include <vector>
#include "../../s2p_amd/csrc/mgm_geom.hpp"
using namespace s2p;

// the direction table of the oracle (oracle/census_oracle.c), restated: mgm_direction() must agree with it
static const int DX[16] = {1, -1, 0, 0, 1, -1, -1, 1, 2, -1, -2, 1, 1, -2, -1, 2}, DY[16] = {0, 0, 1, -1, 1, 1, -1, -1, 1, 2, -1, -2, 2, 1, -2, -1};

static int check(int w, int h, int ND)
{
    int bad = 0;
    const int NL = mgm_nlat(ND);
    std::vector<int> seen((size_t)ND * w * h, 0);
    for (int q = 0; q < NL; q++) {
        const MgmLattice l = mgm_lattice(q, w, h);
        int tx, ty;
        mgm_direction(l.r, &tx, &ty);
        if (l.r < 0 || l.r >= ND || tx != DX[l.r] || ty != DY[l.r]) { printf("  direction table q=%d r=%d\n", q, l.r); bad++; continue; }
        const int dx = DX[l.r], dy = DY[l.r], ex = -dy, ey = dx;
        for (int v = 0; v < l.V; v++) for (int u = 0; u < l.U; u++) {
            int x, y;
            if (!mgm_lattice_pixel(l, w, h, u, v, &x, &y)) continue;
            seen[((size_t)l.r * h + y) * w + x]++;
            // the two lattice predecessors, as pixels (or "outside")
            int ax, ay, bx, by;
            const bool ina = mgm_lattice_pixel(l, w, h, u - 1, v, &ax, &ay), inb = mgm_lattice_pixel(l, w, h, u, v - 1, &bx, &by);
            const int p1x = x - dx, p1y = y - dy, p2x = x - ex, p2y = y - ey;
            const bool in1 = p1x >= 0 && p1x < w && p1y >= 0 && p1y < h, in2 = p2x >= 0 && p2x < w && p2y >= 0 && p2y < h;
            // {a, b} must equal {p1, p2} as sets, with matching in/out status
            const bool direct = (ina == in1) && (inb == in2) && (!ina || (ax == p1x && ay == p1y)) && (!inb || (bx == p2x && by == p2y));
            const bool swapped = (ina == in2) && (inb == in1) && (!ina || (ax == p2x && ay == p2y)) && (!inb || (bx == p1x && by == p1y));
            // the third predecessor of the quadrant (recursion = 2): (u - 1, v - 1) is p - r - r_perp
            int cx3, cy3;
            const bool inc = mgm_lattice_pixel(l, w, h, u - 1, v - 1, &cx3, &cy3);
            const int p3x = x - dx - ex, p3y = y - dy - ey;
            const bool in3 = p3x >= 0 && p3x < w && p3y >= 0 && p3y < h;
            if (inc != in3 || (inc && (cx3 != p3x || cy3 != p3y))) { if (bad < 5) printf("  third pred q=%d (%d,%d)\n", q, u, v); bad++; }
            if (!direct && !swapped) { if (bad < 5) printf("  pred mismatch q=%d r=%d (%d,%d) px (%d,%d)\n", q, l.r, u, v, x, y); bad++; }
        }
    }
    // row intervals: exactly the in-image points of each lattice row
    for (int q = 0; q < NL; q++) {
        const MgmLattice l = mgm_lattice(q, w, h);
        for (int v = -1; v <= l.V; v++) {
            int lo, span, x, y;
            mgm_row_interval(l, w, h, v, &lo, &span);
            for (int u = -2; u < l.U + 2; u++) {
                const bool in = mgm_lattice_pixel(l, w, h, u, v, &x, &y), claimed = (unsigned)(u - lo) < (unsigned)span;
                if (in != claimed) { if (bad < 5) printf("  interval q=%d v=%d u=%d in=%d lo=%d span=%d\n", q, v, u, in, lo, span); bad++; }
            }
        }
    }
    for (size_t i = 0; i < seen.size(); i++) if (seen[i] != 1) { if (bad < 5) printf("  cover %zu = %d\n", i, seen[i]); bad++; }
    return bad;
}

int main()
{
    int total = 0;
    const int dims[][2] = {{1, 1}, {1, 5}, {5, 1}, {2, 2}, {3, 2}, {2, 3}, {4, 4}, {5, 4}, {4, 5}, {7, 7}, {16, 9}, {9, 16}, {33, 20}, {20, 33}, {64, 64}, {257, 131}, {131, 257}};
    for (int nd = 4; nd <= 16; nd *= 2)
        for (auto& d : dims) { int b = check(d[0], d[1], nd); printf("%d directions %dx%d: %s\n", nd, d[0], d[1], b ? "FAIL" : "ok"); total += b; }
    return total != 0;
}
