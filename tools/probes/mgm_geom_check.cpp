// g++ -O2 -o build/mgm_geom_check tools/probes/mgm_geom_check.cpp && build/mgm_geom_check
// Host-only check of s2p_amd/csrc/mgm_geom.hpp: for every direction, the lattices cover each pixel exactly once,
// and the lattice predecessors (u-1, v), (u, v-1) are the MGM predecessors p - r and p - r_perp (or both outside).
#include <cstdio>
#include <vector>
#include "../../s2p_amd/csrc/mgm_geom.hpp"
using namespace s2p;

static const int DX[8] = {1, -1, 0, 0, 1, -1, -1, 1}, DY[8] = {0, 0, 1, -1, 1, 1, -1, -1};

static int check(int w, int h)
{
    int bad = 0;
    std::vector<int> seen((size_t)8 * w * h, 0);
    for (int q = 0; q < MGM_LATTICES; q++) {
        const MgmLattice l = mgm_lattice(q, w, h);
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
            if (!direct && !swapped) { if (bad < 5) printf("  pred mismatch q=%d r=%d (%d,%d) px (%d,%d)\n", q, l.r, u, v, x, y); bad++; }
        }
    }
    // row intervals: exactly the in-image points of each lattice row
    for (int q = 0; q < MGM_LATTICES; q++) {
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
    for (auto& d : dims) { int b = check(d[0], d[1]); printf("%dx%d: %s\n", d[0], d[1], b ? "FAIL" : "ok"); total += b; }
    return total != 0;
}
