// g++ -O2 -shared -fPIC -o mgm_geom_capi.so tools/probes/mgm_geom_capi.cpp
// C wrappers around s2p_amd/csrc/mgm_geom.hpp for the host-side protocol model (tests/test_mgm_protocol_model.py).
#include "../../s2p_amd/csrc/mgm_geom.hpp"
extern "C" {
// out[0..8] = r, U, V, x0, xu, xv, y0, yu, yv
void mgm_capi_lattice(int q, int w, int h, int* out)
{
    const s2p::MgmLattice l = s2p::mgm_lattice(q, w, h);
    out[0] = l.r; out[1] = l.U; out[2] = l.V; out[3] = l.x0; out[4] = l.xu; out[5] = l.xv; out[6] = l.y0; out[7] = l.yu; out[8] = l.yv;
}
void mgm_capi_row_interval(int q, int w, int h, int v, int* lo, int* span)
{
    const s2p::MgmLattice l = s2p::mgm_lattice(q, w, h);
    s2p::mgm_row_interval(l, w, h, v, lo, span);
}
}
