"""s2p_amd/geographiclib.py: the UTM conversion of the path's tail (known answers + round trips; no pyproj here)."""
import numpy as np
import pytest

from s2p_amd import geographiclib as g


def test_known_answers():
    e, n = g.lonlat_to_utm(3.0, 0.0, 31)
    assert e == 500000.0 and n == 0.0
    # the meridian arc from the equator to 45 N on WGS 84 is 4 984 944.378 m (Karney 2011, table 1)
    e, n = g.lonlat_to_utm(3.0, 45.0, 31)
    assert e == 500000.0 and abs(n / 0.9996 - 4984944.378) < 1e-3
    e, n = g.lonlat_to_utm(57.0, -10.0, 40, south=True)
    assert e == 500000.0 and 8.8e6 < n < 8.9e6
    assert g.compute_utm_zone(55.65, -21.23) == "40S" and g.epsg_code_from_utm_zone("40S") == 32740
    assert g.compute_utm_zone(5.44, 43.26) == "31N" and g.epsg_code_from_utm_zone("31N") == 32631
    assert g.utm_zone_from_epsg("epsg:32740") == (40, True) and g.utm_zone_from_epsg(32631) == (31, False)
    with pytest.raises(NotImplementedError):
        g.utm_zone_from_epsg("epsg:2154")


def test_round_trip_is_exact_to_nanometres():
    rng = np.random.default_rng(0)
    lon, lat = rng.uniform(-3, 9, 5000), rng.uniform(-80, 84, 5000)           # up to 3 degrees outside the zone
    e, n = g.lonlat_to_utm(lon, lat, 31)
    lo, la = g.utm_to_lonlat(e, n, 31)
    assert np.abs(lo - lon).max() < 1e-12 and np.abs(la - lat).max() < 1e-12


def test_scale_and_convergence_on_the_reference_dsm_origin():
    """Finite differences at the corner of the reference's pair DSM: the projection is conformal with scale ~0.9996-1.0004."""
    lon, lat = g.utm_to_lonlat(359746.0, 7651923.0, 40, south=True)
    d = 1e-6
    e0, n0 = g.lonlat_to_utm(lon, lat, 40, True)
    e1, n1 = g.lonlat_to_utm(lon + d, lat, 40, True)
    e2, n2 = g.lonlat_to_utm(lon, lat + d, 40, True)
    assert abs(e0 - 359746.0) < 1e-6 and abs(n0 - 7651923.0) < 1e-6
    a, f = 6378137.0, 1 / 298.257223563
    e2_ = f * (2 - f)
    s = np.sin(np.radians(lat))
    N = a / np.sqrt(1 - e2_ * s * s)
    M = a * (1 - e2_) / (1 - e2_ * s * s) ** 1.5
    k_lon = np.hypot(e1 - e0, n1 - n0) / (np.radians(d) * N * np.cos(np.radians(lat)))
    k_lat = np.hypot(e2 - e0, n2 - n0) / (np.radians(d) * M)
    assert abs(k_lon - k_lat) < 1e-6 and 0.9996 <= k_lon < 1.0004


def test_lonlatalt_to_utm_keeps_nan_rows_and_altitude():
    a = np.array([[[55.65, -21.23, 2300.5], [np.nan, np.nan, np.nan]]])
    out = g.lonlatalt_to_utm(a, "epsg:32740")
    assert out.shape == a.shape and np.isnan(out[0, 1]).all() and out[0, 0, 2] == 2300.5
    assert 3.5e5 < out[0, 0, 0] < 3.7e5 and 7.6e6 < out[0, 0, 1] < 7.7e6


def test_compound_crs_with_a_vertical_datum_is_refused():
    """cfg['out_geoid'] makes the reference ask for 'epsg:326xx+5773' (s2p/initialization.py:139-141): pyproj then applies the EGM96
    geoid.  Dropping the '+5773' silently would bias every altitude by the geoid height."""
    import pytest
    a = np.array([[[2.35, 48.85, 100.0]]])
    with pytest.raises(NotImplementedError):
        g.lonlatalt_to_utm(a, "epsg:32631+5773")
    with pytest.raises(NotImplementedError):
        g.utm_zone_from_epsg("EPSG:32740+5773")
    assert g.utm_zone_from_epsg("epsg:32631") == (31, False)
