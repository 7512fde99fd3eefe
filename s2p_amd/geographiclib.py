"""s2p_amd/geographiclib.py -- the one coordinate conversion the path's tail needs without pyproj.

The reference converts the triangulated lon / lat / alt to the output CRS with pyproj
(s2p/geographiclib.py:122-158, called from s2p/triangulation.py:148-162 and :214-219); its default output CRS is the
UTM zone of the ROI on the WGS 84 ellipsoid (s2p/initialization.py:133-138, s2p/geographiclib.py:40-81).  This module
states that default case in numpy float64: the transverse Mercator projection by the Krueger series to n^6 (Karney,
"Transverse Mercator with an accuracy of a few nanometers", 2011 -- the series PROJ evaluates as well), so that the
cloud of a tile can be rasterised where pyproj is not installed.  Other CRSs stay with pyproj.
"""
import numpy as np

_A = 6378137.0                       # WGS 84
_F = 1.0 / 298.257223563
_K0 = 0.9996


def _series():
    n = _F / (2.0 - _F)
    n2, n3, n4, n5, n6 = n * n, n ** 3, n ** 4, n ** 5, n ** 6
    rect = _A / (1.0 + n) * (1.0 + n2 / 4.0 + n4 / 64.0 + n6 / 256.0)
    alpha = (n / 2 - 2 * n2 / 3 + 5 * n3 / 16 + 41 * n4 / 180 - 127 * n5 / 288 + 7891 * n6 / 37800,
             13 * n2 / 48 - 3 * n3 / 5 + 557 * n4 / 1440 + 281 * n5 / 630 - 1983433 * n6 / 1935360,
             61 * n3 / 240 - 103 * n4 / 140 + 15061 * n5 / 26880 + 167603 * n6 / 181440,
             49561 * n4 / 161280 - 179 * n5 / 168 + 6601661 * n6 / 7257600,
             34729 * n5 / 80640 - 3418889 * n6 / 1995840,
             212378941 * n6 / 319334400)
    beta = (n / 2 - 2 * n2 / 3 + 37 * n3 / 96 - n4 / 360 - 81 * n5 / 512 + 96199 * n6 / 604800,
            n2 / 48 + n3 / 15 - 437 * n4 / 1440 + 46 * n5 / 105 - 1118711 * n6 / 3870720,
            17 * n3 / 480 - 37 * n4 / 840 - 209 * n5 / 4480 + 5569 * n6 / 90720,
            4397 * n4 / 161280 - 11 * n5 / 504 - 830251 * n6 / 7257600,
            4583 * n5 / 161280 - 108847 * n6 / 3991680,
            20648693 * n6 / 638668800)
    return rect, alpha, beta


_RECT, _ALPHA, _BETA = _series()
_E = np.sqrt(_F * (2.0 - _F))


def compute_utm_zone(lon, lat):
    """UTM zone number + hemisphere letter of a point, e.g. '40S' (s2p/geographiclib.py:40-57)."""
    zone = int((float(lon) + 180.0) // 6.0) % 60 + 1
    return "%d%s" % (zone, "N" if lat >= 0 else "S")


def epsg_code_from_utm_zone(utm_zone):
    """EPSG code of a UTM zone on WGS 84: 326xx north, 327xx south (s2p/geographiclib.py:60-81)."""
    zone, south = int(utm_zone[:-1]), utm_zone[-1] == "S"
    return (32700 if south else 32600) + zone


def utm_zone_from_epsg(epsg):
    """(zone, south) of "epsg:326xx" / "epsg:327xx" (or the integer).  A compound CRS ("epsg:32631+5773": what the reference
    builds when cfg['out_geoid'] is set, s2p/initialization.py:139-141) carries a VERTICAL datum -- pyproj then turns ellipsoidal
    into orthometric heights with the EGM96 geoid, tens of metres apart; nothing here does that, so it is refused, not ignored."""
    text = str(epsg).lower().replace("epsg:", "")
    if "+" in text:
        raise NotImplementedError("compound CRS %r: the vertical datum (geoid heights) stays with pyproj" % (epsg,))
    code = int(text)
    if 32601 <= code <= 32660:
        return code - 32600, False
    if 32701 <= code <= 32760:
        return code - 32700, True
    raise NotImplementedError("epsg:%d is not a WGS 84 / UTM zone: that CRS stays with pyproj" % code)


def lonlat_to_utm(lon, lat, zone, south=False):
    """WGS 84 longitude / latitude in degrees -> UTM easting, northing in metres of `zone` (arrays, float64)."""
    lon = np.asarray(lon, np.float64)
    lat = np.asarray(lat, np.float64)
    lam = np.radians(lon - (6.0 * zone - 183.0))
    tau = np.tan(np.radians(lat))
    sig = np.sinh(_E * np.arctanh(_E * tau / np.sqrt(1.0 + tau * tau)))
    taup = tau * np.sqrt(1.0 + sig * sig) - sig * np.sqrt(1.0 + tau * tau)
    xi0 = np.arctan2(taup, np.cos(lam))
    eta0 = np.arcsinh(np.sin(lam) / np.sqrt(taup * taup + np.cos(lam) ** 2))
    xi, eta = xi0.copy(), eta0.copy()
    for j, a in enumerate(_ALPHA, 1):
        xi += a * np.sin(2 * j * xi0) * np.cosh(2 * j * eta0)
        eta += a * np.cos(2 * j * xi0) * np.sinh(2 * j * eta0)
    east = 500000.0 + _K0 * _RECT * eta
    north = _K0 * _RECT * xi + (10000000.0 if south else 0.0)
    return east, north


def utm_to_lonlat(east, north, zone, south=False):
    """Inverse of lonlat_to_utm (Krueger series + Newton on the conformal latitude)."""
    east = np.asarray(east, np.float64)
    north = np.asarray(north, np.float64)
    xi = (north - (10000000.0 if south else 0.0)) / (_K0 * _RECT)
    eta = (east - 500000.0) / (_K0 * _RECT)
    xi0, eta0 = xi.copy(), eta.copy()
    for j, b in enumerate(_BETA, 1):
        xi0 -= b * np.sin(2 * j * xi) * np.cosh(2 * j * eta)
        eta0 -= b * np.cos(2 * j * xi) * np.sinh(2 * j * eta)
    taup = np.sin(xi0) / np.sqrt(np.sinh(eta0) ** 2 + np.cos(xi0) ** 2)
    lam = np.arctan2(np.sinh(eta0), np.cos(xi0))
    tau = taup.copy()
    for _ in range(6):                                       # Newton: tau'(tau) = taup
        sig = np.sinh(_E * np.arctanh(_E * tau / np.sqrt(1.0 + tau * tau)))
        f = tau * np.sqrt(1.0 + sig * sig) - sig * np.sqrt(1.0 + tau * tau) - taup
        df = (np.sqrt((1.0 + sig * sig) * (1.0 + tau * tau)) - sig * tau) * (1.0 - _E * _E) * np.sqrt(1.0 + tau * tau) \
            / (1.0 + (1.0 - _E * _E) * tau * tau)
        tau = tau - f / df
    return np.degrees(lam) + (6.0 * zone - 183.0), np.degrees(np.arctan(tau))


def lonlatalt_to_utm(lonlatalt, out_crs):
    """(..., 3) lon / lat / alt -> (..., 3) easting / northing / alt in the UTM zone `out_crs` names ("epsg:32740"): the
    conversion disp_to_xyz / height_map_to_xyz apply through pyproj when out_crs is the default UTM zone
    (s2p/triangulation.py:148-162).  NaN rows stay NaN; the altitude (ellipsoidal) is unchanged."""
    zone, south = utm_zone_from_epsg(out_crs)
    a = np.asarray(lonlatalt, np.float64)
    out = np.empty_like(a)
    out[..., 0], out[..., 1] = lonlat_to_utm(a[..., 0], a[..., 1], zone, south)
    out[..., 2] = a[..., 2]
    return out
