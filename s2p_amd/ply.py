"""s2p_amd/ply.py -- the two functions of s2p/ply.py without the `plyfile` package (absent from this image).

    read_3d_point_cloud_from_ply(path)  -> (n, nprops) array, list of header comments        (s2p/ply.py:7-21)
    write_3d_point_cloud_to_ply(path, coordinates, colors=None, extra_properties=None,
                                extra_properties_names=None, comments=[])                     (s2p/ply.py:24-64)

The writer produces the file plyfile writes for the same arguments -- binary little endian, one `property` line per
column with plyfile's type names -- byte for byte: re-writing the reference's own tests/data/input_ply/cloud.ply
reproduces it exactly (tests/test_oracle_raster.py).  Host-side code: nothing here touches the GPU."""
import numpy as np

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def read_3d_point_cloud_from_ply(path_to_ply_file):
    """The reader of s2p/ply.py:7-21 without the plyfile package: (n, nprops) array with one column per vertex
    property in file order (numpy's common type, float64 for s2p's clouds), and the list of header comments."""
    with open(path_to_ply_file, "rb") as f:
        raw = f.read()
    end = raw.index(b"end_header")
    end = raw.index(b"\n", end) + 1
    lines = raw[:end].decode("ascii", "replace").splitlines()
    if not lines or lines[0].strip() != "ply":
        raise ValueError("%s: not a PLY file" % path_to_ply_file)
    fmt, comments, props, n, in_vertex = None, [], [], 0, False
    for l in lines[1:]:
        t = l.split()
        if not t:
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "comment":
            comments.append(l[len("comment "):])
        elif t[0] == "element":
            in_vertex = t[1] == "vertex"
            if in_vertex:
                n = int(t[2])
        elif t[0] == "property" and in_vertex:
            if t[1] == "list":
                raise ValueError("list properties on vertices are not supported")
            props.append((t[-1], _PLY_TYPES[t[1]]))
    if fmt in ("binary_little_endian", "binary_big_endian"):
        e = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(name, e + ty) for name, ty in props])
        d = np.frombuffer(raw, dtype=dt, count=n, offset=end)
    elif fmt == "ascii":
        rows = np.loadtxt(raw[end:].decode().splitlines()[:n], ndmin=2)
        d = np.empty(n, np.dtype([(name, ty) for name, ty in props]))
        for k, (name, _) in enumerate(props):
            d[name] = rows[:, k]
    else:
        raise ValueError("unknown PLY format %r" % fmt)
    array = np.column_stack([d[name] for name, _ in props]) if n else np.zeros((0, len(props)))
    return array, comments


_PLY_NAMES = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}


def write_3d_point_cloud_to_ply(path_to_ply_file, coordinates, colors=None, extra_properties=None,
                                extra_properties_names=None, comments=[]):
    """
    Write a 3D point cloud to a ply file (arguments as s2p.ply.write_3d_point_cloud_to_ply, s2p/ply.py:24-40).

    Args:
        path_to_ply_file (str): path to a .ply file
        coordinates (array): (n, 3) x, y, z
        colors (array): (n, 3) r, g, b, (n, 4) r, g, b, ir or (n, 1) gray levels (replicated 3 times)
        extra_properties (array): optional (n, k) array
        extra_properties_names (list): the k property names
        comments (list): header comment strings
    """
    coordinates = np.asarray(coordinates)
    cols = [("x", coordinates[:, 0]), ("y", coordinates[:, 1]), ("z", coordinates[:, 2])]
    if colors is not None:
        colors = np.asarray(colors)
        if colors.shape[1] == 1:                                   # replicate grayscale 3 times (:46-47)
            colors = np.column_stack([colors] * 3)
        elif colors.shape[1] not in [3, 4]:
            raise Exception('Error: colors must have either 1, 3 or 4 channels')
        cols += [(n, colors[:, k]) for k, n in enumerate(("red", "green", "blue"))]
        if colors.shape[1] == 4:
            cols += [("ir", colors[:, 3])]
    if extra_properties is not None:
        extra_properties = np.asarray(extra_properties)
        if extra_properties.ndim == 1:                             # np.column_stack takes a 1-D array as one column (:58)
            extra_properties = extra_properties.reshape(-1, 1)
        cols += [(n, extra_properties[:, k]) for k, n in enumerate(extra_properties_names)]
    n = len(coordinates)
    # the reference stacks all columns into one array first (np.column_stack, :50, :58): values pass through the
    # common type of the columns before they are cast to each property's own type
    common = np.result_type(*[c.dtype for _, c in cols])
    dt = np.dtype([(name, c.dtype.newbyteorder("<")) for name, c in cols])
    rec = np.empty(n, dt)
    for name, c in cols:
        rec[name] = c.astype(common).astype(c.dtype)
    header = ["ply", "format binary_little_endian 1.0"] + ["comment " + c for c in comments] + ["element vertex %d" % n]
    header += ["property %s %s" % (_PLY_NAMES[c.dtype.str[1:]], name) for name, c in cols] + ["end_header"]
    with open(path_to_ply_file, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(rec.tobytes())
