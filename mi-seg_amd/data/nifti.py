"""Minimal NIfTI-1 reader / writer (.nii, .nii.gz) + reorientation to RAS: what `LoadImaged` + `EnsureChannelFirstd` + `Orientationd("RAS")`
deliver at reference data/multi_modal.py:37-40, without nibabel / MONAI (neither is in this image).  Single-file NIfTI-1, little or big
endian, the scalar datatypes MM-WHS-style CT / MR volumes and label maps use; sform preferred over qform over pixdim (nibabel's order)."""
import gzip
import struct

import numpy as np

DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32, 1024: np.int64, 1280: np.uint64}
CODES = {np.dtype(v).name: k for k, v in DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def _quat_affine(b, c, d, qx, qy, qz, dx, dy, dz, qfac):
    a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    A = np.eye(4)
    A[:3, :3] = R * np.array([dx, dy, dz * qfac])
    A[:3, 3] = (qx, qy, qz)
    return A


def read_nifti(path):
    """-> (array with the file's axis order [X, Y, Z(, T...)] scaled by scl_slope / scl_inter when set, 4x4 voxel-to-world affine)"""
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 348:
        raise ValueError(f"{path}: not a NIfTI-1 file (too short)")
    end = "<" if struct.unpack("<i", raw[:4])[0] == 348 else ">"
    if struct.unpack(end + "i", raw[:4])[0] != 348:
        raise ValueError(f"{path}: not a NIfTI-1 file (sizeof_hdr)")
    if raw[344:347] not in (b"n+1",):
        raise ValueError(f"{path}: only single-file NIfTI-1 (magic n+1) is supported")
    dim = struct.unpack(end + "8h", raw[40:56])
    datatype, = struct.unpack(end + "h", raw[70:72])
    pixdim = struct.unpack(end + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + "3f", raw[108:120])
    qform_code, sform_code = struct.unpack(end + "2h", raw[252:256])
    quat = struct.unpack(end + "6f", raw[256:280])
    srow = np.array(struct.unpack(end + "12f", raw[280:328]), dtype=np.float64).reshape(3, 4)
    if datatype not in DTYPES:
        raise ValueError(f"{path}: NIfTI datatype {datatype} is not supported")
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    dt = np.dtype(DTYPES[datatype]).newbyteorder(end)
    n = int(np.prod(shape))
    arr = np.frombuffer(raw, dtype=dt, count=n, offset=int(vox_offset)).reshape(shape, order="F")
    # scl_slope / scl_inter as nibabel reads them: scaling applies only with a finite non-zero slope, a non-finite intercept counts as 0,
    # and (1, 0) is the identity (writers leave NaN in these fields to say "no scaling")
    if np.isfinite(slope) and slope != 0.0:
        if not np.isfinite(inter):
            inter = 0.0
        if (slope, inter) != (1.0, 0.0):
            arr = arr.astype(np.float32) * np.float32(slope) + np.float32(inter)
    if sform_code > 0:
        A = np.vstack([srow, [0, 0, 0, 1]])
    elif qform_code > 0:
        A = _quat_affine(*quat, pixdim[1], pixdim[2], pixdim[3], -1.0 if pixdim[0] < 0 else 1.0)
    else:
        A = np.diag([pixdim[1] or 1.0, pixdim[2] or 1.0, pixdim[3] or 1.0, 1.0])
    return np.ascontiguousarray(arr.astype(dt.newbyteorder("=")) if arr.dtype.byteorder not in ("=", "|") else arr), A


def write_nifti(path, array, affine=None):
    """single-file NIfTI-1 with an sform (used by tests and by the prediction export)"""
    array = np.asarray(array)
    if array.dtype.name not in CODES:
        raise ValueError(f"dtype {array.dtype} has no NIfTI code")
    affine = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64)
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, 348)
    dims = [array.ndim] + list(array.shape) + [1] * (7 - array.ndim)
    struct.pack_into("<8h", hdr, 40, *dims)
    struct.pack_into("<h", hdr, 70, CODES[array.dtype.name])
    struct.pack_into("<h", hdr, 72, array.dtype.itemsize * 8)
    vox = np.sqrt((affine[:3, :3] ** 2).sum(0))
    struct.pack_into("<8f", hdr, 76, 1.0, *[float(v) for v in vox], 0, 0, 0, 0)
    struct.pack_into("<3f", hdr, 108, 352.0, 1.0, 0.0)
    struct.pack_into("<2h", hdr, 252, 0, 1)
    struct.pack_into("<12f", hdr, 280, *[float(v) for v in affine[:3].reshape(-1)])
    hdr[344:348] = b"n+1\0"
    with _open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(np.asfortranarray(array).tobytes(order="F"))


def reorient_to_ras(array, affine):
    """axis permutation + flips that make the voxel axes point Right / Anterior / Superior (closest-axis rule, like nibabel's
    io_orientation / MONAI Orientationd): returns (array, new affine)"""
    R = affine[:3, :3]
    vox = np.sqrt((R ** 2).sum(0))
    vox[vox == 0] = 1.0
    cos = R / vox
    order, flips, used = [], [], set()
    for world in range(3):                       # which voxel axis runs along world axis `world`
        cand = [(abs(cos[world, a]), a) for a in range(3) if a not in used]
        a = max(cand)[1]
        used.add(a)
        order.append(a)
        flips.append(cos[world, a] < 0)
    out = np.transpose(array, order + list(range(3, array.ndim)))
    A = affine[:, order + [3]].copy()
    for ax, fl in enumerate(flips):
        if fl:
            out = np.flip(out, ax)
            A[:, 3] = A[:, 3] + A[:, ax] * (out.shape[ax] - 1)
            A[:, ax] = -A[:, ax]
    return np.ascontiguousarray(out), A
