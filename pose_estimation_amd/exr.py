"""Minimal OpenEXR reader / writer for the depth images of the data-creation path.

The reference reads the 'R' channel of the Blender depth renders through the OpenEXR / Imath
packages (data_creator/render_wt_pt_proj/run_proj.py:49-57, written by
modified_RenderForCNN/render_model_views.py:188-206).  This module decodes the same files with numpy
and zlib only: single-part scanline images, FLOAT / HALF / UINT channels, compression NONE, RLE, ZIPS
and ZIP (what Blender's EXR output offers besides the lossy and wavelet codecs, which are rejected with
a clear error).  `write_exr` produces such files (used for fixtures and for caching depth maps).
"""
import struct
import zlib

import numpy as np

_MAGIC = 20000630
_PIXEL = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}
_COMPRESSION = {0: ("NONE", 1), 1: ("RLE", 1), 2: ("ZIPS", 1), 3: ("ZIP", 16)}
_UNSUPPORTED = {4: "PIZ", 5: "PXR24", 6: "B44", 7: "B44A", 8: "DWAA", 9: "DWAB"}


def _cstr(buf, pos):
    end = buf.index(b"\0", pos)
    return buf[pos:end].decode("latin-1"), end + 1


def read_header(buf):
    """-> (attributes dict name -> (type, raw bytes), offset just behind the header)."""
    magic, version = struct.unpack_from("<iI", buf, 0)
    if magic != _MAGIC:
        raise ValueError("not an OpenEXR file (bad magic number)")
    if version & 0xFF != 2:
        raise ValueError(f"unsupported OpenEXR version {version & 0xFF}")
    if version & 0x200:
        raise ValueError("tiled OpenEXR files are not supported (scanline images only)")
    if version & 0x1800:
        raise ValueError("multi-part / deep OpenEXR files are not supported")
    pos = 8
    attrs = {}
    while True:
        name, pos = _cstr(buf, pos)
        if not name:
            break
        kind, pos = _cstr(buf, pos)
        (size,) = struct.unpack_from("<i", buf, pos)
        pos += 4
        attrs[name] = (kind, bytes(buf[pos:pos + size]))
        pos += size
    return attrs, pos


def _channels(raw):
    out, pos = [], 0
    while raw[pos] != 0:
        name, pos = _cstr(raw, pos)
        ptype, _plinear, xs, ys = struct.unpack_from("<iB3xii", raw, pos)
        pos += 16
        out.append((name, ptype, xs, ys))
    return out


def _undo_predictor(data):
    """Inverse of the ZIP / RLE byte predictor and of the even / odd byte split."""
    t = np.frombuffer(data, dtype=np.uint8).copy()
    if t.size:
        # t[i] = t[i-1] + d[i] - 128 (mod 256)  ==  cumulative sum of (d[i] - 128) on top of d[0]; unsigned
        # 8-bit arithmetic wraps modulo 256 by itself
        t[1:] += 128                                   # -128 == +128 (mod 256)
        t = np.cumsum(t, dtype=np.uint8)
    half = (t.size + 1) // 2
    out = np.empty(t.size, dtype=np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _unrle(data, expected):
    out = bytearray()
    pos, n = 0, len(data)
    while pos < n:
        count = data[pos] - 256 if data[pos] > 127 else data[pos]
        pos += 1
        if count < 0:                                  # a run of -count literal bytes
            out += data[pos:pos - count]
            pos += -count
        else:                                          # count + 1 copies of the next byte
            out += bytes([data[pos]]) * (count + 1)
            pos += 1
    if len(out) != expected:
        raise ValueError("corrupt RLE block in the OpenEXR file")
    return bytes(out)


def read_exr(path_or_bytes, channels=None):
    """Decode a scanline OpenEXR image -> dict channel name -> (H, W) array (float32 for FLOAT and HALF
    channels, uint32 for UINT).  `channels`: names to decode (default: all)."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        buf = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as fh:
            buf = fh.read()
    attrs, pos = read_header(buf)
    for need in ("channels", "compression", "dataWindow"):
        if need not in attrs:
            raise ValueError(f"OpenEXR header has no '{need}' attribute")
    chans = _channels(attrs["channels"][1])
    comp = attrs["compression"][1][0]
    if comp in _UNSUPPORTED:
        raise ValueError(f"OpenEXR compression {_UNSUPPORTED[comp]} is not supported (NONE, RLE, ZIPS, ZIP are)")
    if comp not in _COMPRESSION:
        raise ValueError(f"unknown OpenEXR compression code {comp}")
    lines_per_block = _COMPRESSION[comp][1]
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    width, height = x1 - x0 + 1, y1 - y0 + 1
    if width <= 0 or height <= 0:
        raise ValueError("empty OpenEXR data window")
    for name, ptype, xs, ys in chans:
        if ptype not in _PIXEL:
            raise ValueError(f"channel {name}: unknown pixel type {ptype}")
        if xs != 1 or ys != 1:
            raise ValueError(f"channel {name}: subsampled channels are not supported")
    wanted = [c[0] for c in chans] if channels is None else list(channels)
    missing = [c for c in wanted if c not in [d[0] for d in chans]]
    if missing:
        raise KeyError(f"no channel(s) {missing} in the file (it has {[c[0] for c in chans]})")
    out = {name: np.empty((height, width), dtype=_PIXEL[ptype]) for name, ptype, _, _ in chans if name in wanted}
    line_bytes = sum(_PIXEL[ptype].itemsize * width for _, ptype, _, _ in chans)
    nblocks = (height + lines_per_block - 1) // lines_per_block
    offsets = struct.unpack_from(f"<{nblocks}Q", buf, pos)
    for off in offsets:
        y, size = struct.unpack_from("<ii", buf, off)
        data = buf[off + 8:off + 8 + size]
        if len(data) != size:
            raise ValueError("truncated OpenEXR file")
        row0 = y - y0
        rows = min(lines_per_block, height - row0)
        if row0 < 0 or rows <= 0:
            raise ValueError("OpenEXR block outside the data window")
        expected = rows * line_bytes
        if size != expected:                           # a block that did not shrink is stored raw
            if comp in (2, 3):
                data = _undo_predictor(zlib.decompress(data))
            elif comp == 1:
                data = _undo_predictor(_unrle(data, expected))
            if len(data) != expected:
                raise ValueError("OpenEXR block has the wrong size after decompression")
        block = np.frombuffer(data, dtype=np.uint8).reshape(rows, line_bytes)
        col = 0
        for name, ptype, _, _ in chans:                # channels are stored in header (alphabetical) order
            nbytes = _PIXEL[ptype].itemsize * width
            if name in out:
                out[name][row0:row0 + rows] = np.ascontiguousarray(block[:, col:col + nbytes]).view(_PIXEL[ptype])
            col += nbytes
    return {k: (v.astype(np.float32) if v.dtype.kind == "f" else v.astype(np.uint32)) for k, v in out.items()}


def read_exr_channel(path_or_bytes, channel="R"):
    """The reference's get_depthmap_from_exr (run_proj.py:49-57): one channel as (H, W) float32.
    A file whose channel is stored with a layer prefix ('Image.R', 'depth.R') is accepted when the
    match is unique."""
    try:
        return read_exr(path_or_bytes, [channel])[channel]
    except KeyError:
        every = read_exr(path_or_bytes)
        hits = [k for k in every if k.split(".")[-1] == channel]
        if len(hits) != 1:
            raise
        return every[hits[0]]


def write_exr(path, channels, compression="ZIP", half=False):
    """Write (H, W) arrays as a scanline OpenEXR file.  channels: dict name -> array (float arrays are
    stored as FLOAT, or HALF with half=True; unsigned integer arrays as UINT)."""
    codes = {v[0]: k for k, v in _COMPRESSION.items()}
    if compression not in codes:
        raise ValueError(f"compression must be one of {sorted(codes)}")
    comp = codes[compression]
    lines_per_block = _COMPRESSION[comp][1]
    names = sorted(channels)
    arrays, types = [], []
    for name in names:
        a = np.asarray(channels[name])
        if a.ndim != 2 or a.shape != np.asarray(channels[names[0]]).shape:
            raise ValueError("all channels must be (H, W) arrays of one shape")
        if a.dtype.kind == "u":
            arrays.append(a.astype("<u4"))
            types.append(0)
        else:
            arrays.append(a.astype("<f2" if half else "<f4"))
            types.append(1 if half else 2)
    height, width = arrays[0].shape

    def attr(name, kind, payload):
        return name.encode() + b"\0" + kind.encode() + b"\0" + struct.pack("<i", len(payload)) + payload
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", t, 0, 1, 1) for n, t in zip(names, types)) + b"\0"
    window = struct.pack("<4i", 0, 0, width - 1, height - 1)
    header = struct.pack("<iI", _MAGIC, 2)
    header += attr("channels", "chlist", chlist)
    header += attr("compression", "compression", bytes([comp]))
    header += attr("dataWindow", "box2i", window)
    header += attr("displayWindow", "box2i", window)
    header += attr("lineOrder", "lineOrder", b"\0")
    header += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    header += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
    header += attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    header += b"\0"
    blocks = []
    for row0 in range(0, height, lines_per_block):
        rows = min(lines_per_block, height - row0)
        raw = b"".join(a[row0 + r].tobytes() for r in range(rows) for a in arrays)
        data = raw
        if comp in (1, 2, 3):
            t = np.frombuffer(raw, dtype=np.uint8)
            split = np.concatenate([t[0::2], t[1::2]]).astype(np.int64)
            pred = split.copy()
            pred[1:] = (split[1:] - split[:-1] + 128 + 256) & 0xFF
            packed = pred.astype(np.uint8).tobytes()
            data = zlib.compress(packed) if comp in (2, 3) else _rle(packed)
            if len(data) >= len(raw):
                data = raw
        blocks.append(struct.pack("<ii", row0, len(data)) + data)
    table_pos = len(header)
    pos = table_pos + 8 * len(blocks)
    offsets = []
    for blk in blocks:
        offsets.append(pos)
        pos += len(blk)
    with open(path, "wb") as fh:
        fh.write(header + struct.pack(f"<{len(offsets)}Q", *offsets) + b"".join(blocks))


def _rle(data):
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        run = 1
        while i + run < n and run < 128 and data[i + run] == data[i]:
            run += 1
        if run >= 3:
            out += bytes([run - 1, data[i]])
            i += run
            continue
        j = i
        while j < n and j - i < 127:
            if j + 2 < n and data[j] == data[j + 1] == data[j + 2]:
                break
            j += 1
        out += bytes([(-(j - i)) & 0xFF]) + data[i:j]
        i = j
    return bytes(out)
