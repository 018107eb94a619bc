"""Reading the reference's preprocessed cases from disk into HBM.

The reference's preprocessing writes one zarr (v2) group per case -- `<id>.zarr/image`, float32, chunks (1, 1, H, W), compressor
Blosc(cname='zstd', clevel=5, shuffle=BITSHUFFLE) (configuration.py:1403-1412) -- plus `<id>.pkl` with the case's properties, and
`MedicalDataset.load_image` (data_processing.py:535-561) opens `.zarr`, else `.npy`, else `.npz['data']`, else a blosc2 `.b2nd`.
Here a case is read ONCE, on the host, and uploaded (`ResidentDataset.add_case`); there is no per-sample reader in the step.

zarr, numcodecs and blosc are not installed in this image and nothing under /root/reference holds a chunk file: the directory layout
(zarr v2 spec) and the Blosc-1 chunk format (c-blosc README_CHUNK_FORMAT: 16-byte header, block start table, per-block split streams,
byte / bit shuffle) are implemented from their published descriptions -- PARITY UNPINNED (tests write chunks with an encoder stated
from the same description).  zstd comes from pyarrow's codec (the only zstd binding in the image), zlib from the standard library.
`.b2nd` (blosc2 frames) is not read.
"""
from __future__ import annotations

import io
import json
import os
import pickle
import struct
import zlib

import numpy as np

_BLOSC_CODECS = {0: "blosclz", 1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}
_MAX_SPLITS, _MIN_BUFFERSIZE = 16, 128


def _codec_decompress(name, buf, nbytes):
    if name == "zlib":
        return zlib.decompress(buf)
    if name in ("zstd", "lz4", "snappy"):
        import pyarrow as pa
        return pa.Codec({"lz4": "lz4_raw"}.get(name, name)).decompress(buf, decompressed_size=nbytes, asbytes=True)
    raise NotImplementedError(f"Blosc inner codec '{name}' (the reference writes zstd)")


def _unshuffle(block: bytes, typesize: int) -> bytes:
    n = len(block) // typesize
    body = np.frombuffer(block, np.uint8, n * typesize).reshape(typesize, n).T
    return body.tobytes() + block[n * typesize:]


def _bitunshuffle(block: bytes, typesize: int) -> bytes:
    """Inverse of bitshuffle's bit transposition as Blosc applies it: the first 8*floor(n/8) elements are stored as [byte of element]
    [bit][n/8 bytes], bit i of stored byte m belonging to element 8m + i; what is left is stored as it is."""
    n = (len(block) // typesize) // 8 * 8
    if n == 0:
        return block
    rows = np.frombuffer(block, np.uint8, n * typesize).reshape(typesize, 8, n // 8)
    bits = np.unpackbits(rows, axis=-1, bitorder="little")            # [byte][bit][element]
    elems = np.packbits(bits.transpose(2, 0, 1), axis=-1, bitorder="little")  # [element][byte][1]
    return elems.tobytes() + block[n * typesize:]


def blosc_decompress(chunk: bytes) -> bytes:
    """One Blosc-1 chunk -> its bytes."""
    if len(chunk) < 16:
        raise ValueError("Blosc chunk shorter than its header")
    _, _, flags, typesize = chunk[0], chunk[1], chunk[2], chunk[3]
    nbytes, blocksize, cbytes = struct.unpack_from("<III", chunk, 4)
    if cbytes > len(chunk):
        raise ValueError("truncated Blosc chunk")
    if flags & 0x02:  # stored
        return bytes(chunk[16:16 + nbytes])
    if nbytes == 0:
        return b""
    codec = _BLOSC_CODECS.get(flags >> 5)
    nblocks = (nbytes + blocksize - 1) // blocksize
    bstarts = struct.unpack_from(f"<{nblocks}i", chunk, 16)
    out = []
    for b, pos in enumerate(bstarts):
        bsize = min(blocksize, nbytes - b * blocksize)
        leftover = bsize != blocksize
        split = not (flags & 0x10) and typesize <= _MAX_SPLITS and blocksize // typesize >= _MIN_BUFFERSIZE and not leftover
        nsplits = typesize if split else 1
        ne = bsize // nsplits
        parts = []
        for _ in range(nsplits):
            (cb,) = struct.unpack_from("<i", chunk, pos)
            pos += 4
            raw = chunk[pos:pos + cb]
            pos += cb
            parts.append(bytes(raw) if cb == ne else _codec_decompress(codec, raw, ne))
        block = b"".join(parts)
        if len(block) != bsize:
            raise ValueError("Blosc block of unexpected size")
        if flags & 0x04:
            block = _bitunshuffle(block, typesize)
        elif flags & 0x01 and typesize > 1:
            block = _unshuffle(block, typesize)
        out.append(block)
    return b"".join(out)


def _decode_chunk(raw: bytes, compressor, nbytes: int) -> bytes:
    if compressor is None:
        return raw
    cid = compressor["id"]
    if cid == "blosc":
        return blosc_decompress(raw)
    if cid in ("zlib", "gzip"):
        return zlib.decompress(raw, 15 + 32)
    if cid == "zstd":
        return _codec_decompress("zstd", raw, nbytes)
    raise NotImplementedError(f"zarr compressor '{cid}'")


def read_zarr_array(path: str) -> np.ndarray:
    """A zarr v2 array directory (`.zarray` + chunk files 'i.j.k' or 'i/j/k') -> numpy array.  C order, no filters."""
    with open(os.path.join(path, ".zarray")) as f:
        meta = json.load(f)
    if meta.get("zarr_format") != 2 or meta.get("order", "C") != "C" or meta.get("filters"):
        raise NotImplementedError("zarr v2 arrays in C order without filters (what configuration.py:1408-1411 writes)")
    shape, chunks, dtype = tuple(meta["shape"]), tuple(meta["chunks"]), np.dtype(meta["dtype"])
    sep = meta.get("dimension_separator", ".")
    fill = meta.get("fill_value")
    out = np.full(shape, 0 if fill is None else fill, dtype=dtype)
    grid = [(s + c - 1) // c for s, c in zip(shape, chunks)]
    for idx in np.ndindex(*grid):
        fn = os.path.join(path, sep.join(str(i) for i in idx))
        if not os.path.isfile(fn):
            continue  # a missing chunk is all fill_value
        with open(fn, "rb") as f:
            raw = f.read()
        data = np.frombuffer(_decode_chunk(raw, meta.get("compressor"), int(np.prod(chunks)) * dtype.itemsize), dtype).reshape(chunks)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sl] = data[tuple(slice(0, s.stop - s.start) for s in sl)]  # edge chunks are stored whole
    return out


class _PropertiesUnpickler(pickle.Unpickler):
    """The case properties are a dict of numbers, strings, lists and numpy arrays: nothing else is constructed from the file."""
    _OK = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
           ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("collections", "OrderedDict"),
           ("numpy._core.numeric", "_frombuffer"), ("numpy.core.numeric", "_frombuffer")}

    def find_class(self, module, name):
        if (module, name) in self._OK:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"{module}.{name} is not allowed in a properties file")


def load_properties(path: str) -> dict:
    with open(path, "rb") as f:
        return _PropertiesUnpickler(io.BytesIO(f.read())).load()


def load_image(data_path: str, name: str):
    """MedicalDataset.load_image (DATA:535-561): (image [C, D, H, W] or [C, H, W], properties)."""
    zarr_path = os.path.join(data_path, name + ".zarr")
    if os.path.isdir(zarr_path):
        image = read_zarr_array(os.path.join(zarr_path, "image"))
    elif os.path.isfile(os.path.join(data_path, name + ".npy")):
        image = np.load(os.path.join(data_path, name + ".npy"), mmap_mode="r")
    elif os.path.isfile(os.path.join(data_path, name + ".npz")):
        image = np.load(os.path.join(data_path, name + ".npz"))["data"]
    elif os.path.isfile(os.path.join(data_path, name + ".b2nd")):
        raise NotImplementedError("blosc2 .b2nd frames are not read here: re-run the reference's preprocessing (it writes .zarr) or unpack to .npy")
    else:
        raise FileNotFoundError(f"no {name}.zarr / .npy / .npz under {data_path}")
    pkl = os.path.join(data_path, name + ".pkl")
    return image, (load_properties(pkl) if os.path.isfile(pkl) else {})
