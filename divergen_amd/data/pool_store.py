"""Instance-pool shard store.

The reference keeps the 1.2 M generated instances as PNG pairs on disk and opens them per sample with PIL
(`InstPool._load_RGBA`, DG/divergen/data/custom_build_copypaste_mapper.py:359-383: `Image.open(path).convert('RGBA')`,
alpha replaced by `Image.open(mask_path)` for `path|maskpath` keys, `*path` = already pre-processed); the pool index is the
`INST_POOL_PATH` json `{cat_id: [key, ...]}` (:115-154, written by DG/tools/merge_inst_pool_json.py).  This store keeps the
json index and its keys unchanged and replaces only the per-sample decode: the RGBA pixels of every key are decoded ONCE
into length-prefixed records of flat shard files that loader workers mmap (page cache shared by all workers of a node, no
zlib inflate, no per-sample allocation besides the resize the mapper does anyway).

Shard layout (little endian, every record 64-byte aligned):
    header  64 B : magic "DGXPOOL1" | u32 version | u32 n_records | u64 index_offset | u64 reserved...
    record       : u32 h | u32 w | u32 flags | u32 crc32(pixels) | u32 key_len | u32 pad[3] | key bytes | pad to 16 |
                   h*w*4 bytes RGBA, row-major | pad to 64
    index        : n_records x (u64 key_hash, u64 record_offset), sorted by hash (key_hash = blake2b-8 of the key)
`flags` bit 0 = the key was a `*` (pre-processed) entry.  A reader verifies the stored key (hash collisions) and, on
request, the crc.  `PoolStore.loader` has the signature `InstPool(loader=...)` expects and returns exactly the array
`InstPool._pil_loader` returns for the same key."""
import glob
import hashlib
import json
import mmap
import os
import struct
import zlib

import numpy as np

MAGIC = b"DGXPOOL1"
VERSION = 1
HEADER = struct.Struct("<8sIIQ")           # magic, version, n_records, index_offset  (padded to 64 bytes)
RECORD = struct.Struct("<IIIII12x")        # h, w, flags, crc32, key_len  (32 bytes)
INDEX_DTYPE = np.dtype([("hash", "<u8"), ("offset", "<u8")])
FLAG_PREPROCESSED = 1


def key_hash(key):
    return int.from_bytes(hashlib.blake2b(key.encode("utf-8"), digest_size=8).digest(), "little")


def decode_key(key):
    """The reference's decode of one pool key (mapper.py:366-383) -> (h,w,4) uint8."""
    from PIL import Image
    path, mask_path = key, None
    if path.startswith("*"):
        path = path[1:]
    elif "|" in path:
        path, mask_path = path.split("|")[:2]
    rgba = np.array(Image.open(path).convert("RGBA"))
    if mask_path is not None:
        rgba[:, :, -1] = np.array(Image.open(mask_path))
    return rgba


def _align(n, a):
    return (n + a - 1) // a * a


class ShardWriter:
    """Appends records to one shard file; `close()` writes the index and the header."""

    def __init__(self, path):
        self.path = path
        self.f = open(path, "wb")
        self.f.write(b"\0" * 64)
        self.entries = []          # (hash, offset)
        self.keys = set()

    def add(self, key, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        if rgba.ndim != 3 or rgba.shape[2] != 4:
            raise ValueError("pool record must be (h, w, 4) uint8, got %r" % (rgba.shape,))
        if key in self.keys:
            raise ValueError("duplicate pool key %r" % key)
        self.keys.add(key)
        kb = key.encode("utf-8")
        off = self.f.tell()
        assert off % 64 == 0
        px = rgba.tobytes()
        flags = FLAG_PREPROCESSED if key.startswith("*") else 0
        self.f.write(RECORD.pack(rgba.shape[0], rgba.shape[1], flags, zlib.crc32(px) & 0xFFFFFFFF, len(kb)))
        self.f.write(kb)
        self.f.write(b"\0" * (_align(len(kb), 16) - len(kb)))
        self.f.write(px)
        end = self.f.tell()
        self.f.write(b"\0" * (_align(end, 64) - end))
        self.entries.append((key_hash(key), off))
        return off

    @property
    def nbytes(self):
        return self.f.tell()

    def close(self):
        idx = np.array(self.entries, dtype=INDEX_DTYPE) if self.entries else np.zeros(0, INDEX_DTYPE)
        idx.sort(order="hash", kind="stable")
        index_offset = self.f.tell()
        self.f.write(idx.tobytes())
        self.f.seek(0)
        self.f.write(HEADER.pack(MAGIC, VERSION, len(idx), index_offset))
        self.f.close()


def build_shards(pool, out_dir, decode=decode_key, shard_bytes=1 << 30, skip_errors=True, log=None):
    """Decode every key of a pool json ({cat_id: [key, ...]} or its path) once into `out_dir/pool-%05d.dgxpool`.
    The json itself is copied next to the shards (`inst_pool.json`) and stays the index the sampler reads.
    Keys that fail to decode are left out (the reference skips them at run time, mapper.py:374-378) and returned."""
    if isinstance(pool, str):
        with open(pool) as f:
            pool = json.load(f)
    os.makedirs(out_dir, exist_ok=True)
    shard_id, writer, failed, seen, n = 0, None, [], set(), 0
    for cat in pool:
        for key in pool[cat]:
            if key in seen:
                continue
            seen.add(key)
            try:
                rgba = decode(key)
            except Exception as e:      # noqa: BLE001 -- same policy as the reference's bare except
                if not skip_errors:
                    raise
                failed.append(key)
                if log:
                    log("skip %s: %s" % (key, e))
                continue
            if writer is None or (writer.nbytes + rgba.nbytes > shard_bytes and writer.entries):
                if writer is not None:
                    writer.close()
                    shard_id += 1
                writer = ShardWriter(os.path.join(out_dir, "pool-%05d.dgxpool" % shard_id))
            writer.add(key, rgba)
            n += 1
    if writer is None:
        writer = ShardWriter(os.path.join(out_dir, "pool-%05d.dgxpool" % shard_id))
    writer.close()
    with open(os.path.join(out_dir, "inst_pool.json"), "w") as f:
        json.dump(pool, f)
    return {"records": n, "shards": shard_id + 1, "failed": failed}


class PoolStore:
    """Read side: mmap of all shards of a directory + one merged, sorted hash index."""

    def __init__(self, root, verify_crc=False):
        paths = sorted(glob.glob(os.path.join(root, "pool-*.dgxpool"))) if os.path.isdir(root) else [root]
        if not paths:
            raise FileNotFoundError("no pool-*.dgxpool shards under %s" % root)
        self.verify_crc = verify_crc
        self.maps, hashes, offsets, shard_of = [], [], [], []
        for si, p in enumerate(paths):
            f = open(p, "rb")
            size = os.fstat(f.fileno()).st_size
            if size < 64:
                raise ValueError("%s: truncated shard" % p)
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            f.close()
            magic, version, n, index_offset = HEADER.unpack_from(mm, 0)
            if magic != MAGIC or version != VERSION:
                raise ValueError("%s: not a DGXPOOL1 shard (magic %r, version %d)" % (p, magic, version))
            if index_offset + n * INDEX_DTYPE.itemsize > size:
                raise ValueError("%s: index runs past the end of the file" % p)
            idx = np.frombuffer(mm, dtype=INDEX_DTYPE, count=n, offset=index_offset)
            self.maps.append(mm)
            hashes.append(idx["hash"])
            offsets.append(idx["offset"])
            shard_of.append(np.full(n, si, np.int32))
        h = np.concatenate(hashes) if hashes else np.zeros(0, np.uint64)
        order = np.argsort(h, kind="stable")
        self._hash = h[order]
        self._offset = np.concatenate(offsets)[order]
        self._shard = np.concatenate(shard_of)[order]

    def __len__(self):
        return len(self._hash)

    def _record(self, si, off):
        mm = self.maps[si]
        h, w, flags, crc, klen = RECORD.unpack_from(mm, off)
        kstart = off + RECORD.size
        key = bytes(mm[kstart:kstart + klen])
        pstart = kstart + _align(klen, 16)
        return h, w, flags, crc, key, pstart

    def __contains__(self, key):
        return self._find(key) is not None

    def _find(self, key):
        hv = np.uint64(key_hash(key))
        lo = int(np.searchsorted(self._hash, hv, side="left"))
        kb = key.encode("utf-8")
        while lo < len(self._hash) and self._hash[lo] == hv:            # walk equal hashes: stored key decides
            rec = self._record(int(self._shard[lo]), int(self._offset[lo]))
            if rec[4] == kb:
                return int(self._shard[lo]), rec
            lo += 1
        return None

    def view(self, key):
        """Zero-copy read-only (h,w,4) view into the mapping.  KeyError for an unknown key."""
        hit = self._find(key)
        if hit is None:
            raise KeyError(key)
        si, (h, w, flags, crc, _, pstart) = hit
        a = np.frombuffer(self.maps[si], dtype=np.uint8, count=h * w * 4, offset=pstart).reshape(h, w, 4)
        if self.verify_crc and (zlib.crc32(a) & 0xFFFFFFFF) != crc:
            raise IOError("pool record %r: crc mismatch (corrupt shard)" % key)
        return a

    def loader(self, key):
        """`InstPool(loader=store.loader)`: a private writable copy, as `_load_RGBA` edits the alpha in place."""
        return np.array(self.view(key))

    def keys(self):
        for si, off in zip(self._shard, self._offset):
            yield self._record(int(si), int(off))[4].decode("utf-8")

    def close(self):
        self._hash = self._offset = self._shard = None
        for mm in self.maps:
            try:
                mm.close()
            except BufferError:        # views still alive: the mapping goes with them
                pass
        self.maps = []
